// p3m_internal.h -- context, geometry and launch helpers shared by the HIP translation units.
// gfx950 only (wave64, 256 CUs / 8 XCDs, 160 KiB LDS per CU).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/p3m_hip.h"

#define P3M_PI_F 3.141592654f                 /* cubepm.par:148 */
#define P3M_G_F (1.0f / 6.0f / P3M_PI_F)      /* cubepm.par:149 */
#define P3M_EPS_F 1.0e-03f                    /* cubepm.par:150 */

void p3m_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      p3m_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));      \
      return (_e == hipErrorOutOfMemory) ? P3M_ENOMEM : P3M_EDEVICE;                           \
    }                                                                                          \
  } while (0)
#define P3M_TRY(expr)            \
  do {                           \
    int _r = (expr);             \
    if (_r != P3M_OK) return _r; \
  } while (0)

#define P3M_CAND_SLOTS 64
#define P3M_GL_SLOTS 8
#define P3M_NSLOT 64
#define P3M_SUM_SPAN (P3M_NSLOT * 8)    // doubles per reduced sum
#define P3M_RED_SPAN (P3M_NSLOT * 16)   // floats per reduced maximum
#ifdef __HIPCC__
// Running maximum of non-negative floats (their bit patterns order like unsigned integers).  The atomic is only issued when the
// value beats what a plain look at the word shows: atomics on ONE address serialise at ~12 ns each on this part, and a kernel
// cannot retire before its atomics have -- 270 000 wavefronts each posting their maximum to one word are 3.2 ms, however little
// the wavefronts compute (that was the extended-PP kernel at uniform density).  A stale look only costs a superfluous atomic.
__device__ __forceinline__ void p3m_atomic_max_nonneg(float *addr, float v) {
  if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}
__device__ __forceinline__ int p3m_slot() { return (int)((blockIdx.x + 13u * blockIdx.y + 31u * blockIdx.z) & (P3M_NSLOT - 1)); }
#endif

#ifdef __HIPCC__
// Cross-lane scans and reductions by DPP (data-parallel primitives: an operand modifier of the vector instruction) instead of __shfl_*, which
// compile to ds_bpermute -- an LDS round trip per step, six dependent ones per scan: in the latency-bound particle kernels (one wavefront per
// row, a chain of round trips) that chain is time, not instructions.  Four row_shr steps inside the rows of 16 lanes, then the rows' totals
// by row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3); a lane without a source keeps `old` (the neutral element).
__device__ __forceinline__ int wave_scan_incl_i(int v) {   // inclusive prefix sum over the 64 lanes
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_scan_incl_f(float v) {   // the same on floats (lane 63: the wavefront's sum, added in lane order by halves)
  auto step = [&](auto CTRL, auto RM) { v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(CTRL)::value, decltype(RM)::value, 0xf, false)); };
  step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}); step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return v;
}
// the maximum of a non-negative float over the wavefront, in lane 63 (0 where a step has no source lane)
__device__ __forceinline__ float wave_max_nonneg_to_last(float v) {
  auto step = [&](auto CTRL, auto RM) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(CTRL)::value, decltype(RM)::value, 0xf, false);
    v = fmaxf(v, __int_as_float(t));
  };
  step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}); step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return v;
}
__device__ __forceinline__ int wave_max_nonneg_to_last_i(int v) {
  auto step = [&](auto CTRL, auto RM) { v = max(v, __builtin_amdgcn_update_dpp(0, v, decltype(CTRL)::value, decltype(RM)::value, 0xf, false)); };
  step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}); step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return v;
}
__device__ __forceinline__ int rec_index(const float4 &p) { return __float_as_int(p.w); }
__device__ __forceinline__ float4 with_index(float x, float y, float z, int i) { return make_float4(x, y, z, __int_as_float(i)); }
#endif

// ------------------------------------------------------------------ per-phase GPU times of a step (timers.f90:68-77, -DMPI_TIME)
// Off by default (p3m_hip_phase_timing / p3m_hip_group_phase_timing): a span is a pair of timing events on the stream the phase is
// queued on; after the step's last synchronisation the spans are folded into one figure per phase.  The coarse transform runs on the
// second stream underneath the fine mesh: its span overlaps the others'.
enum { P3M_PH_DRIFT = 0, P3M_PH_SORT, P3M_PH_GHOST, P3M_PH_FINE_DEPOSIT, P3M_PH_FINE_FFT, P3M_PH_FINE_KICK, P3M_PH_PP_INTRA, P3M_PH_PP_EXT,
       P3M_PH_COARSE_DEPOSIT, P3M_PH_COARSE_FORCE, P3M_PH_COARSE_KICK, P3M_PH_DELETE, P3M_NPHASE };
struct PhaseTimer {
  bool on = false;
  std::vector<hipEvent_t> pool; size_t used = 0;
  struct Span { int phase; hipEvent_t a, b; };
  std::vector<Span> spans;
  static_assert(P3M_NPHASE == P3M_NPHASES, "the phase enum and the public header's P3M_NPHASES");
  float ms[P3M_NPHASE] = {0};
  hipEvent_t take() {
    if (used == pool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
    return pool[used++];
  }
  void reset() { used = 0; spans.clear(); }
  void collect() {   // after a stream synchronisation that covers every span
    for (float &m : ms) m = 0.f;
    for (const Span &s : spans) { float t = 0.f; if (s.a && s.b && hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) ms[s.phase] += t; }
    reset();
  }
  ~PhaseTimer() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};
struct PhaseScope {   // one span: constructor and destructor record on `stream`
  PhaseTimer *t; hipStream_t s; int idx = -1;
  PhaseScope(PhaseTimer *timer, int phase, hipStream_t stream) : t(timer && timer->on ? timer : nullptr), s(stream) {
    if (!t) return;
    hipEvent_t a = t->take(), b = t->take();
    if (!a || !b) { t = nullptr; return; }
    (void)hipEventRecord(a, s);
    idx = (int)t->spans.size(); t->spans.push_back({phase, a, b});
  }
  ~PhaseScope() { if (t) (void)hipEventRecord(t->spans[idx].b, s); }
};

// ------------------------------------------------------------------ one launch for the small buffers a step accumulates into
// A step zeroes a dozen counters, histograms and reduction slots per rank; as hipMemsetAsync calls each is a launch of its own (5 us of
// kernel + the gap around it, ~90 per step on 8 ranks).  zero_add collects (pointer, bytes) pairs, zero_flush clears them in ONE launch.
#define P3M_ZERO_MAX 12
struct ZeroList { void *p[P3M_ZERO_MAX]; unsigned long long n[P3M_ZERO_MAX]; int cnt; };   // bytes: multiples of 4, pointers 4-byte aligned

// ------------------------------------------------------------------ FFT plan (fft.hip)
struct FftPlan {
  int n = 0;               // real transform length per axis
  int px = 0;              // complex row pitch: n/2+1 rounded up to 16 (128-byte lines)
  int nfac_full = 0, fac_full[16];  // radices of n      (strided c2c passes)
  int nfac_half = 0, fac_half[16];  // radices of n/2    (packed real x pass)
  float2 *d_tw = nullptr;  // exp(-2 pi i q / n), q in [0,n)
};

// Derived sizes of cubepm.par:186-208 plus the device layout.
struct Geometry {
  int nodes_dim, nodes, T, ntiles, nf, nb, ncut, ms, pp_range;
  int pt;      // nf_physical_tile_dim
  int Nn;      // nf_physical_node_dim
  int E;       // extended fine cells per axis: Nn + 2*nb  (ghost zone included)
  int nc_buf, nct, ncn, nc, nc_slab;
  int hx;      // nf/2+1 complex per row (reference layout)
  int px;      // device complex row pitch of a fine array (fft plan), real pitch = 2*px
  int pxc;     // same for the coarse mesh
  int fb;      // force box edge: pt+3   (force_f(3, nb-1:nf-nb+1,...), cubep3m.fh:36)
  int fbp;     // row pitch of the force box on the device: fb rounded up to 4 floats (16-byte row stores)
  int cart[3]; // z,y,x rank coordinates (mpi_initialization.f90:60-64)
  int nbr[6];  // -z,+z,-y,+y,-x,+x
  int64_t max_np;
};

struct p3m_ctx {
  p3m_params p;
  Geometry g;
  int device = 0;
  hipStream_t stream = nullptr;
  // ---- particle store (persistent between steps): cubep3m.fh:75-79 as SoA
  int64_t cap = 0;        // capacity in records (physical + ghosts)
  int np_local = 0;       // physical particles
  int np_all = 0;         // after the ghost pass
  // Records (round 3): only the POSITIONS are sorted.  Velocity and PID stay where they are and are reached through an index
  // carried in the fourth lane of the records (integer bits; never touched by arithmetic):
  //   pos[i]   arrival order   {x, y, z, -}
  //   vel[i]   arrival order   {vx, vy, vz, slot}  slot = index of the record's PID in pid_home (the kicks gather / scatter these
  //                            16-byte records through spos[s].w; 12-byte velocity records were measured slower: a gather of
  //                            records that straddle cache lines costs more than the 4 bytes it saves)
  //   tpos[j]  row-bucketed    {x, y, z, i}        i = arrival index        (intermediate of the sort)
  //   spos[s]  cell-sorted     {x, y, z, i}
  //   pid_home[slot]           written at upload and when a migrant arrives, read at download and when a migrant leaves
  // delete_particles (the compaction, deferred into the next drift) writes the next arrival arrays into the buffers the sort's
  // intermediate used (pos <-> tpos, vel <-> vel_alt swap roles).  A step moves 16 B per record through the two sort
  // passes instead of 16 + 16 + 8 + 4.
  float4 *pos = nullptr; float4 *vel = nullptr, *vel_alt = nullptr;
  int64_t *pid_home = nullptr; int n_home = 0;   // slots in use (upper bound: departed records leave holes; repacked when full)
  float4 *spos = nullptr;      // sorted by extended fine cell
  int *cell_end = nullptr;     // [E^3+1] inclusive prefix of per-cell counts, shifted by one: start(c)=cell_end[c], end(c)=cell_end[c+1]
  int *row_end = nullptr;      // [E^2+1] the same for whole x-rows of cells (first level of the sort)
  // PM-only NGP whole steps need cell offsets at few places only: per x-row the ncn+2 starts at stride mesh_scale the
  // coarse deposit reads and, per tile column, the start and end of the force-box row the kick walks.  The sort then
  // writes this compact table instead of the 4 E^3 bytes of cell_end (cells_compact == true); anything else that reads
  // cell_end calls particles_full_cells first, which rebuilds it from the sorted records.
  int *crow = nullptr; int crow_w = 0; bool cells_compact = false;
  float4 *tpos = nullptr;      // row-bucketed intermediate of the sort: position and arrival index
  int *scan_tmp = nullptr; size_t scan_tmp_n = 0;
  unsigned long long *scan_state = nullptr; size_t scan_state_n = 0; unsigned scan_epoch = 0;   // the one-launch scan's look-back words (scan.hip)
  int *flags = nullptr;        // [cap] compaction flags / offsets
  unsigned char *cflag = nullptr; // [(E/ms)^3] coarse cells holding a record whose tile-local cell differs from floor(x)
  // sorted indices of the records within 2^-10 below a cell face (the only ones xv + offset_tile can round into the next cell):
  // P3M_CAND_SLOTS lists of cand_seg entries each, filled by k_row_sort (list = row mod slots); their lengths sit on cache lines
  // of their own in cand_cnt[slot * 16] -- appending to ONE list cost k_row_sort half its run time (66 000 atomics on one
  // address serialise at ~12 ns each); cand_cnt[16 * slots] is set when a list overflowed: the fix-up then scans every record
  int *cand = nullptr; int *cand_cnt = nullptr; int cand_seg = 0;
  unsigned char *pp_intra_done = nullptr; bool pp_intra_fused = false;   // extended PP with -DPPINT fused (pp.hip): a byte per sorted record whose bucket pairs are summed already
  bool step_begun = false;     // p3m_hip_particle_mesh: the step has changed state (an error from here on resets what was queued)
  int *pp_plan = nullptr, *pp_task_group = nullptr, *pp_counter = nullptr, *pp_htask = nullptr, *pp_slow = nullptr;   // extended PP (pp.hip): first task of every patch, task -> {patch, sub-task}, task counters, the heavy-task list
  int *d_counters = nullptr;   // small device counter block
  int *h_counters = nullptr;   // pinned mirror
  // ---- fine mesh, all tiles batched
  int tile_batch = 0;          // tiles processed per sweep
  bool pending_compact = false; int pend_n = 0; float pend_mb[3] = {0, 0, 0};   // deferred ghost removal (particles.hip)
  hipStream_t stream2 = nullptr; hipEvent_t ev_dep = nullptr, ev_cf = nullptr;   // single-rank whole steps: the coarse force forms on stream2 underneath the fine-mesh force sweep
  int sort_ncur = 0;           // records handed to the sort queued by particles_sort_enqueue
  int cnt_from_kick = 0;       // records the NGP kick counted into c->flags per block of 256 (survivors of delete_particles without move_grid_back); 0: not counted
  // Ghost candidates: the arrival indices of the records within nf_buf of a face of the rank's volume (the only ones the ghost
  // pass sends anywhere), listed by the compaction + drift pass that forms their positions -- P3M_GL_SLOTS lists in the dead
  // arrival buffer (tpos after the swap), their lengths on cache lines of their own in gl_cnt[slot * 16].  Valid exactly as long
  // as hist_done is (same writers, same invalidations); the ghost pack then reads a quarter of the records
  int *gl_cnt = nullptr; int gl_cap = 0; bool gl_valid = false; int64_t gl_longest = 0;   // gl_longest: no list holds more (the records of its blocks)
  bool hist_done = false;      // the x-row counts of the next sort were accumulated while the arrival arrays were written (particles.hip)
  bool finalize_queued = false; // particles_finalize_enqueue ran, particles_finalize_finish has not
  bool lazy_counters = false;  // whole steps: the sort's deleted count has not been read yet, np_all is its upper bound (the tail is padded)
  bool coarse_first = false;   // whole-step PM-only NGP runs: the coarse force is ready before the fine kick, which then adds the coarse kick in the same pass
  bool rho_from_sort = false;  // the sort of this step already wrote the NGP density of every tile (particles.hip)
  bool rho_u8_force = false; float rho_mass = 0.f;   // fine_deposit -> fine_force: this sweep's density is bytes; its mass_p
  bool rho_u8 = false;         // ... as one byte per cell (its count) at the head of the rho array (RowDep::rho8; fft3d_forward_xy turns counts into masses)
  bool cell_max_reported = false, rho_u8_step = false;   // this step's sort reports the count / wrote bytes (rho_u8_check after the fold)
  bool cell_max_known = false; float cell_max = 0.f;   // the largest fine-cell count the last whole step's sort saw (0: below 64); unknown after an upload
  // NGP steps (round 5): the force phase stops after the inverse y pass (work holds LY rows of the three components) and the inverse x
  // pass runs inside the kick (kick_fused.hip), fuse_nr box rows per batch; rowflag[ntiles*fb*fb]: rows that also go to the force box
  bool xinv_deferred = false; int fuse_nr = 0; unsigned char *rowflag = nullptr;
  float *rho = nullptr;        // [batch][nf][nf][2*px]  density -> rho-hat
  float *work = nullptr;       // [3][batch][nf][nf][2*px]  i*K_c*rho-hat -> force, all three components
  float *fbox = nullptr;       // [3][ntiles][fb][fb][fbp] extracted force (SoA planes, pad columns zero)
  float *kern_f = nullptr;     // [3][nf][nf][px]  SoA planes of kern_f
  FftPlan plan_f;
  // ---- coarse mesh
  float *rho_c = nullptr;      // [ncn][ncn][ncn]
  float *cmom = nullptr;       // [8][(ncn+1)^3] corner sums of the coarse CIC deposit (coarse_mesh.hip)
  float *slab = nullptr;       // [nc][nc][2*pxc] (single rank) density -> hat
  float *slab_w = nullptr, *slab_o = nullptr;  // scratch (LY) and real output of the inverse
  float *force_c = nullptr;    // [3][ncn+2][ncn+2][ncn+2] SoA planes incl. halo
  float *kern_c = nullptr;     // [3][nc][nc][pxc]
  FftPlan plan_c;
  bool have_kf = false, have_kc = false;
  bool kf_zmirror = false;   // kern_f is exactly mirror-symmetric in kz (build_fine_kernel: kf_symmetrise_z): the fused z pass reads the lower half only
  // ---- per-step reductions (device) and results
  // Reduced scalars live in P3M_NSLOT slots of one 64-byte line each: a kernel's workgroups spread their
  // atomics over the slots (tens of thousands of atomics on ONE address serialise at ~12 ns each), the host
  // folds the slots after the download (reductions_fold).
  float *d_red = nullptr;      // [8][P3M_RED_SPAN]: [0] f_force_max^2 [1] pp_force_max [2] c_force_max [3..] scratch
  float *d_tile_ext = nullptr; // [ntiles] per-tile pp_ext max
  double *d_sums = nullptr;    // [4][P3M_SUM_SPAN]: [0] sum rho_f (interior) [1] sum rho_c
  float *h_red_raw = nullptr; double *h_sums_raw = nullptr; float *h_tile_ext = nullptr;  // pinned mirrors
  char *d_redblk = nullptr, *h_redblk = nullptr; size_t red_bytes = 0;   // d_sums | d_red | d_tile_ext are ONE allocation (one download, one clear)
  float h_red[8] = {0}; double h_sums[4] = {0};   // folded values
  p3m_step_out last{};
  int np_ghost = 0, np_deleted = 0;
  // ---- transport
  ZeroList zl{}; bool step_zeroed = false;   // whole steps: everything the step accumulates into was cleared in one launch (step_prezero, p3m_api.hip)
  PhaseTimer *pt = nullptr; bool own_pt = false;   // per-phase times (a group's contexts share the group's timer)
  p3m_transport transport{}; bool have_transport = false;   // unused: exchanges belong to the group (group.hip)
  void *rccl_comm = nullptr;
};

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// A P3M_FLAG_COARSE_ONLY context holds the coarse mesh of its rank and nothing else (no particle store, no fine mesh: p3m_api.hip):
// every entry point that touches records, cells or fine arrays refuses it instead of launching kernels on null pointers
static inline int need_particles(const p3m_ctx *c, const char *what) {
  if (c->p.flags & P3M_FLAG_COARSE_ONLY) { p3m_set_error("%s on a P3M_FLAG_COARSE_ONLY context (it holds the coarse mesh only)", what); return P3M_ESTATE; }
  return P3M_OK;
}

// ---- fft.hip
extern int p3m_ctx_share_hint;   // p3m_api.hip: contexts that share the device's free memory (set by group.hip around p3m_hip_create)
int fft_plan_create(FftPlan *pl, int n);
void fft_plan_destroy(FftPlan *pl);
// batched 3-D r2c: data holds real ROWS ([n][n][2*px]) on entry and rho-hat in the bundle layout LZ on
// exit (see fft.hip "memory layouts"); scratch is one more array of the same size
int fft3d_forward(p3m_ctx *c, const FftPlan &pl, float *data, float *scratch, int batch);
// out (real ROWS) <- c2r(hat [* i*kern]) / n^3 ; hat and kern in LZ; tmp: scratch array; out != tmp
// (the multiply is particle_mesh_threaded.f90:183-192; hat is left intact)
int fft3d_inverse(p3m_ctx *c, const FftPlan &pl, const float *hat, float *tmp, float *out, int batch, const float *kern);
int fft_rows_to_lz(p3m_ctx *c, const FftPlan &pl, const float *rows, float *lz);
int fft_lz_to_rows(p3m_ctx *c, const FftPlan &pl, const float *lz, float *rows);

// ---- particles.hip
int particles_drift(p3m_ctx *c, float dt, float dt_old, const float *offset);
int particles_pass_and_sort(p3m_ctx *c);
int particles_pass_self(p3m_ctx *c);
int particles_sort(p3m_ctx *c, float deposit_mass);
int particles_sort_enqueue(p3m_ctx *c, float deposit_mass);   // the device half ...
int particles_sort_finish(p3m_ctx *c, bool wait);             // ... and the host half (wait: one stream sync for the exact counters; else deferred)
void particles_collect_counters(p3m_ctx *c);                  // deferred counters, after the step's synchronisation
int particles_full_cells(p3m_ctx *c);   // cell_end valid again after a sort that wrote the compact table only
int particles_ghost_pack(p3m_ctx *c, float4 *sbuf, const int64_t *seg_off, const int *seg_cap, int *d_counts);
int particles_ghost_unpack(p3m_ctx *c, const float4 *rbuf, const int64_t *seg_off, const int *cnt, int base);
int particles_finalize(p3m_ctx *c, const float *move_back);
int particles_finalize_enqueue(p3m_ctx *c, const float *move_back);
int particles_finalize_finish(p3m_ctx *c, bool wait);
int particles_preload();
int scan_reserve(p3m_ctx *c, int64_t n_max);   // scan.hip
int particles_compact(p3m_ctx *c, bool drift, float dt, float dt_old, const float *offset);
void particles_reset_after_error(p3m_ctx *c);   // p3m_api.hip: consistent state after a step that failed half-way
int particles_resolve(p3m_ctx *c);   // finish a deferred ghost removal before the arrival arrays are read

// ---- fine_mesh.hip
int fine_deposit(p3m_ctx *c, int tile0, int ntile, float mass_p, bool fuse = false);   // fuse: flag the box rows the fused kick must also store
int fine_force(p3m_ctx *c, int tile0, int ntile, bool defer_x = false);                // defer_x: stop after the inverse y pass (the x pass runs with the kick)
int fine_time_fused_kick(p3m_ctx *c);   // timing hook: the fused inverse-x + kick pass, dry
bool fine_kick_fusable(const p3m_ctx *c);                                               // fine_mesh.hip: this context's whole steps run the fused inverse-x + kick pass
int fine_max_and_kick(p3m_ctx *c, float a_mid, float dt, bool count_survivors = true);   // false: timing hook (leaves c->flags and cnt_from_kick alone)
int fine_projection(p3m_ctx *c, float mass_p, float *d_pxy, float *d_pxz, float *d_pyz);   // projection.f90: adds this rank's tiles to the device maps
bool coarse_kick_rides_on_fine(const p3m_ctx *c);   // p3m_api.hip
int fine_mesh_force_phase(p3m_ctx *c, float mass_p, bool may_clear);          // p3m_api.hip: density + force box of every tile; may_clear: phase-level call, the reductions were not cleared before the sort
int fine_mesh_kick_phase(p3m_ctx *c, float a_mid, float dt, float mass_p);    // maximum, kick, PP
int fine_force_max(p3m_ctx *c);
int fine_sum_mass(p3m_ctx *c, int tile0, int ntile);
int build_fine_kernel(p3m_ctx *c, const float *table16_host);

// ---- pp.hip
int rho_u8_check(p3m_ctx *c);   // p3m_api.hip
bool fft_x_forward_reads_u8(const FftPlan &pl);   // fft.hip: the tile size has a forward x pass that reads one byte per cell
int pp_intra(p3m_ctx *c, float a_mid, float dt, float mass_p);
int pp_extended(p3m_ctx *c, float a_mid, float dt, float mass_p, bool fuse_intra);   // fuse_intra: sum the -DPPINT bucket pairs on the way where the lean light pass runs (pp_intra, called AFTERWARDS, then works the rest)

// ---- coarse_mesh.hip
int coarse_deposit(p3m_ctx *c, float mass_p);
int coarse_force(p3m_ctx *c);
int coarse_kick(p3m_ctx *c, float a_mid, float dt);
int build_coarse_kernel(p3m_ctx *c, const float *table4_host);

// ---- scan.hip
int reductions_clear(p3m_ctx *c);      // zero d_red / d_sums
int reductions_download(p3m_ctx *c);   // enqueue the copies to the pinned mirrors
void reductions_fold(p3m_ctx *c);      // after the stream sync: slots -> c->h_red[], c->h_sums[]
int exclusive_scan_i32(p3m_ctx *c, int *data, int64_t n);  // in place; data[n] (one past) receives the total
int zero_add(p3m_ctx *c, void *p, size_t bytes);   // scan.hip: queue a buffer for zero_flush (flushes first when the list is full)
int zero_flush(p3m_ctx *c);                        // one launch on c->stream
int step_prezero(p3m_ctx *c);                      // p3m_api.hip: whole steps, after the drift
