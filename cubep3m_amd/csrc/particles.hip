// particles.hip -- particle bookkeeping of the gravity step on the device:
//   update_position.f90:68-76   -> k_drift
//   link_list.f90 + particle_pass.f90 (single rank: periodic self exchange) -> k_pass_axis
//   link_list's chaining mesh hoc/ll (and llf, hoc_fine/ll_fine) -> sort by extended fine cell:
//     k_row_hist / exclusive scan / k_row_scatter (by x-row), k_row_sort (inside each row)
//   delete_particles.f90 + move_grid_back.f90 -> k_count_physical / scan of the block counts / k_compact
// Records: float4 positions whose fourth lane carries an index (p3m_internal.h), float4 velocities in arrival order whose fourth
// lane carries the PID slot (12-byte velocity records were measured and dropped: DESIGN section 5, round 3), PIDs at rest in pid_home.  Everything here is HBM-bound streaming
// (the row histogram / scatter aggregate their atomics per block in LDS).
#include "p3m_internal.h"
#include <stdlib.h>
#include <algorithm>

#define PT 256

// ------------------------------------------------------------------ update_position.f90:68-76
__global__ __launch_bounds__(PT) void k_drift(float4 *__restrict__ pos, const float4 *__restrict__ vel, int n, float dt, float dt_old,
                                              float ox, float oy, float oz, int use_off) {
  const int i = blockIdx.x * PT + threadIdx.x;
  if (i >= n) return;
  float4 p = pos[i]; const float4 v = vel[i];
  const float hs = 0.5f * (dt + dt_old);
  if (use_off) { p.x = p.x + v.x * hs + ox; p.y = p.y + v.y * hs + oy; p.z = p.z + v.z * hs + oz; }  // :71
  else { p.x = p.x + v.x * hs; p.y = p.y + v.y * hs; p.z = p.z + v.z * hs; }                           // :73
  pos[i] = p;
}

int particles_drift(p3m_ctx *c, float dt, float dt_old, const float *offset) {
  if (c->pending_compact) return particles_compact(c, true, dt, dt_old, offset);   // ghost removal of the last step + this drift in one pass
  c->hist_done = false; c->gl_valid = false;                                        // positions change in place: no row counts carried
  if (c->np_local == 0) return P3M_OK;
  hipLaunchKernelGGL(k_drift, dim3(cdiv(c->np_local, PT)), dim3(PT), 0, c->stream, c->pos, (const float4 *)c->vel, c->np_local, dt, dt_old,
                     offset ? offset[0] : 0.f, offset ? offset[1] : 0.f, offset ? offset[2] : 0.f, offset ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

__device__ __forceinline__ float comp(const float4 &p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }
__device__ __forceinline__ void setcomp(float4 &p, int a, float v) { if (a == 0) p.x = v; else if (a == 1) p.y = v; else p.z = v; }
__device__ __forceinline__ bool in_hoc_range(const float4 &p, float lo, float hi) {
  // link_list.f90:26-31: floor(x/mesh_scale)+1 within hoc_nc_l..hoc_nc_h  <=>  -nf_buf <= x < Nn+nf_buf
  return p.x >= lo && p.x < hi && p.y >= lo && p.y < hi && p.z >= lo && p.z < hi;
}

// particle_pass.f90 on a single rank (the rank is its own +/- neighbour in every direction).
// The three sequential axis exchanges (+x,-x | re-link | -y,+y | re-link | +z,-z) deliver, for every
// record, the Cartesian product of its per-axis image sets: coordinate a contributes itself, plus
// max(x_a-Nn,-nb) if x_a >= Nn-nb (:83,:162), plus min(guard(x_a)+Nn, Nn+nb-eps) if x_a < nb
// (:185,:257-265); an axis' second direction never re-sends what its first direction delivered (not
// yet in hoc), and the later axes see all earlier arrivals (re-link :274-298).  One kernel writes all
// images: per-block exclusive scan of the image counts, one atomic per block for the base.
__device__ __forceinline__ int axis_images(float x, float Nn, float nb, float out[3]) {
  int n = 0;
  out[n++] = x;
  if (x >= Nn - nb) out[n++] = fmaxf(x - Nn, -nb);
  if (x < nb) {
    float xs = x;
    if (fabsf(xs) < P3M_EPS_F) xs = (xs < 0.0f) ? -P3M_EPS_F : P3M_EPS_F;
    out[n++] = fminf(xs + Nn, Nn + nb - P3M_EPS_F);
  }
  return n;
}
// ------------------------------------------------------------------ sort by extended fine cell
// cell = ((cz*E + cy)*E + cx), c_d = floor(x_d) + nb in [0,E).  cs[c] = start(c), cs[c+1] = end(c).
// Two levels, because a device-scope atomic costs a fabric transaction and the cell array (E^3 ints) is ten
// times larger than the particle arrays at the reference's density of 1/8 per fine cell:
//   1. counting sort by x-ROW (cz*E + cy): k_row_hist / exclusive scan of E^2 counters / k_row_scatter.
//      Records arrive nearly sorted (the previous step's order, ghosts appended face by face), so a block of
//      2048 records touches a few dozen rows: counts and cursor reservations are aggregated per block in an
//      LDS table and hit global memory once per (block, row).  A full table (random order, first step)
//      degrades to one global atomic per record, never to a wrong result.
//   2. k_row_sort: one wavefront per row orders the row's records by x cell with an LDS histogram and writes
//      the row of cs[] -- write-only, no memset, no scan, no atomics on the big array.
// k_row_hist also flags the coarse (hoc) cells that hold a physical record whose TILE-LOCAL fine cell
// floor(x + offset_tile) (particle_mesh_threaded.f90:248-249) differs from floor(x): the intra-cell PP
// buckets of such a coarse cell (:276-284) are not the sorted fine cells and take the slow path.
#define SORT_RPT 8
#define SORT_HB 512   // entries of the per-block (row -> count/base) table, a power of two
__device__ __forceinline__ int rowtab_slot(int *key, int row) {
  const unsigned h = ((unsigned)row * 2654435761u) >> 23;
#pragma unroll 1
  for (int t = 0; t < 8; t++) {
    const int e = (int)((h + t) & (SORT_HB - 1));
    const int k = key[e];
    if (k == row) return e;
    if (k == -1) { const int old = atomicCAS(&key[e], -1, row); if (old == -1 || old == row) return e; }
  }
  return -1;   // crowded: the caller goes to global memory directly
}
// Records arrive nearly sorted, so the lanes of a wavefront hold a few RUNS of equal rows: only the first lane of a run
// (head) touches the row table, with the run length; the others take their rank from the head.  hl: lane of this
// lane's head, cnt: run length (valid in the head).  An unsorted first step degrades to one-lane runs, i.e. to the
// per-record table updates, plus a ballot.
__device__ __forceinline__ bool row_run(int row, int &hl, int &cnt) {
  const int lane = threadIdx.x & 63;
  const int prev = __builtin_amdgcn_update_dpp(row, row, 0x138, 0xf, 0xf, false);   // wave_shr:1 (an operand modifier, not the LDS round trip of __shfl_up; lane 0 heads a run anyway)
  const bool head = lane == 0 || row != prev;
  const unsigned long long hm = __ballot(head);
  const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  hl = 63 - __clzll(hm & upto);
  const unsigned long long above = hm & ~upto;
  cnt = (above ? __ffsll((long long)above) - 1 : 64) - lane;
  return head;
}
// Row-histogram pieces for the kernels that PRODUCE the arrival arrays (k_compact_drift_hist, k_ghost_unpack, k_make_images):
// in steady state every record passes through one of them right before the sort, so they count the x-rows on the way and
// k_row_hist's pass over the positions (16 B per record) is skipped.
// the LDS counter through a pointer that is LDS by its type: `e >= 0 ? &val[e] : &rs[row + 1]` under one atomicAdd is what the compiler made of
// the two branches -- a flat atomic on a pointer selected per lane, eight per thread in k_compact_drift_hist
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void lds_add(int *val, int e, int cnt) { __hip_atomic_fetch_add((lds_int *)val + e, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void hist_row(int *key, int *val, int *rs, int row) {   // every lane of the wavefront calls (row < 0: nothing to count)
  int hl, cnt;
  if (row_run(row, hl, cnt) && row >= 0) {
    const int e = rowtab_slot(key, row);
    if (e >= 0) lds_add(val, e, cnt); else atomicAdd(&rs[row + 1], cnt);
  }
}
__device__ __forceinline__ void hist_row_one(int *key, int *val, int *rs, int row) {   // callers in divergent code
  const int e = rowtab_slot(key, row);
  if (e >= 0) lds_add(val, e, 1); else atomicAdd(&rs[row + 1], 1);
}
__device__ __forceinline__ void hist_flush(const int *key, const int *val, int *rs) {   // behind a __syncthreads()
  for (int e = threadIdx.x; e < SORT_HB; e += blockDim.x) if (val[e] > 0) atomicAdd(&rs[key[e] + 1], val[e]);
}
// the coarse cell of a physical record whose tile-local fine cell differs from floor(x) (see k_row_hist) is flagged
__device__ __forceinline__ void flag_displaced(const float4 &p, float nb, int E, int ms, int pt, unsigned char *cflag) {
  const int nct = pt / ms; const float xs[3] = {p.x, p.y, p.z}; bool displaced = false; int cc[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    cc[d] = (int)floorf(xs[d] / (float)ms);
    const int t = cc[d] / nct;
    const float xl = xs[d] + (nb - (float)(t * pt));
    displaced = displaced || ((int)floorf(xl) != (int)floorf(xs[d]) + (int)nb - t * pt);
  }
  if (displaced) { const int Ec = E / ms, cb = (int)nb / ms; cflag[((cc[2] + cb) * Ec + (cc[1] + cb)) * Ec + (cc[0] + cb)] = 1; }
}
template <bool HIST>   // HIST: the images' x-rows are counted on the way (rs; the originals were counted by k_compact_drift_hist)
__global__ __launch_bounds__(PT) void k_make_images(float4 *__restrict__ pos, float4 *__restrict__ vel, int n_cur,
                                                    int cap, float Nn, float nb, int *__restrict__ counter, int *__restrict__ overflow, int E, int *__restrict__ rs,
                                                    unsigned char *__restrict__ cflag, int ms, int pt) {   // cflag: see flag_displaced (PPINT runs)
  __shared__ int wsum[PT / 64];
  __shared__ int base_sh;
  __shared__ int key[HIST ? SORT_HB : 1], val[HIST ? SORT_HB : 1];
  if (HIST) { for (int e = threadIdx.x; e < SORT_HB; e += PT) { key[e] = -1; val[e] = 0; } }
  const int i = blockIdx.x * PT + threadIdx.x;
  float ox[3], oy[3], oz[3]; int nx = 1, ny = 1, nz = 1, cnt = 0;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n_cur) {
    p = pos[i];
    if (in_hoc_range(p, -nb, Nn + nb)) {  // else dropped by link_list ("PARTICLE DELETED")
      nx = axis_images(p.x, Nn, nb, ox); ny = axis_images(p.y, Nn, nb, oy); nz = axis_images(p.z, Nn, nb, oz);
      cnt = nx * ny * nz - 1;
    }
  }
  // block exclusive scan of cnt
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int inc = wave_scan_incl_i(cnt);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < PT / 64; k++) { if (k < w) off += wsum[k]; tot += wsum[k]; }
  if (threadIdx.x == 0) base_sh = tot ? atomicAdd(counter, tot) : 0;
  __syncthreads();
  if (cnt != 0) {
    int s = n_cur + base_sh + off + inc - cnt;
    const float4 v = vel[i];
    for (int c = 0; c < nz; c++)
      for (int b = 0; b < ny; b++)
        for (int a = 0; a < nx; a++) {
          if ((a | b | c) == 0) continue;
          if (s < cap) {
            pos[s] = make_float4(ox[a], oy[b], oz[c], 0.f); vel[s] = v;   // v.w: the image keeps the PID slot of its original (same rank)
            if (HIST) {
              hist_row_one(key, val, rs, ((int)floorf(oz[c]) + (int)nb) * E + (int)floorf(oy[b]) + (int)nb);   // images lie inside [-nb, Nn+nb)
              // the image of a record that drifted out of [0, Nn) is the physical one: the flag k_row_hist would have set for it
              if (cflag && ox[a] >= 0.f && ox[a] < Nn && oy[b] >= 0.f && oy[b] < Nn && oz[c] >= 0.f && oz[c] < Nn) flag_displaced(make_float4(ox[a], oy[b], oz[c], 0.f), nb, E, ms, pt, cflag);
            }
          } else *overflow = 1;
          s++;
        }
  }
  if (HIST) { __syncthreads(); hist_flush(key, val, rs); }
}

__global__ __launch_bounds__(PT) void k_row_hist(const float4 *__restrict__ pos, int n, int np_orig, float Nn, float nb, int E,
                                                 int *__restrict__ rs, int *__restrict__ ndeleted, unsigned char *__restrict__ cflag, int ms, int pt) {
  __shared__ int key[SORT_HB], val[SORT_HB];
  for (int e = threadIdx.x; e < SORT_HB; e += PT) { key[e] = -1; val[e] = 0; }
  __syncthreads();
  float4 pl[SORT_RPT];
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {   // all loads first: eight independent requests in flight per lane
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    pl[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) pl[u] = pos[i];
  }
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    const float4 p = pl[u];
    const bool inr = i < n && in_hoc_range(p, -nb, Nn + nb);
    const int row = inr ? ((int)floorf(p.z) + (int)nb) * E + (int)floorf(p.y) + (int)nb : -1;
    int hl, cnt;
    if (row_run(row, hl, cnt) && row >= 0) {
      const int e = rowtab_slot(key, row);
      if (e >= 0) lds_add(val, e, cnt); else atomicAdd(&rs[row + 1], cnt);
    }
    if (i >= n) continue;
    if (inr) {
      if (cflag && p.x >= 0.f && p.x < Nn && p.y >= 0.f && p.y < Nn && p.z >= 0.f && p.z < Nn) {
        const int nct = pt / ms; const float xs[3] = {p.x, p.y, p.z}; bool displaced = false; int cc[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          cc[d] = (int)floorf(xs[d] / (float)ms);
          const int t = cc[d] / nct;
          const float xl = xs[d] + (nb - (float)(t * pt));
          displaced = displaced || ((int)floorf(xl) != (int)floorf(xs[d]) + (int)nb - t * pt);
        }
        if (displaced) { const int Ec = E / ms, cb = (int)nb / ms; cflag[((cc[2] + cb) * Ec + (cc[1] + cb)) * Ec + (cc[0] + cb)] = 1; }
      }
    } else if (i < np_orig) {
      atomicAdd(ndeleted, 1);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < SORT_HB; e += PT) if (val[e] > 0) atomicAdd(&rs[key[e] + 1], val[e]);
}

// rs[r+1] holds start(r) on entry and end(r) = start(r+1) on exit
__global__ __launch_bounds__(PT) void k_row_scatter(const float4 *__restrict__ pos, int n, float Nn, float nb, int E, int *__restrict__ rs,
                                                    float4 *__restrict__ tpos) {
  __shared__ int key[SORT_HB], val[SORT_HB];
  for (int e = threadIdx.x; e < SORT_HB; e += PT) { key[e] = -1; val[e] = 0; }
  __syncthreads();
  float4 p[SORT_RPT]; int ent[SORT_RPT], rank[SORT_RPT];   // ent: table entry | -1 dropped | -2 rank is already the global slot
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {   // all loads first
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) p[u] = pos[i];
  }
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    const bool inr = i < n && in_hoc_range(p[u], -nb, Nn + nb);
    const int row = inr ? ((int)floorf(p[u].z) + (int)nb) * E + (int)floorf(p[u].y) + (int)nb : -1;
    // the head's entry and base reach its run in ONE word (one ds_bpermute, an LDS round trip, instead of two): table entry e < 512 and
    // the base inside the block (< PT * SORT_RPT = 2^11 records) as e << 12 | base; a global slot (< 2^31) with the top bit set; -1: dropped
    static_assert(PT * SORT_RPT <= 4096 && SORT_HB <= (1 << 19), "the packed (entry, base) word");
    int hl, cnt; unsigned pk = 0xffffffffu;
    if (row_run(row, hl, cnt) && row >= 0) {
      const int e = rowtab_slot(key, row);
      if (e >= 0) pk = ((unsigned)e << 12) | (unsigned)atomicAdd(&val[e], cnt); else pk = 0x80000000u | (unsigned)atomicAdd(&rs[row + 1], cnt);
    }
    pk = (unsigned)__shfl((int)pk, hl, 64);
    const int dl = (int)(threadIdx.x & 63) - hl;
    if (pk == 0xffffffffu) { ent[u] = -1; rank[u] = 0; }
    else if (pk & 0x80000000u) { ent[u] = -2; rank[u] = (int)(pk & 0x7fffffffu) + dl; }
    else { ent[u] = (int)(pk >> 12); rank[u] = (int)(pk & 0xfffu) + dl; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < SORT_HB; e += PT) if (val[e] > 0) val[e] = atomicAdd(&rs[key[e] + 1], val[e]);   // count -> base
  __syncthreads();
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    if (ent[u] == -1) continue;
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    const int s = ent[u] >= 0 ? val[ent[u]] + rank[u] : rank[u];
    tpos[s] = with_index(p[u].x, p[u].y, p[u].z, i);   // velocity and PID stay where they are: the record carries its arrival index
  }
}

// one wavefront per row: bins[] = LDS histogram of the row's x cells -> exclusive prefix (the row of cs) -> cursors
// fused NGP deposit (rho == nullptr: off).  rho8 (round 6): the density as ONE BYTE per cell, its count -- the forward x pass turns a count into
// mass_p added count times from a table (k_fft_x_fwd2<.., U8>): a quarter of the bytes written here and read there.  cmax: where the largest
// count of a cell is reported (a reduction slot; counts beyond 255 saturate, the step then ends with an error: see particles_sort_enqueue)
struct RowDep { float *rho; double *sum_interior; float mass_p; int T, nf, pt, rp; unsigned char *rho8; float *cmax; };
// compact cell table (p3m_internal.h, crow): entry ci < ncn+2 = start of cell ms*ci - ms/2 + nb (what k_coarse_moments reads),
// entries ncn+2 + 2*tx, +1 = start of cells tx*pt + lo and tx*pt + lo + fb (the force-box row of tile column tx)
struct RowCompact { int *crow; int w, ncn, ms, T, pt, lo, fb; };   // crow == nullptr: write the full cell_end row
#ifndef P3M_SORT_WPB
#define P3M_SORT_WPB 1      // rows (wavefronts) per workgroup
#endif
// the rows of a workgroup never wait for each other: their barriers are wavefront-wide (LDS operations of a wavefront complete in order)
__device__ __forceinline__ void row_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__global__ __launch_bounds__(64 * P3M_SORT_WPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_row_sort(const float4 *__restrict__ tpos, const int *__restrict__ rs, int E, int nrows, float nb, int *__restrict__ cs,
                                                 float4 *__restrict__ spos,
                                                 int *__restrict__ cand, int *__restrict__ cand_cnt, int cand_seg, RowDep dep, RowCompact cc, int n_cur) {
  extern __shared__ int bins_all[];
  // (rows to the XCDs in contiguous eighths -- the lines two neighbouring rows share in one L2 -- was measured: 318 us against 306)
  const int row = blockIdx.x * P3M_SORT_WPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= nrows) return;
  int *bins = bins_all + (threadIdx.x >> 6) * E;
  const int r0 = rs[row], r1 = rs[row + 1];
  // the first 128 records of the row (a row holds ~70 at the reference's density) live in registers from here on: their
  // positions feed both the histogram and the scatter.  Only positions move: the fourth lane carries the arrival index
  // through which the kicks and the compaction reach velocity and PID (p3m_internal.h)
  constexpr int RR = 2;
  float4 rp[RR]; bool rin[RR];
#pragma unroll
  for (int u = 0; u < RR; u++) { const int i = r0 + u * 64 + lane; rin[u] = i < r1; rp[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (rin[u]) rp[u] = tpos[i]; }
  for (int j = lane * 4; j < E; j += 256) *reinterpret_cast<int4 *>(bins + j) = make_int4(0, 0, 0, 0);   // E is a multiple of four
  row_sync();
#pragma unroll
  for (int u = 0; u < RR; u++) if (rin[u]) atomicAdd(&bins[(int)floorf(rp[u].x) + (int)nb], 1);
  for (int i = r0 + RR * 64 + lane; i < r1; i += 64) atomicAdd(&bins[(int)floorf(tpos[i].x) + (int)nb], 1);
  row_sync();
  const int chunk = (E + 63) / 64, j0 = min(lane * chunk, E), j1 = min(j0 + chunk, E);
  int sum = 0;
  for (int j = j0; j < j1; j++) sum += bins[j];
  const int inc = wave_scan_incl_i(sum);            // (DPP: six dependent LDS round trips as __shfl_up)
  int run = r0 + inc - sum;
  for (int j = j0; j < j1; j++) { const int t = bins[j]; bins[j] = run; run += t; }
  row_sync();
  if (cc.crow) {
    int *o = cc.crow + (int64_t)row * cc.w;
    const int nbi = (int)nb, ntab = cc.ncn + 2 + 2 * cc.T;
    for (int e = lane; e < ntab; e += 64) {
      int x;
      if (e < cc.ncn + 2) x = cc.ms * e - cc.ms / 2 + nbi;
      else { const int q = e - (cc.ncn + 2); x = (q >> 1) * cc.pt + cc.lo + ((q & 1) ? cc.fb : 0); }
      o[e] = x < E ? bins[x] : r1;
    }
  } else {
    int *csr = cs + (int64_t)row * E;
    for (int j = lane; j < E; j += 64) csr[j] = bins[j];
    if (row == nrows - 1 && lane == 0) csr[E] = r1;
  }
  // NGP deposit straight from the row histogram (particle_mesh_threaded.f90:131-151): rho(cell) = mass_p added
  // count(cell) times, cells outside the chain window [4, nf-4) zero (:120-121).  Every row of every tile is
  // the image of exactly one extended row, so each rho row is written once, here, while its counts are in
  // LDS; records whose xv + offset_tile rounds into the next cell are moved afterwards by k_ngp_fixup.
  if (dep.rho) {
    const int cz = row / E, cy = row - cz * E, nbi = (int)nb;
    const float m2 = dep.mass_p + dep.mass_p, m3 = m2 + dep.mass_p;
    float part = 0.f;
    int icount = 0, cmax8 = 0;
    for (int tz = max(0, (cz - dep.nf + dep.pt) / dep.pt); tz < dep.T && tz * dep.pt <= cz; tz++)
      for (int ty = max(0, (cy - dep.nf + dep.pt) / dep.pt); ty < dep.T && ty * dep.pt <= cy; ty++) {
        const int k = cz - tz * dep.pt, j = cy - ty * dep.pt;
        if (k >= dep.nf || j >= dep.nf) continue;
        const bool row_in = (j >= 4 && j < dep.nf - 4 && k >= 4 && k < dep.nf - 4);
        const bool row_int = (j >= nbi && j < dep.nf - nbi && k >= nbi && k < dep.nf - nbi);
        for (int tx = 0; tx < dep.T; tx++) {
          if (dep.rho8) {
            // counts: one byte per cell, four cells per lane and trip; the interior mass as count x mass_p in double at the end (what the
            // selection of partial sums below costs per cell -- half of this kernel's instructions -- is spent once, in the x pass's table)
            unsigned char *out8 = dep.rho8 + ((((int64_t)(tz * dep.T + ty) * dep.T + tx) * dep.nf + k) * dep.nf + j) * dep.rp;
            for (int i4 = lane * 4; i4 < dep.rp; i4 += 256) {
              unsigned pk = 0u;
              if (row_in) {
                const int c = tx * dep.pt + i4;
                int b[5];
                if (c + 4 <= E) { const int4 q = *reinterpret_cast<const int4 *>(bins + c); b[0] = q.x; b[1] = q.y; b[2] = q.z; b[3] = q.w; b[4] = c + 4 < E ? bins[c + 4] : r1; }
                else {
#pragma unroll
                  for (int u = 0; u < 5; u++) b[u] = c + u < E ? bins[c + u] : r1;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                  const int i = i4 + u;
                  const int cnt = (i >= 4 && i < dep.nf - 4) ? b[u + 1] - b[u] : 0;
                  cmax8 = max(cmax8, cnt);
                  if (row_int && i >= nbi && i < dep.nf - nbi) icount += cnt;           // :167-173
                  pk |= (unsigned)min(cnt, 255) << (8 * u);
                }
              }
              *reinterpret_cast<unsigned *>(out8 + i4) = pk;
            }
            continue;
          }
          float *out = dep.rho + ((((int64_t)(tz * dep.T + ty) * dep.T + tx) * dep.nf + k) * dep.nf + j) * dep.rp;
          // four cells per lane and trip (the kernel is bound by its instruction count: 650 vector instructions per row, half of
          // them here when every lane wrote one cell per trip); rp, pt and E are multiples of four
          for (int i4 = lane * 4; i4 < dep.rp; i4 += 256) {
            float rr[4] = {0.f, 0.f, 0.f, 0.f};
            if (row_in) {
              const int c = tx * dep.pt + i4;
              int b[5];
              if (c + 4 <= E) { const int4 q = *reinterpret_cast<const int4 *>(bins + c); b[0] = q.x; b[1] = q.y; b[2] = q.z; b[3] = q.w; b[4] = c + 4 < E ? bins[c + 4] : r1; }
              else {
#pragma unroll
                for (int u = 0; u < 5; u++) b[u] = c + u < E ? bins[c + u] : r1;
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int i = i4 + u;
                if (i >= 4 && i < dep.nf - 4) {
                  const int cnt = b[u + 1] - b[u];
                  cmax8 = max(cmax8, cnt);
                  // :148, same partial sums: 0, m, m + m, (m + m) + m by selection (nearly every cell), the loop beyond three
                  float r = cnt >= 3 ? m3 : (cnt == 2 ? m2 : (cnt == 1 ? dep.mass_p : 0.f));
                  for (int q = 3; q < cnt; q++) r = r + dep.mass_p;
                  rr[u] = r;
                  if (row_int && i >= nbi && i < dep.nf - nbi) part += r;            // :167-173
                }
              }
            }
            *reinterpret_cast<float4 *>(out + i4) = make_float4(rr[0], rr[1], rr[2], rr[3]);
          }
        }
      }
    if (dep.rho8) {
      icount = wave_scan_incl_i(icount); cmax8 = wave_max_nonneg_to_last_i(cmax8);   // lane 63: the row's sum / maximum
      if (lane == 63 && icount != 0 && dep.sum_interior) atomicAdd(dep.sum_interior + p3m_slot() * 8, (double)icount * (double)dep.mass_p);
      if (lane == 63 && cmax8 >= 64) p3m_atomic_max_nonneg(dep.cmax + p3m_slot() * 16, (float)cmax8);   // (small counts are not worth an atomic: the host only asks "below 128?")
    } else {
      if (dep.sum_interior) {
        for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
        if (lane == 0 && part != 0.f) atomicAdd(dep.sum_interior + p3m_slot() * 8, (double)part);
      }
      if (dep.cmax && __any(cmax8 >= 64)) {
        for (int o = 32; o > 0; o >>= 1) cmax8 = max(cmax8, __shfl_down(cmax8, o, 64));
        if (lane == 0) p3m_atomic_max_nonneg(dep.cmax + p3m_slot() * 16, (float)cmax8);
      }
    }
  }
  // records with a coordinate within 2^-10 below a cell face: only these can be moved into the next
  // cell by the rounding of xv + offset_tile (fine_mesh.hip, count-based NGP deposit fix-up)
  const float thr = 1.0f - 0.0009765625f;
  auto place = [&](const float4 &p) {
    const int s = atomicAdd(&bins[(int)floorf(p.x) + (int)nb], 1);
    spos[s] = p;
    if ((p.x - floorf(p.x) >= thr) || (p.y - floorf(p.y) >= thr) || (p.z - floorf(p.z) >= thr)) {
      const int slot = row & (P3M_CAND_SLOTS - 1);            // one of 64 lists (p3m_internal.h)
      const int k = atomicAdd(&cand_cnt[slot * 16], 1);
      if (k < cand_seg) cand[(int64_t)slot * cand_seg + k] = s; else cand_cnt[16 * P3M_CAND_SLOTS] = 1;
    }
  };
#pragma unroll
  for (int u = 0; u < RR; u++) if (rin[u]) place(rp[u]);
  for (int i = r0 + RR * 64 + lane; i < r1; i += 64) place(tpos[i]);
  // records that left the chaining mesh were dropped (link_list.f90:26-53): the slots behind the last row's records get a position no
  // range test accepts, so everything downstream may run over n_cur records without the host knowing the deleted count
  if (row == nrows - 1) for (int i = r1 + lane; i < n_cur; i += 64) spos[i] = make_float4(-1.0e30f, -1.0e30f, -1.0e30f, 0.f);
}

// cell_end from the sorted records (after a sort that wrote only the compact table): one wavefront per row, as k_row_sort
__global__ __launch_bounds__(64) void k_cells_from_sorted(const float4 *__restrict__ spos, const int *__restrict__ rs, int E, int nrows, float nb, int *__restrict__ cs) {
  extern __shared__ int bins[];
  const int row = blockIdx.x, lane = threadIdx.x;
  const int r0 = rs[row], r1 = rs[row + 1];
  for (int j = lane; j < E; j += 64) bins[j] = 0;
  __syncthreads();
  for (int i = r0 + lane; i < r1; i += 64) atomicAdd(&bins[(int)floorf(spos[i].x) + (int)nb], 1);
  __syncthreads();
  const int chunk = (E + 63) / 64, j0 = min(lane * chunk, E), j1 = min(j0 + chunk, E);
  int sum = 0;
  for (int j = j0; j < j1; j++) sum += bins[j];
  const int inc = wave_scan_incl_i(sum);            // (DPP: six dependent LDS round trips as __shfl_up)
  int run = r0 + inc - sum;
  int *csr = cs + (int64_t)row * E;
  for (int j = j0; j < j1; j++) { const int t = bins[j]; csr[j] = run; run += t; }
  if (row == nrows - 1 && lane == 0) csr[E] = r1;
}
int particles_full_cells(p3m_ctx *c) {
  if (!c->cells_compact) return P3M_OK;
  const Geometry &g = c->g;
  const int nrows = g.E * g.E;
  hipLaunchKernelGGL(k_cells_from_sorted, dim3(nrows), dim3(64), (size_t)g.E * sizeof(int), c->stream, (const float4 *)c->spos, (const int *)c->row_end, g.E, nrows,
                     (float)g.nb, c->cell_end);
  HIP_TRY(hipGetLastError());
  c->cells_compact = false;
  return P3M_OK;
}

// single rank: all ghost images in one kernel; leaves c->np_all = records incl. ghosts (unsorted)
int particles_pass_self(p3m_ctx *c) {
  P3M_TRY(particles_resolve(c));
  const Geometry &g = c->g;
  int *cnt = c->d_counters;  // [0] image count, [3] overflow, [4] deleted, [5] candidates (the last two belong to the sort / the row histogram)
  HIP_TRY(hipMemsetAsync(cnt, 0, 4 * sizeof(int), c->stream));
  int n_cur = c->np_local;
  if (n_cur > 0) {
    if (c->hist_done)
      hipLaunchKernelGGL(k_make_images<true>, dim3(cdiv(n_cur, PT)), dim3(PT), 0, c->stream, c->pos, c->vel, n_cur, (int)c->cap, (float)g.Nn,
                         (float)g.nb, cnt, cnt + 3, g.E, c->row_end, (c->p.flags & P3M_FLAG_PPINT) ? c->cflag : (unsigned char *)nullptr, g.ms, g.pt);
    else
      hipLaunchKernelGGL(k_make_images<false>, dim3(cdiv(n_cur, PT)), dim3(PT), 0, c->stream, c->pos, c->vel, n_cur, (int)c->cap, (float)g.Nn,
                         (float)g.nb, cnt, cnt + 3, g.E, c->row_end, (unsigned char *)nullptr, g.ms, g.pt);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(c->h_counters, cnt, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->h_counters[3] || (int64_t)n_cur + c->h_counters[0] > c->cap) {
    p3m_set_error("exceeded max_np in pass: %lld > %lld (particle_pass.f90:136-139); raise density_buffer",
                  (long long)n_cur + c->h_counters[0], (long long)c->cap);
    return P3M_ECAPACITY;
  }
  c->np_all = n_cur + c->h_counters[0];
  return P3M_OK;
}

// sort of the c->np_all unsorted records (physical + ghosts) by extended fine cell; deposit_mass >= 0 (whole-step
// entry points, where mass_p is known here) also writes the NGP density of every tile (c->rho_from_sort)
int particles_sort_enqueue(p3m_ctx *c, float deposit_mass) {
  P3M_TRY(particles_resolve(c));
  const Geometry &g = c->g;
  int *cnt = c->d_counters;
  const int n_cur = c->np_all;
  c->np_ghost = n_cur - c->np_local;
  const int nrows = g.E * g.E;
  const bool want_cflag = (c->p.flags & P3M_FLAG_PPINT) != 0;
  const int nblk = cdiv(n_cur, PT * SORT_RPT);
  const bool counted = c->hist_done;   // the rows were counted by the kernels that wrote the arrival arrays (k_compact_drift_hist, ...)
  c->hist_done = false; c->gl_valid = false;
  c->cnt_from_kick = 0;                // spos is rewritten: per-block survivor counts of an earlier kick are stale
  if (!counted) {
    HIP_TRY(hipMemsetAsync(cnt + 4, 0, 2 * sizeof(int), c->stream));
    HIP_TRY(hipMemsetAsync(c->row_end - 3, 0, (size_t)(nrows + 8) * sizeof(int), c->stream));
    if (want_cflag) { const int64_t ec = g.E / g.ms; HIP_TRY(hipMemsetAsync(c->cflag, 0, (size_t)(ec * ec * ec), c->stream)); }
  }
  if (n_cur > 0 && !counted) {
    hipLaunchKernelGGL(k_row_hist, dim3(nblk), dim3(PT), 0, c->stream, (const float4 *)c->pos, n_cur, c->np_local, (float)g.Nn, (float)g.nb, g.E,
                       c->row_end, cnt + 4, want_cflag ? c->cflag : (unsigned char *)nullptr, g.ms, g.pt);
    HIP_TRY(hipGetLastError());
  }
  P3M_TRY(exclusive_scan_i32(c, c->row_end + 1, nrows));
  if (n_cur > 0) {
    hipLaunchKernelGGL(k_row_scatter, dim3(nblk), dim3(PT), 0, c->stream, (const float4 *)c->pos, n_cur, (float)g.Nn, (float)g.nb, g.E, c->row_end,
                       c->tpos);
    HIP_TRY(hipGetLastError());
  }
  RowDep dep{nullptr, nullptr, 0.f, g.T, g.nf, g.pt, 2 * g.px, nullptr, nullptr};
  c->rho_from_sort = false; c->rho_u8 = false;
  if (deposit_mass >= 0.f && (c->p.flags & P3M_FLAG_NGP) && c->tile_batch == g.ntiles) {
    dep.rho = c->rho; dep.sum_interior = c->d_sums; dep.mass_p = deposit_mass; c->rho_from_sort = true;
    dep.cmax = c->d_red + 3 * P3M_RED_SPAN;            // the largest count of a cell, back with the step's maxima (reductions_fold)
    c->cell_max_reported = true;
    // One byte per cell where (a) the forward x pass of this tile size reads bytes, (b) the LAST whole step of these particles saw no cell
    // of 128 records (cell_max_known: nothing is known right after an upload, that step writes floats).  A count beyond 255 in a byte
    // step -- a cell that more than doubled in one step -- saturates and fails the step loudly (particle_mesh_step); P3M_RHO_F32=1:
    // floats always (a test switch)
    static const bool f32 = getenv("P3M_RHO_F32") && getenv("P3M_RHO_F32")[0] == '1';
    if (!f32 && c->cell_max_known && c->cell_max < 128.0f && fft_x_forward_reads_u8(c->plan_f)) { dep.rho8 = reinterpret_cast<unsigned char *>(c->rho); c->rho_u8 = true; c->rho_u8_step = true; }
  }
  // whole-step PM-only NGP calls: nothing downstream reads per-cell offsets, only the compact table (p3m_internal.h)
  RowCompact cc{nullptr, c->crow_w, g.ncn, g.ms, g.T, g.pt, g.nb - 2, g.fb};
  c->cells_compact = dep.rho != nullptr && !(c->p.flags & (P3M_FLAG_PPINT | P3M_FLAG_PP_EXT));
  if (c->cells_compact) cc.crow = c->crow;
  if (!c->step_zeroed) HIP_TRY(hipMemsetAsync(c->cand_cnt, 0, sizeof(int) * (16 * P3M_CAND_SLOTS + 16), c->stream));   // empty candidate lists (whole steps: step_prezero)
  const int sort_grid = cdiv(nrows, P3M_SORT_WPB);
  hipLaunchKernelGGL(k_row_sort, dim3(sort_grid), dim3(64 * P3M_SORT_WPB), (size_t)g.E * sizeof(int) * P3M_SORT_WPB, c->stream, (const float4 *)c->tpos,
                     (const int *)c->row_end, g.E, nrows, (float)g.nb, c->cell_end, c->spos, c->cand, c->cand_cnt,
                     c->cand_seg, dep, cc, n_cur);
  HIP_TRY(hipGetLastError());
  // (the tail behind the sorted records is padded by k_row_sort's last row: whole steps read the deleted count with the step's other results)
  HIP_TRY(hipMemcpyAsync(c->h_counters, cnt, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->sort_ncur = n_cur;
  return P3M_OK;
}
// the host half: counters of the sort queued by particles_sort_enqueue.  wait = false (whole steps): no host wait here; the
// record count stays at its upper bound (the padded tail is inert) and the deleted count is collected by
// particles_collect_counters once the step's last synchronisation has happened.
int particles_sort_finish(p3m_ctx *c, bool wait) {
  if (wait) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->np_deleted = c->h_counters[4];
    c->np_all = c->sort_ncur - c->np_deleted;  // sorted records
    c->lazy_counters = false;
  } else {
    c->np_deleted = 0;
    c->np_all = c->sort_ncur;
    c->lazy_counters = true;
  }
  return P3M_OK;
}
void particles_collect_counters(p3m_ctx *c) {   // after a stream synchronisation that followed the sort
  if (c->lazy_counters) { c->np_deleted = c->h_counters[4]; c->lazy_counters = false; }
}
int particles_sort(p3m_ctx *c, float deposit_mass) {
  P3M_TRY(particles_sort_enqueue(c, deposit_mass));
  return particles_sort_finish(c, true);
}

int particles_pass_and_sort(p3m_ctx *c) {
  if (c->g.nodes != 1) { p3m_set_error("multi-rank contexts are stepped through a p3m_group (p3m_hip_group_*)"); return P3M_ECOMM; }
  P3M_TRY(particles_pass_self(c));
  return particles_sort(c, -1.f);
}

// ------------------------------------------------------------------ multi-rank ghost pass: all 26 directions in one round
// particle_pass.f90 exchanges +x,-x | -y,+y | +z,-z in sequence, each axis seeing the arrivals of the earlier
// ones, so that a record's ghosts are the Cartesian product of its per-axis image sets (see k_make_images).
// Here every image is sent straight to the rank that owns it: shift m = a + 3b + 9c, a/b/c in {0 none,
// 1 to the + neighbour (x_a >= Nn-nb, image max(x_a-Nn,-nb), :83,:162), 2 to the - neighbour (x_a < nb, image
// min(guard(x_a)+Nn, Nn+nb-eps), :185,:257-265)} for x/y/z.  One pack, one count exchange, one payload exchange
// over up to 7 different peers (all xGMI links at once instead of one link three times), one unpack.
// Two classes per shift m (slot 2m: ghosts, slot 2m+1: migrants).  A GHOST (its image lies in the destination's buffer
// zone) travels as ONE float4 {x,y,z,0}: ghosts only contribute mass and PP partners, are never kicked, and
// delete_particles drops them before the step returns, so the velocity and PID the reference ships along
// (particle_pass.f90 sends xv(1:6) and PID) would be dead weight on xGMI.  A MIGRANT (a record that drifted out of this
// rank's volume: its image is PHYSICAL at the destination and survives there) travels with everything:
// {x,y,z,vx | vy,vz,pid}.  Coordinates are final in both.
#define GSLOTS 54
struct GhostSegs { int64_t off[GSLOTS]; int cap[GSLOTS]; };   // float4 offset / record capacity of each slot's segment in the send and receive buffers
// the (at most one, since Nn >= 2 nb) shift of a coordinate: 0 none, 1 image at the + neighbour, 2 at the - neighbour
__device__ __forceinline__ int axis_shift(float x, float Nn, float nb, float *img) {
  if (x >= Nn - nb) { *img = fmaxf(x - Nn, -nb); return 1; }
  if (x < nb) {
    float xg = x; if (fabsf(xg) < P3M_EPS_F) xg = (xg < 0.0f) ? -P3M_EPS_F : P3M_EPS_F;
    *img = fminf(xg + Nn, Nn + nb - P3M_EPS_F); return 2;
  }
  *img = x; return 0;
}
constexpr int GP_RPT = 8;   // records per thread
// 512 threads: <= 52 reservation atomics per 4096 records.  Measured on the default step (us per launch, 16.7 M records): 256 / 512 /
// 1024 threads 165 / 159 / 161; without the global reservation 153, without the LDS ranks as well 110 -- the kernel is bound by
// its instruction count (two passes over seven image subsets per record), not by the atomics
constexpr int GP_NT = 512;
// LIST: the records to look at are the ghost candidates the compaction listed (blockIdx.y = list, p3m_internal.h): a quarter of the
// records at the bench's geometry; otherwise every record of the arrival array
template <bool LIST>
__global__ __launch_bounds__(GP_NT) void k_ghost_pack(const float4 *__restrict__ pos, const float4 *__restrict__ vel, const int64_t *__restrict__ pid_home,
                                                   int n, float Nn, float nb, float4 *__restrict__ sbuf, GhostSegs S, int *__restrict__ counts, int all_full,
                                                   const int *__restrict__ glist, const int *__restrict__ gl_cnt, int gl_cap) {
  __shared__ int lc[GSLOTS], base[GSLOTS], scap[GSLOTS];
  __shared__ int64_t soff[GSLOTS];   // the segment table by LDS: indexed per lane, the kernel argument was read through global loads, two per image sent
  int nl = n;
  if (LIST) {
    nl = min(gl_cnt[blockIdx.y * 16], gl_cap);
    if ((int)(blockIdx.x * GP_RPT * GP_NT) >= nl) return;   // the whole workgroup: past the list's end
    glist += (int64_t)blockIdx.y * gl_cap;
  }
  if (threadIdx.x < GSLOTS) { lc[threadIdx.x] = 0; scap[threadIdx.x] = S.cap[threadIdx.x]; soff[threadIdx.x] = S.off[threadIdx.x]; }
  __syncthreads();
  // the images of a record are the non-empty subsets t = 1..7 of its shifted axes (bit 0: x, 1: y, 2: z)
  float4 p[GP_RPT]; int rk[GP_RPT][7]; int sh[GP_RPT];   // sh: sx | sy << 2 | sz << 4, 0: nothing to send
  auto slot_of = [&](const float4 &q, int t, int sx, int sy, int sz, float ix, float iy, float iz, float4 *img) {
    const float x = (t & 1) ? ix : q.x, y = (t & 2) ? iy : q.y, z = (t & 4) ? iz : q.z;
    *img = make_float4(x, y, z, 0.f);
    const int m = ((t & 1) ? sx : 0) + 3 * ((t & 2) ? sy : 0) + 9 * ((t & 4) ? sz : 0);
    // survives delete_particles over there; with -DMOVE_GRID_BACK the shift by shake_offset before the deletion can make
    // any ghost physical: then every image travels as a full record
    const bool phys = all_full || (x >= 0.f && x < Nn && y >= 0.f && y < Nn && z >= 0.f && z < Nn);
    return 2 * m + (phys ? 1 : 0);
  };
  int idx[GP_RPT];   // the record's arrival index, -1: none
#pragma unroll
  for (int r = 0; r < GP_RPT; r++) {
    const int e = (blockIdx.x * GP_RPT + r) * GP_NT + threadIdx.x;
    idx[r] = e < nl ? (LIST ? glist[e] : e) : -1;
  }
#pragma unroll
  for (int r = 0; r < GP_RPT; r++) {   // all loads first: eight independent requests in flight per lane
    p[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx[r] >= 0) p[r] = pos[idx[r]];
  }
#pragma unroll
  for (int r = 0; r < GP_RPT; r++) {
    sh[r] = 0;
    if (idx[r] < 0 || !in_hoc_range(p[r], -nb, Nn + nb)) continue;   // dropped by link_list ("PARTICLE DELETED")
    float ix, iy, iz;
    const int sx = axis_shift(p[r].x, Nn, nb, &ix), sy = axis_shift(p[r].y, Nn, nb, &iy), sz = axis_shift(p[r].z, Nn, nb, &iz);
    sh[r] = sx | (sy << 2) | (sz << 4);
    // the x-only image first: every wavefront of cell-sorted records holds +-x face records, but only the rows near a y or z
    // face (18 % of them) have any other subset -- those six are skipped by one branch elsewhere
    float4 img;
    if (sx) rk[r][0] = atomicAdd(&lc[slot_of(p[r], 1, sx, sy, sz, ix, iy, iz, &img)], 1);
    if (sy | sz) {
#pragma unroll
      for (int t = 2; t < 8; t++) {
        const bool ok = (!(t & 1) || sx) && (!(t & 2) || sy) && (!(t & 4) || sz);
        if (ok) rk[r][t - 1] = atomicAdd(&lc[slot_of(p[r], t, sx, sy, sz, ix, iy, iz, &img)], 1);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < GSLOTS) base[threadIdx.x] = lc[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]) : 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < GP_RPT; r++) {
    if (sh[r] == 0) continue;
    const int i = idx[r];
    const int sx = sh[r] & 3, sy = (sh[r] >> 2) & 3, sz = sh[r] >> 4;
    float ix, iy, iz;
    (void)axis_shift(p[r].x, Nn, nb, &ix); (void)axis_shift(p[r].y, Nn, nb, &iy); (void)axis_shift(p[r].z, Nn, nb, &iz);
    auto send = [&](int t) {
      float4 img;
      const int k = slot_of(p[r], t, sx, sy, sz, ix, iy, iz, &img);
      const int s = base[k] + rk[r][t - 1];
      if (s >= scap[k]) return;
      if (k & 1) {   // migrant: the whole record
        const float4 v = vel[i]; const int64_t id = pid_home[rec_index(v)];
        float4 *o = sbuf + soff[k] + 2 * (int64_t)s;
        img.w = v.x;
        o[0] = img; o[1] = make_float4(v.y, v.z, __int_as_float((int)(id & 0xffffffffLL)), __int_as_float((int)(id >> 32)));
      } else sbuf[soff[k] + s] = img;
    };
    if (sx) send(1);
    if (sy | sz) {
#pragma unroll
      for (int t = 2; t < 8; t++) {
        const bool ok = (!(t & 1) || sx) && (!(t & 2) || sy) && (!(t & 4) || sz);
        if (ok) send(t);
      }
    }
  }
}
// appends the received segments: blockIdx.y = slot - 2; dst[k] = first record index of that segment in pos/vel/pid
struct GhostIn { int64_t off[GSLOTS]; int cnt[GSLOTS]; int dst[GSLOTS]; int slot0[GSLOTS]; };   // slot0: first pid_home slot of a migrant segment
template <bool HIST>   // HIST: the arrivals' x-rows are counted on the way (see hist_row)
__global__ __launch_bounds__(PT) void k_ghost_unpack(const float4 *__restrict__ rbuf, GhostIn T, float4 *__restrict__ pos, float4 *__restrict__ vel,
                                                     int64_t *__restrict__ pid_home, float Nn, float nb, int E, int *__restrict__ rs,
                                                     unsigned char *__restrict__ cflag, int ms, int pt) {   // cflag: see flag_displaced (PPINT runs)
  __shared__ int key[HIST ? SORT_HB : 1], val[HIST ? SORT_HB : 1];
  if (HIST) { for (int e = threadIdx.x; e < SORT_HB; e += PT) { key[e] = -1; val[e] = 0; } __syncthreads(); }
  const int k = blockIdx.y + 2, n = T.cnt[k];
  for (int base = blockIdx.x * PT; base < n; base += gridDim.x * PT) {   // uniform trip count: hist_row is a wavefront operation
    const int i = base + threadIdx.x;
    int row = -1;
    if (i < n) {
      float4 q;
      if (k & 1) {
        const float4 *r = rbuf + (int64_t)T.off[k] + 2 * (int64_t)i;
        const float4 r0 = r[0], r1 = r[1];
        const int o = T.dst[k] + i;
        q = make_float4(r0.x, r0.y, r0.z, 0.f);
        pos[o] = q; vel[o] = with_index(r0.w, r1.x, r1.y, T.slot0[k] + i);   // a migrant settles here: its PID gets a slot of this rank's pid_home
        pid_home[T.slot0[k] + i] = (int64_t)(unsigned int)__float_as_int(r1.z) | ((int64_t)__float_as_int(r1.w) << 32);
      } else {
        // ghosts: the velocity / PID slots stay unwritten -- nothing reads them before delete_particles drops the record
        q = rbuf[(int64_t)T.off[k] + i];
        pos[T.dst[k] + i] = q;
      }
      if (HIST && in_hoc_range(q, -nb, Nn + nb)) {
        row = ((int)floorf(q.z) + (int)nb) * E + (int)floorf(q.y) + (int)nb;
        // a migrant is physical here: the flag k_row_hist would have set for it (only the fused histogram skips that kernel)
        if (cflag && q.x >= 0.f && q.x < Nn && q.y >= 0.f && q.y < Nn && q.z >= 0.f && q.z < Nn) flag_displaced(q, nb, E, ms, pt, cflag);
      }
    }
    if (HIST) hist_row(key, val, rs, row);
  }
  if (HIST) { __syncthreads(); hist_flush(key, val, rs); }
}
int particles_ghost_pack(p3m_ctx *c, float4 *sbuf, const int64_t *seg_off, const int *seg_cap, int *d_counts) {
  P3M_TRY(particles_resolve(c));
  if (c->np_local == 0) return P3M_OK;
  GhostSegs S; for (int k = 0; k < GSLOTS; k++) { S.off[k] = seg_off[k]; S.cap[k] = seg_cap[k]; }
  const int all_full = (c->p.flags & P3M_FLAG_MOVE_GRID_BACK) ? 1 : 0;
  if (c->gl_valid) {
    // a list holds at most the records of its blocks of the compaction pass (c->gl_longest)
    hipLaunchKernelGGL(k_ghost_pack<true>, dim3((unsigned)cdiv(std::min<int64_t>(c->gl_longest, c->gl_cap), GP_NT * GP_RPT), P3M_GL_SLOTS), dim3(GP_NT), 0, c->stream,
                       (const float4 *)c->pos, (const float4 *)c->vel, (const int64_t *)c->pid_home, c->np_local, (float)c->g.Nn, (float)c->g.nb, sbuf, S, d_counts,
                       all_full, reinterpret_cast<const int *>(c->tpos), (const int *)c->gl_cnt, c->gl_cap);
  } else
    hipLaunchKernelGGL(k_ghost_pack<false>, dim3(cdiv(c->np_local, GP_NT * GP_RPT)), dim3(GP_NT), 0, c->stream, (const float4 *)c->pos, (const float4 *)c->vel,
                       (const int64_t *)c->pid_home, c->np_local, (float)c->g.Nn, (float)c->g.nb, sbuf, S, d_counts, all_full, (const int *)nullptr, (const int *)nullptr, 0);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
// pid_home is full of holes (records that left this rank keep their slot): the physical records take slots 0..np_local-1 again
__global__ __launch_bounds__(PT) void k_pid_repack(float4 *__restrict__ vel, const int64_t *__restrict__ pid_home, int64_t *__restrict__ tmp, int n) {
  const int i = blockIdx.x * PT + threadIdx.x;
  if (i >= n) return;
  float4 v = vel[i];
  tmp[i] = pid_home[rec_index(v)];
  v.w = __int_as_float(i);
  vel[i] = v;
}
static int pid_repack(p3m_ctx *c) {
  const int n = c->np_local;
  if (n > 0) {
    int64_t *tmp = reinterpret_cast<int64_t *>(c->spos);   // free between the compaction and the sort (16 B per record of room)
    hipLaunchKernelGGL(k_pid_repack, dim3(cdiv(n, PT)), dim3(PT), 0, c->stream, c->vel, (const int64_t *)c->pid_home, tmp, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->pid_home, tmp, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  }
  c->n_home = n;
  return P3M_OK;
}
int particles_ghost_unpack(p3m_ctx *c, const float4 *rbuf, const int64_t *seg_off, const int *cnt, int base) {
  GhostIn T; int mx = 0, run = base; int64_t nmig = 0;
  for (int k = 2; k < GSLOTS; k += 1) if (k & 1) nmig += cnt[k];
  // pid_home gains a slot per arriving migrant and never frees the slots of the records that left: repacked when the holes exceed a
  // quarter of the records (every few dozen steps of a run; a pass over np_local records), and in any case before the array is full
  // (cap >= np_local + arrivals was checked by the caller)
  if ((int64_t)c->n_home + nmig > std::min<int64_t>(c->cap, (int64_t)c->np_local + c->np_local / 4 + nmig + 64)) P3M_TRY(pid_repack(c));
  int slot = c->n_home;
  for (int k = 0; k < GSLOTS; k++) {
    T.off[k] = seg_off[k]; T.cnt[k] = k >= 2 ? cnt[k] : 0; T.dst[k] = run; run += T.cnt[k]; mx = std::max(mx, T.cnt[k]);
    T.slot0[k] = slot; if (k & 1) slot += T.cnt[k];
  }
  if (mx == 0) return P3M_OK;
  c->n_home = slot;
  unsigned char *cflag = (c->p.flags & P3M_FLAG_PPINT) ? c->cflag : nullptr;
  if (c->hist_done)
    hipLaunchKernelGGL(k_ghost_unpack<true>, dim3(std::min(1024, cdiv(mx, PT)), GSLOTS - 2), dim3(PT), 0, c->stream, rbuf, T, c->pos, c->vel, c->pid_home, (float)c->g.Nn,
                       (float)c->g.nb, c->g.E, c->row_end, cflag, c->g.ms, c->g.pt);
  else
    hipLaunchKernelGGL(k_ghost_unpack<false>, dim3(std::min(1024, cdiv(mx, PT)), GSLOTS - 2), dim3(PT), 0, c->stream, rbuf, T, c->pos, c->vel, c->pid_home, (float)c->g.Nn,
                       (float)c->g.nb, c->g.E, c->row_end, cflag, c->g.ms, c->g.pt);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// ------------------------------------------------------------------ delete_particles.f90:17-47 (+ move_grid_back.f90:17-24)
// move_grid_back (xv -= shake_offset) runs BEFORE delete_particles (particle_mesh_threaded.f90:716-720),
// so the range test is applied to the shifted-back positions.
// Survivors of delete_particles, counted per block of PT consecutive sorted records (SORT_RPT such blocks per workgroup,
// the chunking of k_compact_drift_hist).  The exclusive scan of these counts gives every block its first destination;
// the rank of a record inside its block is a ballot away (block_rank), so no per-record flag / offset array is written,
// scanned or read (it used to be 4 B written + scanned + read per record).
__device__ __forceinline__ bool survives(float4 p, float Nn, float mx, float my, float mz) {
  p.x -= mx; p.y -= my; p.z -= mz;
  return p.x >= 0.0f && p.x < Nn && p.y >= 0.0f && p.y < Nn && p.z >= 0.0f && p.z < Nn;
}
__global__ __launch_bounds__(PT) void k_count_physical(const float4 *__restrict__ spos, int n, float Nn, int *__restrict__ cnt, float mx, float my, float mz) {
  __shared__ int wc[SORT_RPT][PT / 64];
  float4 pl[SORT_RPT];
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    pl[u] = make_float4(-1.f, -1.f, -1.f, 0.f);
    if (i < n) pl[u] = spos[i];
  }
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    const unsigned long long m = __ballot(i < n && survives(pl[u], Nn, mx, my, mz));
    if ((threadIdx.x & 63) == 0) wc[u][threadIdx.x >> 6] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < SORT_RPT) {
    const int blk = blockIdx.x * SORT_RPT + threadIdx.x;
    if ((int64_t)blk * PT < n) { int t = 0; for (int w = 0; w < PT / 64; w++) t += wc[threadIdx.x][w]; cnt[blk] = t; }
  }
}
// destination of a surviving record: first destination of its block + survivors before it in the block (every thread calls)
__device__ __forceinline__ int block_rank(bool keep, int *wcnt) {   // wcnt: PT/64 ints of LDS, not in use by anybody else until the next barrier
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wcnt[wv] = __popcll(m);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wv; w++) base += wcnt[w];
  return base + __popcll(m & ((1ull << lane) - 1ull));
}
// DRIFT: the next step's update_position (update_position.f90:68-76) rides on the copy, applied to the very value
// k_compact alone would have stored
// vel_old: the arrival-order velocities of the step that ends here (read through the index in spos[s].w); pos / vel: the next arrival arrays
template <bool DRIFT>
__global__ __launch_bounds__(PT) void k_compact(const float4 *__restrict__ spos, const float4 *__restrict__ vel_old,
                                                const int *__restrict__ offs, int n, float Nn, float4 *__restrict__ pos, float4 *__restrict__ vel,
                                                float mx, float my, float mz, float hs, float ox, float oy, float oz, int use_off) {
  __shared__ int wcnt[PT / 64];
  const int i = blockIdx.x * PT + threadIdx.x;
  float4 p = make_float4(-1.f, -1.f, -1.f, 0.f);
  if (i < n) p = spos[i];
  const bool keep = i < n && survives(p, Nn, mx, my, mz);
  const int o = offs[blockIdx.x] + block_rank(keep, wcnt);   // offs: first destination of every block of PT records
  if (!keep) return;
  p.x -= mx; p.y -= my; p.z -= mz;
  const int src = rec_index(p);
  const float4 v = vel_old[src];   // its fourth lane is the PID slot: it travels with the velocity
  if (DRIFT) {
    if (use_off) { p.x = p.x + v.x * hs + ox; p.y = p.y + v.y * hs + oy; p.z = p.z + v.z * hs + oz; }  // :71
    else { p.x = p.x + v.x * hs; p.y = p.y + v.y * hs; p.z = p.z + v.z * hs; }                           // :73
  }
  pos[o] = p; vel[o] = v;
}

// k_compact<true> with the row histogram of the sort that follows (k_row_hist) counted on the way: SORT_RPT records per
// thread like k_row_hist, so that a block's 2048 consecutive (cell-sorted) records hit a few dozen rows of the LDS table.
// Records that the drift carries out of the chaining mesh are counted as deleted (link_list.f90:26-53), coarse cells with a
// displaced record are flagged (see k_row_hist).
__global__ __launch_bounds__(PT) void k_compact_drift_hist(const float4 *__restrict__ spos, const float4 *__restrict__ vel_old,
                                                           const int *__restrict__ offs, int n, float Nn, float4 *__restrict__ pos, float4 *__restrict__ vel,
                                                           float mx, float my, float mz, float hs, float ox, float oy, float oz, int use_off,
                                                           float nb, int E, int *__restrict__ rs, int *__restrict__ ndeleted, unsigned char *__restrict__ cflag, int ms, int pt,
                                                           int *__restrict__ glist, int *__restrict__ gl_cnt, int gl_cap) {   // ghost candidates (p3m_internal.h)
  __shared__ int key[SORT_HB], val[SORT_HB];
  __shared__ int wc[SORT_RPT][PT / 64];
  __shared__ int gl[PT * SORT_RPT], gn, gbase;   // this block's candidates, then one reservation in the block's list
  if (threadIdx.x == 0) gn = 0;
  for (int e = threadIdx.x; e < SORT_HB; e += PT) { key[e] = -1; val[e] = 0; }
  float4 pl[SORT_RPT];
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {   // all position loads first
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    pl[u] = make_float4(-1.f, -1.f, -1.f, 0.f);
    if (i < n) pl[u] = spos[i];
  }
  int rk[SORT_RPT];   // survivors before this record in its wavefront; the wavefronts' totals meet in LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    const unsigned long long m = __ballot(i < n && survives(pl[u], Nn, mx, my, mz));
    rk[u] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wc[u][wv] = __popcll(m);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < SORT_RPT; u++) {
    const int i = (blockIdx.x * SORT_RPT + u) * PT + threadIdx.x;
    float4 p = pl[u];
    int row = -1, oc = 0;
    bool cand = false;
    if (i < n && survives(p, Nn, mx, my, mz)) {   // delete_particles.f90:17-47
      p.x -= mx; p.y -= my; p.z -= mz;
      int o = offs[blockIdx.x * SORT_RPT + u] + rk[u];   // offs: first destination of every block of PT records
      for (int w = 0; w < wv; w++) o += wc[u][w];
      const int src = rec_index(p);
      const float4 v = vel_old[src];   // its fourth lane is the PID slot: it travels with the velocity
      if (use_off) { p.x = p.x + v.x * hs + ox; p.y = p.y + v.y * hs + oy; p.z = p.z + v.z * hs + oz; }  // update_position.f90:71
      else { p.x = p.x + v.x * hs; p.y = p.y + v.y * hs; p.z = p.z + v.z * hs; }                           // :73
      pos[o] = p; vel[o] = v;
      if (in_hoc_range(p, -nb, Nn + nb)) {
        row = ((int)floorf(p.z) + (int)nb) * E + (int)floorf(p.y) + (int)nb;
        if (cflag && p.x >= 0.f && p.x < Nn && p.y >= 0.f && p.y < Nn && p.z >= 0.f && p.z < Nn) flag_displaced(p, nb, E, ms, pt, cflag);
        // the records k_ghost_pack would find a shift for (axis_shift != 0 on some axis)
        cand = p.x >= Nn - nb || p.x < nb || p.y >= Nn - nb || p.y < nb || p.z >= Nn - nb || p.z < nb;
        oc = o;
      } else atomicAdd(ndeleted, 1);
    }
    {
      const unsigned long long m = __ballot(cand);
      if (m) {
        int b = 0;
        if (lane == __ffsll((long long)m) - 1) b = atomicAdd(&gn, __popcll(m));
        b = __builtin_amdgcn_readlane(b, __ffsll((long long)m) - 1);   // (a uniform lane: not the LDS round trip of __shfl)
        if (cand) gl[b + __popcll(m & ((1ull << lane) - 1ull))] = oc;
      }
    }
    hist_row(key, val, rs, row);
  }
  __syncthreads();
  hist_flush(key, val, rs);
  // the block's candidates go to one of P3M_GL_SLOTS lists (by block: the lists fill evenly; one global atomic per block)
  const int slot = blockIdx.x & (P3M_GL_SLOTS - 1);
  if (threadIdx.x == 0) gbase = gn ? atomicAdd(&gl_cnt[slot * 16], gn) : 0;
  __syncthreads();
  for (int e = threadIdx.x; e < gn; e += PT) if (gbase + e < gl_cap) glist[(int64_t)slot * gl_cap + gbase + e] = gl[e];
}

// delete_particles: the survivors are counted now (the step's np_local), but the copy back to the arrival arrays is
// deferred: the next step's drift does it in the same pass (particles_drift), anything else that reads the arrival
// arrays first calls particles_resolve.
int particles_finalize_enqueue(p3m_ctx *c, const float *move_back) {
  const int n = c->np_all;
  c->pending_compact = false;
  c->pend_n = n;
  c->finalize_queued = true;
  if (n == 0) { c->np_local = 0; return P3M_OK; }
  float mx = 0, my = 0, mz = 0;
  if (move_back) { mx = move_back[0]; my = move_back[1]; mz = move_back[2]; }
  const int nblk = cdiv(n, PT);   // c->flags: survivors per block of PT sorted records, then (scanned) every block's first destination
  const bool counted = c->cnt_from_kick == n && mx == 0.f && my == 0.f && mz == 0.f;   // the NGP kick of this step counted them on its way (fine_max_and_kick)
  c->cnt_from_kick = 0;
  if (!counted) hipLaunchKernelGGL(k_count_physical, dim3(cdiv(nblk, SORT_RPT)), dim3(PT), 0, c->stream, (const float4 *)c->spos, n, (float)c->g.Nn, c->flags, mx, my, mz);
  HIP_TRY(hipGetLastError());
  P3M_TRY(exclusive_scan_i32(c, c->flags, nblk));
  HIP_TRY(hipMemcpyAsync(c->h_counters, c->flags + nblk, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->pend_mb[0] = mx; c->pend_mb[1] = my; c->pend_mb[2] = mz;
  return P3M_OK;
}
int particles_finalize_finish(p3m_ctx *c, bool wait) {   // wait = false: the caller synchronised the stream after particles_finalize_enqueue
  c->finalize_queued = false;
  if (c->pend_n == 0) { particles_collect_counters(c); return P3M_OK; }
  if (wait) HIP_TRY(hipStreamSynchronize(c->stream));
  particles_collect_counters(c);
  c->np_local = c->h_counters[0];
  c->pending_compact = true;
  return P3M_OK;
}
int particles_finalize(p3m_ctx *c, const float *move_back) {
  P3M_TRY(particles_finalize_enqueue(c, move_back));
  return particles_finalize_finish(c, true);
}
// the first launch of a kernel pays for loading it: the deferred-compaction kernels are first needed in the SECOND step,
// where that shows up as a host stall in the middle of a timed run; ask for their attributes at context creation instead
int particles_preload() {
  hipFuncAttributes fa;
  HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_compact<true>)));
  HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_compact<false>)));
  return P3M_OK;
}
// zeroes what the row histogram of the next sort accumulates into (k_row_hist's preamble) -- for the kernels that count on the way
static int particles_hist_begin(p3m_ctx *c) {
  const Geometry &g = c->g;
  const int nrows = g.E * g.E;
  P3M_TRY(zero_add(c, c->d_counters + 4, 2 * sizeof(int)));
  P3M_TRY(zero_add(c, c->gl_cnt, 16 * P3M_GL_SLOTS * sizeof(int)));
  P3M_TRY(zero_add(c, c->row_end - 3, (size_t)(nrows + 8) * sizeof(int)));
  if (c->p.flags & P3M_FLAG_PPINT) { const int64_t ec = g.E / g.ms; P3M_TRY(zero_add(c, c->cflag, (size_t)(ec * ec * ec))); }   // (allocated with 16 bytes of slack)
  return zero_flush(c);   // one launch (p3m_internal.h, ZeroList)
}
int particles_compact(p3m_ctx *c, bool drift, float dt, float dt_old, const float *offset) {
  if (!c->pending_compact) return P3M_OK;
  c->pending_compact = false;
  const int n = c->pend_n;
  const float hs = 0.5f * (dt + dt_old);
  if (drift) {
    // the sort follows: count its x-rows here and in the kernels that append the ghosts (c->hist_done)
    P3M_TRY(particles_hist_begin(c));
    const Geometry &g = c->g;
    hipLaunchKernelGGL(k_compact_drift_hist, dim3(cdiv(n, PT * SORT_RPT)), dim3(PT), 0, c->stream, (const float4 *)c->spos, (const float4 *)c->vel,
                       (const int *)c->flags, n, (float)g.Nn, c->tpos, c->vel_alt, c->pend_mb[0], c->pend_mb[1], c->pend_mb[2], hs,
                       offset ? offset[0] : 0.f, offset ? offset[1] : 0.f, offset ? offset[2] : 0.f, offset ? 1 : 0, (float)g.nb, g.E, c->row_end, c->d_counters + 4,
                       (c->p.flags & P3M_FLAG_PPINT) ? c->cflag : (unsigned char *)nullptr, g.ms, g.pt,
                       reinterpret_cast<int *>(c->pos), c->gl_cnt, c->gl_cap);   // c->pos: the arrival buffer this pass retires (tpos after the swap below)
    c->hist_done = true; c->gl_valid = c->cap >= 4096;   // (a list's capacity, cap / 2, holds any list's blocks from two blocks' worth of records on)
    c->gl_longest = (int64_t)cdiv(cdiv(n, PT * SORT_RPT), P3M_GL_SLOTS) * PT * SORT_RPT;   // every P3M_GL_SLOTS-th block of this pass (n counts the ghosts it drops, too)
  } else if (drift)
    hipLaunchKernelGGL(k_compact<true>, dim3(cdiv(n, PT)), dim3(PT), 0, c->stream, (const float4 *)c->spos, (const float4 *)c->vel,
                       (const int *)c->flags, n, (float)c->g.Nn, c->tpos, c->vel_alt, c->pend_mb[0], c->pend_mb[1], c->pend_mb[2], hs,
                       offset ? offset[0] : 0.f, offset ? offset[1] : 0.f, offset ? offset[2] : 0.f, offset ? 1 : 0);
  else
    hipLaunchKernelGGL(k_compact<false>, dim3(cdiv(n, PT)), dim3(PT), 0, c->stream, (const float4 *)c->spos, (const float4 *)c->vel,
                       (const int *)c->flags, n, (float)c->g.Nn, c->tpos, c->vel_alt, c->pend_mb[0], c->pend_mb[1], c->pend_mb[2], 0.f, 0.f, 0.f, 0.f, 0);
  HIP_TRY(hipGetLastError());
  // the new arrival arrays were written into the buffers of the sort's intermediate; the old ones take over that role
  std::swap(c->pos, c->tpos); std::swap(c->vel, c->vel_alt);
  return P3M_OK;
}
int particles_resolve(p3m_ctx *c) { return particles_compact(c, false, 0.f, 0.f, nullptr); }
