// group.hip -- multi-rank gravity step: the reference's nodes_dim^3 cubic sub-volumes ("logical
// ranks", mpi_initialization.f90:42-76) distributed over the GPUs of one node.  One process drives
// one GPU and owns a contiguous block of logical ranks (1, 2, 4 or 8 of the 8 when nodes_dim = 2);
// exchanges between ranks of the same process are device-to-device copies (one launch per exchange),
// exchanges between processes are RCCL send/recv over xGMI or the host-callback transport.  Replaces
// the MPI traffic of
//   particle_pass.f90            -> ghost_pass()        all 26 shifts in one round, counts first
//   fftw3ds.f90 pack/unpack_slab -> cube_to_slab() / slab_to_cube()  (all-to-all inside a z-layer)
//   rfftwnd_f77_mpi (FFTW-MPI)   -> slab FFT: local x,y passes, ONE global all-to-all transpose per
//                                   transform written directly by the y/z pass kernels, local z pass
//   coarse_force_buffer.f90      -> force_halo()        3 axes x 2 directions
//   mpi_reduce/bcast of dt limits-> reduce_step_out()   one max all-reduce + one sum all-reduce
// k-space stays in "transposed order" (each rank owns a ky-slab with all kz), which saves the second
// transpose FFTW's NORMAL_ORDER pays; kern_c is built through the same pipeline and is therefore
// stored consistently.
#include "p3m_internal.h"
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <rccl/rccl.h>
#include "fft_core.h"

int fft_x_forward_rows(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int64_t rows, int rpp = 0);
int fft_x_inverse(p3m_ctx *c, const FftPlan &pl, const float *src, float *out, int batch, int mode, float *box, int fb, int lo, int ntile, int64_t bcs, int rpp = 0);
bool fft_x_has_cubes(const FftPlan &pl, const RowGeom &q);
int fft_x_forward_cubes(p3m_ctx *c, const FftPlan &pl, const float *cubes, float *dst, const RowGeom &q);
int fft_x_inverse_cubes(p3m_ctx *c, const FftPlan &pl, const float *src, float *fc, const RowGeom &q, const RankPtrs &red);
int fft_slab_y_fwd(p3m_ctx *c, const FftPlan &pl, const float *ly, float *send, int planes, int batch = 1, bool direct = false);
int fft_slab_z_fwd(p3m_ctx *c, const FftPlan &pl, const float *src, float *lz, int planes, int seg, int batch = 1);
bool fft_has_segmented(const FftPlan &pl);
int fft_slab_z_inv3(p3m_ctx *c, const FftPlan &pl, const float *lz, float *send3, const float *kern3, int planes, int64_t kern_comp_stride,
                    int64_t send_comp_stride, int batch = 1, int64_t kern_batch_stride = 0, bool direct = false);
int fft_slab_y_inv(p3m_ctx *c, const FftPlan &pl, const float *src, float *ly3, int planes, int batch, int seg);

#define NCCL_TRY(expr)                                                                             \
  do {                                                                                             \
    ncclResult_t _r = (expr);                                                                      \
    if (_r != ncclSuccess) { p3m_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); return P3M_ECOMM; } \
  } while (0)

struct CoarseDist {            // per local logical rank; slabs: nxb = nd^2, rpp = nc, ncl = nchunk; pencils: nxb = nd, rpp = ncn, ncl = nchunk/nd
  float *blocks_in = nullptr;  // [nxb][s][ncn][ncn]       cube -> slab / pencil arrivals
  float *rows = nullptr;       // [3][s][rpp][2*px]        real rows of the local planes
  float *ly = nullptr;         // [3][s][nchunk][rpp][16]  complex, LY of the local planes (pencils: also [3][s][ncl][nc][16] after the x<->y transpose)
  float *send = nullptr;       // [3][nc][ncl][s][16]      complex, all-to-all send layout
  float *recv = nullptr;       // same
  float *lz = nullptr;         // [s][ncl][nc][16]         complex rho-hat, own ky slab (pencils: of the own kx chunks)
  float *kern = nullptr;       // [3][s*ncl*nc*16]         Im K_c, same order as lz
  float *blocks_out = nullptr; // [nxb][3][s][ncn][ncn]    slab / pencil -> cube departures
  float *blocks_back = nullptr;// [nxb][3][s][ncn][ncn]    arrivals at the cube
  float *halo_s[2] = {nullptr, nullptr}, *halo_r[2] = {nullptr, nullptr};  // [3][(ncn+2)^2]
  float4 *sb = nullptr, *rb = nullptr;   // ghost (16 B) and migrant (32 B) records, 52 segments (G->seg_off, in float4 units)
  int *d_cnt = nullptr;        // [0..53] own send counts per slot, [64..117] counts announced by the neighbours
};

struct p3m_group {
  p3m_params base{};
  int proc = 0, nprocs = 1, nodes = 1, nd = 1, device = 0;
  hipStream_t stream = nullptr;
  // the coarse slab transform with its all-to-all exchanges (many small kernels, launch- and
  // latency-bound) runs on `stream2` underneath the fine-mesh force sweeps of the local ranks, which need the coarse force
  // only when they kick
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_dep = nullptr, ev_cf = nullptr;
  // Several local ranks on one device (the one-GPU bench line: eight): between the ghost pass and the end of the step a rank's kernels depend
  // on no other rank's -- except for the coarse force, which waits for every rank's coarse density and is waited for by every rank's kick.
  // Each local rank queues that stretch on a stream of its own (`rstream`), so that one rank's latency-bound particle kernels, launch gaps
  // and kernel tails run underneath another rank's bandwidth-bound passes; the step forks behind the ghost pass (`ev_fork`) and joins before
  // its one host wait (`rev_done`).  OPT-IN (P3M_GROUP_STREAMS=n, n streams): the one-GPU headline runs 30.5 -> 29.9 ms with eight, but kernels of
  // different ranks then share the device and a kernel's rocprofv3 duration is no longer its own (the fused z pass: 735 -> 2090 us average) --
  // the roofline evidence of bench.py and profiles/ is per-kernel, so the default keeps one stream.  Never on with one local rank (one rank per
  // GPU: nothing to overlap) or with phase timing on (the spans are taken on ONE stream).
  std::vector<hipStream_t> rstream; std::vector<hipEvent_t> rev_dep, rev_done; hipEvent_t ev_fork = nullptr; bool multi = false; size_t nrs = 1;   // nrs: streams in use (local rank i queues on rstream[i % nrs]; P3M_GROUP_STREAMS=n)
  std::vector<p3m_ctx *> ctx; std::vector<int> lrank, owner, lidx;
  std::vector<CoarseDist> cd;
  // The coarse arrays of all local ranks are slices of group-wide allocations.  `batched` (slabs whose line length has
  // register-stage FFT kernels): laid out [component][local rank][...], so that every stage of the distributed transform is ONE
  // launch over all local ranks (the rank is the batch index of the FFT kernels) instead of one per rank; otherwise (pencils,
  // Stockham sizes) [local rank][component][...] and the per-rank loops.  cstride: floats between two components of one rank.
  bool batched = false; size_t cstride = 0, rstride = 0;   // rstride: the same for the real rows
  // direct (batched, ONE process holding every rank, no forced RCCL): the three redistributions of the transform are no
  // exchanges at all -- the x pass gathers its rows from the ranks' cubes, the transposing y / z passes store into the peers'
  // receive blocks, the inverse x pass's rows go straight into the owners' force_c (70 GB of device-to-device copies per
  // coarse_force at nc = 1024 otherwise).  P3M_COARSE_COPY=1 keeps the message path (what several processes run).
  bool direct = false;
  float *a_blocks_in = nullptr, *a_rows = nullptr, *a_ly = nullptr, *a_send = nullptr, *a_recv = nullptr, *a_lz = nullptr, *a_kern = nullptr,
        *a_blocks_out = nullptr, *a_blocks_back = nullptr, *a_rho_c = nullptr, *a_force_c = nullptr, *a_halo = nullptr;
  ncclComm_t comm = nullptr; bool force_nccl = false;
  p3m_transport tr{}; bool have_tr = false;
  char *h_stage[2] = {nullptr, nullptr}; size_t stage_cap[2] = {0, 0};   // pinned send / receive staging of the host transport
  FftPlan plan_c; int s = 0, nchunk = 0;
  // P3M_FLAG_PENCIL: x-pencils of (nc, ncn, s = ncn/nd) cells as pack_pencils leaves them (p3dfft_coarse.f90:69-127); the
  // transform then splits the kx chunks over the nd ranks that hold the same z planes (x<->y transpose) and ky over the nd^2
  // ranks that hold the same chunks (y<->z transpose).  plan_l: plan_c with the row pitch of the local chunks (shares the twiddles)
  bool pencil = false; int ncl = 0, rpp = 0, nxb = 0; FftPlan plan_l;
  int64_t seg_off[54] = {0}; int seg_cap[54] = {0}; int64_t seg_total = 0;   // segments by slot (2m ghosts, 2m+1 migrants), offsets in float4 units
  int *h_cnt = nullptr;        // pinned [nlocal*4]
  int *d_gather = nullptr, *h_gather = nullptr, *h_hdr = nullptr;   // ghost pass: [nodes][64] counts of every rank (device, pinned), [nlocal] pinned headers
  float *d_red4 = nullptr; double *d_sum3 = nullptr; float *h_red4 = nullptr; double *h_sum3 = nullptr;
  bool have_k = false;
  p3m_step_out last{};
  PhaseTimer pt;               // per-phase times of the last step; every context of the group points here
};

struct XMsg { int src, dst; const void *sptr; void *rptr; size_t bytes; };

// Host-callback route: the messages to / from each peer process are concatenated (in list order, which is
// the same on both sides) into pinned staging, handed to ONE exchange callback, and copied back up.
static int host_exchange(p3m_group *G, const std::vector<XMsg> &msgs) {
  const int np = G->nprocs;
  std::vector<size_t> sb(np, 0), rb(np, 0), so(np, 0), ro(np, 0);
  for (const XMsg &m : msgs) {
    const int ps = G->owner[m.src], pd = G->owner[m.dst];
    if (m.bytes == 0 || ps == pd) continue;
    if (ps == G->proc) sb[pd] += m.bytes;
    if (pd == G->proc) rb[ps] += m.bytes;
  }
  size_t tot[2] = {0, 0};
  for (int q = 0; q < np; q++) { so[q] = tot[0]; ro[q] = tot[1]; tot[0] += sb[q]; tot[1] += rb[q]; }
  for (int k = 0; k < 2; k++)
    if (tot[k] > G->stage_cap[k]) {
      if (G->h_stage[k]) (void)hipHostFree(G->h_stage[k]);
      G->h_stage[k] = nullptr; G->stage_cap[k] = 0;
      const size_t cap = tot[k] + tot[k] / 4 + 4096;
      if (hipHostMalloc(reinterpret_cast<void **>(&G->h_stage[k]), cap) != hipSuccess) { p3m_set_error("pinned staging of %zu bytes failed", cap); return P3M_ENOMEM; }
      G->stage_cap[k] = cap;
    }
  std::vector<size_t> cur = so;
  for (const XMsg &m : msgs) {
    const int ps = G->owner[m.src], pd = G->owner[m.dst];
    if (m.bytes == 0 || ps == pd || ps != G->proc) continue;
    HIP_TRY(hipMemcpyAsync(G->h_stage[0] + cur[pd], m.sptr, m.bytes, hipMemcpyDeviceToHost, G->stream));
    cur[pd] += m.bytes;
  }
  HIP_TRY(hipStreamSynchronize(G->stream));
  std::vector<int32_t> peer; std::vector<const void *> sp; std::vector<void *> rp; std::vector<int64_t> sn, rn;
  for (int q = 0; q < np; q++)
    if (sb[q] || rb[q]) { peer.push_back(q); sp.push_back(G->h_stage[0] + so[q]); rp.push_back(G->h_stage[1] + ro[q]); sn.push_back((int64_t)sb[q]); rn.push_back((int64_t)rb[q]); }
  if (!peer.empty()) {
    const int rc = G->tr.exchange(G->tr.user, (int32_t)peer.size(), peer.data(), sp.data(), sn.data(), rp.data(), rn.data());
    if (rc) { p3m_set_error("host transport: exchange callback returned %d", rc); return P3M_ECOMM; }
  }
  cur = ro;
  for (const XMsg &m : msgs) {
    const int ps = G->owner[m.src], pd = G->owner[m.dst];
    if (m.bytes == 0 || ps == pd || pd != G->proc) continue;
    HIP_TRY(hipMemcpyAsync(m.rptr, G->h_stage[1] + cur[ps], m.bytes, hipMemcpyHostToDevice, G->stream));
    cur[ps] += m.bytes;
  }
  return P3M_OK;
}

// All device-to-device messages of one exchange in ONE launch (an exchange is dozens of small messages; as
// separate copies each costs ~6 us of dependent-launch latency on the stream).  blockIdx.y = message.
#define XCOPY_MAX 40
struct XCopyBatch { const char *src[XCOPY_MAX]; char *dst[XCOPY_MAX]; unsigned long long bytes[XCOPY_MAX]; };
__global__ __launch_bounds__(256) void k_multi_copy(XCopyBatch b) {
  const char *src = b.src[blockIdx.y]; char *dst = b.dst[blockIdx.y];
  const unsigned long long n = b.bytes[blockIdx.y];
  const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x, stride = (unsigned long long)gridDim.x * 256;
  if ((((unsigned long long)src | (unsigned long long)dst | n) & 15) == 0) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src); float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (unsigned long long i = t; i < (n >> 4); i += stride) d4[i] = s4[i];
  } else if ((((unsigned long long)src | (unsigned long long)dst | n) & 3) == 0) {
    const int *s1 = reinterpret_cast<const int *>(src); int *d1 = reinterpret_cast<int *>(dst);
    for (unsigned long long i = t; i < (n >> 2); i += stride) d1[i] = s1[i];
  } else {
    for (unsigned long long i = t; i < n; i += stride) dst[i] = src[i];
  }
}
static int flush_copies(p3m_group *G, XCopyBatch &b, int &nb, size_t &maxb) {
  if (nb == 0) return P3M_OK;
  const unsigned gx = (unsigned)std::min<size_t>(256, std::max<size_t>(1, (maxb + 16 * 256 * 8 - 1) / (16 * 256 * 8)));
  hipLaunchKernelGGL(k_multi_copy, dim3(gx, nb), dim3(256), 0, G->stream, b);
  HIP_TRY(hipGetLastError());
  nb = 0; maxb = 0;
  return P3M_OK;
}

// every process builds the SAME global message list (same order); pointers are only needed for local ends
static int do_exchange(p3m_group *G, const std::vector<XMsg> &msgs) {
  bool any_remote = false;
  XCopyBatch batch; int nb = 0; size_t maxb = 0;
  for (const XMsg &m : msgs) {
    const bool sl = G->owner[m.src] == G->proc, dl = G->owner[m.dst] == G->proc;
    if (m.bytes == 0) continue;
    if (sl && dl && !G->force_nccl) {
      batch.src[nb] = static_cast<const char *>(m.sptr); batch.dst[nb] = static_cast<char *>(m.rptr); batch.bytes[nb] = m.bytes;
      maxb = std::max(maxb, m.bytes);
      if (++nb == XCOPY_MAX) P3M_TRY(flush_copies(G, batch, nb, maxb));
    } else if (sl || dl) any_remote = true;
  }
  P3M_TRY(flush_copies(G, batch, nb, maxb));
  if (!any_remote) return P3M_OK;
  if (!G->comm && G->have_tr) return host_exchange(G, msgs);
  if (!G->comm) { p3m_set_error("group exchange between processes needs RCCL (p3m_hip_group_comm_init_rccl) or a host transport (p3m_hip_group_set_transport)"); return P3M_ECOMM; }
  NCCL_TRY(ncclGroupStart());
  for (const XMsg &m : msgs) {
    if (m.bytes == 0) continue;
    const bool sl = G->owner[m.src] == G->proc, dl = G->owner[m.dst] == G->proc;
    if (sl && dl && !G->force_nccl) continue;
    if (sl) NCCL_TRY(ncclSend(m.sptr, m.bytes, ncclChar, G->owner[m.dst], G->comm, G->stream));
    if (dl) NCCL_TRY(ncclRecv(m.rptr, m.bytes, ncclChar, G->owner[m.src], G->comm, G->stream));
  }
  NCCL_TRY(ncclGroupEnd());
  return P3M_OK;
}

template <typename T> static int galloc(T **p, size_t n) {
  *p = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) { p3m_set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e)); return P3M_ENOMEM; }
  return P3M_OK;
}
template <typename T> static void gfree(T *&p) { if (p) (void)hipFree(p); p = nullptr; }

// ================================================================== lifecycle
extern "C" void p3m_hip_group_destroy(p3m_group *G) {
  if (!G) return;
  (void)hipSetDevice(G->device);
  if (G->stream) (void)hipStreamSynchronize(G->stream);
  for (CoarseDist &d : G->cd) { gfree(d.sb); gfree(d.rb); gfree(d.d_cnt); }
  gfree(G->a_blocks_in); gfree(G->a_rows); gfree(G->a_ly); gfree(G->a_send); gfree(G->a_recv); gfree(G->a_lz); gfree(G->a_kern);
  gfree(G->a_blocks_out); gfree(G->a_blocks_back); gfree(G->a_halo);
  for (p3m_ctx *c : G->ctx) {
    if (!c) continue;
    if (G->a_rho_c) c->rho_c = nullptr;       // slices of the group's arrays
    if (G->a_force_c) c->force_c = nullptr;
    c->stream = nullptr; p3m_hip_destroy(c);
  }
  gfree(G->a_rho_c); gfree(G->a_force_c);
  if (G->comm) (void)ncclCommDestroy(G->comm);
  gfree(G->d_red4); gfree(G->d_sum3);
  if (G->h_cnt) (void)hipHostFree(G->h_cnt);
  gfree(G->d_gather);
  if (G->h_gather) (void)hipHostFree(G->h_gather);
  if (G->h_hdr) (void)hipHostFree(G->h_hdr);
  if (G->h_red4) (void)hipHostFree(G->h_red4);
  if (G->h_sum3) (void)hipHostFree(G->h_sum3);
  for (int k = 0; k < 2; k++) if (G->h_stage[k]) (void)hipHostFree(G->h_stage[k]);
  fft_plan_destroy(&G->plan_c);
  for (hipStream_t st : G->rstream) if (st) (void)hipStreamDestroy(st);
  for (hipEvent_t e : G->rev_dep) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : G->rev_done) if (e) (void)hipEventDestroy(e);
  if (G->ev_fork) (void)hipEventDestroy(G->ev_fork);
  if (G->stream2) (void)hipStreamDestroy(G->stream2);
  if (G->ev_dep) (void)hipEventDestroy(G->ev_dep);
  if (G->ev_cf) (void)hipEventDestroy(G->ev_cf);
  if (G->stream) (void)hipStreamDestroy(G->stream);
  delete G;
}

extern "C" int p3m_hip_group_create(const p3m_params *base, int32_t proc, int32_t nprocs, p3m_group **out) {
  if (!base || !out || nprocs < 1 || proc < 0 || proc >= nprocs) return P3M_EINVAL;
  *out = nullptr;
  const int nd = base->nodes_dim, nodes = nd * nd * nd;
  if (nodes % nprocs) { p3m_set_error("nodes_dim^3 = %d logical ranks cannot be split evenly over %d processes", nodes, nprocs); return P3M_EINVAL; }
  p3m_group *G = new p3m_group();
  G->base = *base; G->proc = proc; G->nprocs = nprocs; G->nodes = nodes; G->nd = nd;
  auto fail = [&](int code) { p3m_hip_group_destroy(G); return code; };
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { p3m_set_error("no HIP device (this library has no CPU fallback)"); delete G; return P3M_EDEVICE; }
  if (base->device >= 0) G->device = base->device; else (void)hipGetDevice(&G->device);
  if (hipSetDevice(G->device) != hipSuccess) { p3m_set_error("hipSetDevice(%d) failed", G->device); delete G; return P3M_EDEVICE; }
  if (hipStreamCreateWithFlags(&G->stream, hipStreamNonBlocking) != hipSuccess) { delete G; return P3M_EDEVICE; }
  const int per = nodes / nprocs;
  G->owner.resize(nodes); G->lidx.assign(nodes, -1);
  for (int r = 0; r < nodes; r++) G->owner[r] = r / per;   // contiguous blocks: whole z-layers stay on one GPU where possible
  for (int r = proc * per; r < (proc + 1) * per; r++) {
    p3m_params p = *base; p.rank = r; p.device = G->device;
    p3m_ctx *c = nullptr;
    p3m_ctx_share_hint = (proc + 1) * per - r;   // this context and the ones still to come share what is free now
    int rc = p3m_hip_create(&p, &c);
    p3m_ctx_share_hint = 1;
    if (rc) return fail(rc);
    G->rstream.push_back(c->stream); c->stream = G->stream;   // the group's stream; the context's own one serves the forked stretch of a step (rstream)
    G->lidx[r] = (int)G->ctx.size(); G->ctx.push_back(c); G->lrank.push_back(r);
  }
  if (nodes > 1 && !(getenv("P3M_ONE_STREAM") && getenv("P3M_ONE_STREAM")[0] == '1')) {
    if (hipStreamCreateWithFlags(&G->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&G->ev_dep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&G->ev_cf, hipEventDisableTiming) != hipSuccess) return fail(P3M_EDEVICE);
  }
  if (G->stream2 && G->ctx.size() > 1 && getenv("P3M_GROUP_STREAMS") && atoi(getenv("P3M_GROUP_STREAMS")) > 0) {
    G->rev_dep.assign(G->ctx.size(), nullptr); G->rev_done.assign(G->ctx.size(), nullptr);
    bool ok = hipEventCreateWithFlags(&G->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (size_t i = 0; ok && i < G->ctx.size(); i++)
      ok = hipEventCreateWithFlags(&G->rev_dep[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&G->rev_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) return fail(P3M_EDEVICE);
    G->multi = true; G->nrs = std::min<size_t>(G->ctx.size(), (size_t)atoi(getenv("P3M_GROUP_STREAMS")));
  }
  const Geometry &g = G->ctx[0]->g;
  if (nodes > 1 && g.Nn < 2 * g.nb) {   // one ghost shift per axis (k_ghost_pack)
    p3m_set_error("multi-rank groups need nf_physical_node_dim >= 2*nf_buf (%d < %d)", g.Nn, 2 * g.nb);
    return fail(P3M_EINVAL);
  }
  if (nodes > 1) {
    G->pencil = (base->flags & P3M_FLAG_PENCIL) != 0;
    int rc = fft_plan_create(&G->plan_c, g.nc);
    if (rc) return fail(rc);
    if (G->pencil) {
      G->plan_c.px = (G->plan_c.px + 16 * nd - 1) / (16 * nd) * (16 * nd);   // whole chunks for every rank of a y-row; the pad columns are zeros
      G->s = g.ncn / nd; G->nchunk = G->plan_c.px / 16; G->ncl = G->nchunk / nd; G->rpp = g.ncn; G->nxb = nd;   // s = nc_pen (cubepm.par:212)
    } else {
      G->s = g.nc_slab; G->nchunk = G->plan_c.px / 16; G->ncl = G->nchunk; G->rpp = g.nc; G->nxb = nd * nd;
    }
    G->plan_l = G->plan_c; G->plan_l.px = 16 * G->ncl;
    // ghost segments, sized from the rank's capacity: a face shell holds nb/Nn of the particles, an edge (nb/Nn)^2, ...
    {
      // 2.5x the uniform-density share to start with; ghost_pass grows a segment that a clustered shell overfills
      // (P3M_GHOST_SEG_FACTOR: another starting factor, for tests of that path)
      const double f0 = getenv("P3M_GHOST_SEG_FACTOR") ? atof(getenv("P3M_GHOST_SEG_FACTOR")) : 2.5;
      const double f = std::min(1.0, f0 * (double)g.nb / (double)g.Nn);
      int64_t run = 0;
      for (int m = 0; m < 27; m++) {
        const int nz = (m % 3 != 0) + ((m / 3) % 3 != 0) + (m / 9 != 0);
        int64_t cap = nz == 0 ? 0 : (int64_t)((double)g.max_np * (nz == 1 ? f : (nz == 2 ? f * f : f * f * f))) + (f0 < 1.0 ? 16 : 4096);
        cap = std::min<int64_t>(cap, g.max_np);
        // migrants: records that left the rank through this face / edge / corner in one step -- or, with -DMOVE_GRID_BACK
        // (ghosts can turn physical when the grid moves back), every image
        const int64_t capm = nz == 0 ? 0 : ((base->flags & P3M_FLAG_MOVE_GRID_BACK) ? cap : cap / 4 + (f0 < 1.0 ? 16 : 4096));
        G->seg_off[2 * m] = run; G->seg_cap[2 * m] = (int)cap; run += cap;                 // ghosts: one float4 each
        G->seg_off[2 * m + 1] = run; G->seg_cap[2 * m + 1] = (int)capm; run += 2 * capm;   // migrants: two
      }
      if (run > ((int64_t)1 << 36)) { p3m_set_error("ghost segments of %lld records exceed the exchange buffer limit", (long long)run); return fail(P3M_ECAPACITY); }
      G->seg_total = run;
    }
    if (galloc(&G->d_gather, (size_t)64 * nodes) != P3M_OK) return fail(P3M_ENOMEM);
    if (hipHostMalloc(reinterpret_cast<void **>(&G->h_gather), (size_t)64 * nodes * sizeof(int)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&G->h_hdr), (size_t)(per + 1) * sizeof(int)) != hipSuccess) return fail(P3M_ENOMEM);
    const size_t NB = (size_t)G->s * G->ncl * g.nc * 16 * 2;             // floats of one component's complex slab / pencil
    const size_t nrows = (size_t)3 * G->s * G->rpp * 2 * G->plan_c.px;
    const size_t blk = (size_t)G->s * g.ncn * g.ncn;
    const size_t nl = G->ctx.size();
    G->cd.resize(nl);
    G->batched = !G->pencil && fft_has_segmented(G->plan_c) && g.ncn % 4 == 0 && nl <= P3M_MAX_LOCAL && (uint64_t)3 * nl * G->s * G->rpp * G->rpp < 0xffffffffull &&
                 !(getenv("P3M_COARSE_PER_RANK") && getenv("P3M_COARSE_PER_RANK")[0] == '1');   // the switch: A/B runs and tests of the per-rank path
    const size_t face = (size_t)3 * (g.ncn + 2) * (g.ncn + 2), fcs = (size_t)(g.ncn + 2) * (g.ncn + 2) * (g.ncn + 2), n3 = (size_t)g.ncn * g.ncn * g.ncn;
#define A(x) do { int _r = (x); if (_r) return fail(_r); } while (0)
    A(galloc(&G->a_blocks_in, nl * G->nxb * blk)); A(galloc(&G->a_rows, nl * nrows));
    A(galloc(&G->a_ly, nl * 3 * NB)); A(galloc(&G->a_send, nl * 3 * NB)); A(galloc(&G->a_recv, nl * 3 * NB)); A(galloc(&G->a_lz, nl * NB)); A(galloc(&G->a_kern, nl * 3 * NB / 2));
    A(galloc(&G->a_blocks_out, nl * G->nxb * 3 * blk)); A(galloc(&G->a_blocks_back, nl * G->nxb * 3 * blk));
    A(galloc(&G->a_halo, nl * 4 * face)); A(galloc(&G->a_rho_c, nl * n3)); A(galloc(&G->a_force_c, nl * 3 * fcs));
    if (hipMemset(G->a_rows, 0, sizeof(float) * nl * nrows) != hipSuccess) return fail(P3M_EDEVICE);
    G->direct = G->batched && nprocs == 1 && !(getenv("P3M_COARSE_COPY") && getenv("P3M_COARSE_COPY")[0] == '1');
    G->cstride = G->batched ? nl * NB : NB; G->rstride = G->batched ? nl * (nrows / 3) : nrows / 3;
    for (size_t i = 0; i < nl; i++) {
      CoarseDist &d = G->cd[i];
      const size_t ro = G->batched ? i : 3 * i;   // first component of rank i, in slabs
      d.blocks_in = G->a_blocks_in + i * G->nxb * blk; d.rows = G->a_rows + ro * (nrows / 3);
      d.ly = G->a_ly + ro * NB; d.send = G->a_send + ro * NB; d.recv = G->a_recv + ro * NB; d.lz = G->a_lz + i * NB; d.kern = G->a_kern + ro * (NB / 2);
      d.blocks_out = G->a_blocks_out + i * G->nxb * 3 * blk; d.blocks_back = G->a_blocks_back + i * G->nxb * 3 * blk;
      for (int k = 0; k < 2; k++) { d.halo_s[k] = G->a_halo + (i * 4 + k) * face; d.halo_r[k] = G->a_halo + (i * 4 + 2 + k) * face; }
      A(galloc(&d.sb, (size_t)G->seg_total)); A(galloc(&d.rb, (size_t)G->seg_total));
      A(galloc(&d.d_cnt, 128));
      // rho_c and force_c of the contexts become slices too (uniform strides for the batched launches)
      p3m_ctx *c = G->ctx[i];
      (void)hipFree(c->rho_c); (void)hipFree(c->force_c);
      c->rho_c = G->a_rho_c + i * n3; c->force_c = G->a_force_c + i * 3 * fcs;
    }
#undef A
  }
  if (galloc(&G->d_red4, 8) || galloc(&G->d_sum3, 4)) return fail(P3M_ENOMEM);
  if (hipHostMalloc(reinterpret_cast<void **>(&G->h_cnt), sizeof(int) * 128 * G->ctx.size()) != hipSuccess) return fail(P3M_ENOMEM);
  if (hipHostMalloc(reinterpret_cast<void **>(&G->h_red4), sizeof(float) * 8) != hipSuccess) return fail(P3M_ENOMEM);
  if (hipHostMalloc(reinterpret_cast<void **>(&G->h_sum3), sizeof(double) * 4) != hipSuccess) return fail(P3M_ENOMEM);
  G->last.dt_f_acc = G->last.dt_pp_acc = G->last.dt_pp_ext_acc = G->last.dt_c_acc = 1000.f;
  *out = G;
  return P3M_OK;
}

extern "C" int p3m_hip_rccl_unique_id(void *unique_id_128) {
  if (!unique_id_128) return P3M_EINVAL;
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the 128-byte buffer of the ABI");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memset(unique_id_128, 0, 128);
  memcpy(unique_id_128, &id, sizeof(id));
  return P3M_OK;
}
extern "C" int p3m_hip_group_comm_init_rccl(p3m_group *G, const void *unique_id_128, int32_t force_for_local_peers) {
  if (!G || !unique_id_128) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(G->device));
  ncclUniqueId id; memcpy(&id, unique_id_128, sizeof(id));
  NCCL_TRY(ncclCommInitRank(&G->comm, G->nprocs, id, G->proc));
  G->force_nccl = force_for_local_peers != 0;
  if (G->force_nccl) G->direct = false;   // every exchange through RCCL: keep the message path
  return P3M_OK;
}
extern "C" int p3m_hip_group_comm_info(p3m_group *G, int32_t *comm_count, int32_t *comm_rank, int32_t *device, char *uuid_hex33) {
  if (!G) return P3M_EINVAL;
  int cnt = -1, rk = -1;
  if (G->comm) { NCCL_TRY(ncclCommCount(G->comm, &cnt)); NCCL_TRY(ncclCommUserRank(G->comm, &rk)); }
  if (comm_count) *comm_count = cnt;
  if (comm_rank) *comm_rank = rk;
  if (device) *device = G->device;
  if (uuid_hex33) {
    hipUUID u; memset(&u, 0, sizeof(u));
    HIP_TRY(hipDeviceGetUuid(&u, G->device));
    for (int i = 0; i < 16; i++) snprintf(uuid_hex33 + 2 * i, 3, "%02x", (unsigned)(unsigned char)u.bytes[i]);
  }
  return P3M_OK;
}
extern "C" int p3m_hip_group_set_transport(p3m_group *G, const p3m_transport *t) {
  if (!G || !t || !t->exchange || !t->allreduce_max_f32 || !t->allreduce_sum_f64) return P3M_EINVAL;
  G->tr = *t; G->have_tr = true;
  return P3M_OK;
}
extern "C" int32_t p3m_hip_group_nlocal(const p3m_group *G) { return G ? (int32_t)G->ctx.size() : -1; }
extern "C" int32_t p3m_hip_group_local_rank(const p3m_group *G, int32_t i) { return (G && i >= 0 && i < (int)G->lrank.size()) ? G->lrank[i] : -1; }
extern "C" p3m_ctx *p3m_hip_group_ctx(p3m_group *G, int32_t i) { return (G && i >= 0 && i < (int)G->ctx.size()) ? G->ctx[i] : nullptr; }

// ================================================================== ghost pass (particle_pass.f90)
// One round for all 26 directions (particles.hip, k_ghost_pack): counts first, then the records.
static int shift_neighbour(const p3m_group *G, int r, int m) {
  const int nd = G->nd, c1 = r / (nd * nd), c2 = (r / nd) % nd, c3 = r % nd;
  const int a = m % 3, b = (m / 3) % 3, c = m / 9;                  // x <-> c3, y <-> c2, z <-> c1; 1: + neighbour, 2: - neighbour
  auto mv = [&](int v, int s) { return s == 1 ? (v + 1) % nd : (s == 2 ? (v - 1 + nd) % nd : v); };
  return mv(c1, c) * nd * nd + mv(c2, b) * nd + mv(c3, a);
}
// segment layout from the capacities: slot k = 2m (ghosts, one float4 each) or 2m+1 (migrants, two)
static int64_t layout_segments(p3m_group *G) {
  int64_t run = 0;
  for (int k = 0; k < 54; k++) { G->seg_off[k] = run; run += (int64_t)G->seg_cap[k] * ((k & 1) ? 2 : 1); }
  return run;
}
static int ghost_pass(p3m_group *G) {
  const int nl = (int)G->ctx.size();
  if (G->nodes == 1) return particles_pass_self(G->ctx[0]);
  // slot k = 2m (ghosts of shift m, 16 B each) or 2m+1 (migrants, 32 B each); d_cnt[0..53]: own counts, [54]: np_local
  // 1. pack every image into the segment of its slot
  auto pack_all = [&]() -> int {
    for (int i = 0; i < nl; i++) {
      CoarseDist &d = G->cd[i];
      HIP_TRY(hipMemsetAsync(d.d_cnt, 0, 64 * sizeof(int), G->stream));
      P3M_TRY(particles_ghost_pack(G->ctx[i], d.sb, G->seg_off, G->seg_cap, d.d_cnt));
    }
    return P3M_OK;
  };
  P3M_TRY(pack_all());
  // 2. every process learns EVERY rank's counts (and record count): 64 ints per rank, gathered at each process.  All
  //    decisions below -- capacity errors, growing the segments, message sizes -- are then taken from the same numbers by
  //    every process: nobody walks into a payload exchange that a peer has already abandoned.
  const int per = G->nodes / G->nprocs;
  for (int i = 0; i < nl; i++) {
    G->h_hdr[i] = G->ctx[i]->np_local;
    HIP_TRY(hipMemcpyAsync(G->cd[i].d_cnt + 54, G->h_hdr + i, sizeof(int), hipMemcpyHostToDevice, G->stream));
  }
  std::vector<XMsg> cm;
  for (int r = 0; r < G->nodes; r++)
    for (int q = 0; q < G->nprocs; q++) {
      const int li = G->lidx[r];
      cm.push_back({r, q * per, li >= 0 ? (const void *)G->cd[li].d_cnt : nullptr, q == G->proc ? (void *)(G->d_gather + 64 * r) : nullptr, 64 * sizeof(int)});
    }
  P3M_TRY(do_exchange(G, cm));
  HIP_TRY(hipMemcpyAsync(G->h_gather, G->d_gather, (size_t)64 * G->nodes * sizeof(int), hipMemcpyDeviceToHost, G->stream));
  HIP_TRY(hipStreamSynchronize(G->stream));
  const int *hg = G->h_gather;
  const int64_t cap = G->ctx[0]->cap;
  // 3. the same checks for every rank on every process (particle_pass.f90:96-99, :136-139)
  std::vector<int64_t> in(G->nodes, 0);
  int need[54] = {0};
  for (int r = 0; r < G->nodes; r++)
    for (int k = 2; k < 54; k++) { in[shift_neighbour(G, r, k / 2)] += hg[64 * r + k]; need[k] = std::max(need[k], hg[64 * r + k]); }
  for (int r = 0; r < G->nodes; r++)
    if ((int64_t)hg[64 * r + 54] + in[r] > cap) {
      p3m_set_error("rank %d: exceeded max_np in pass: %lld > %lld (particle_pass.f90:136-139); raise density_buffer", r, (long long)hg[64 * r + 54] + in[r], (long long)cap);
      return P3M_ECAPACITY;
    }
  bool grow = false;
  for (int k = 2; k < 54; k++) grow = grow || need[k] > G->seg_cap[k];
  if (grow) {
    // a segment was too small for this step's (clustered) shell: the reference's per-direction buffer holds max_buf = 2.2
    // max_np floats (cubepm.par:174); here the segments grow to what is needed (+25 %), identically on every process, and
    // the pack runs again (it counts past a full segment but does not write there)
    for (int k = 2; k < 54; k++)
      if (need[k] > G->seg_cap[k]) G->seg_cap[k] = (int)std::min<int64_t>(cap, (int64_t)need[k] + need[k] / 4 + 1024);
    const int64_t run = layout_segments(G);
    if (run > ((int64_t)1 << 36)) { p3m_set_error("ghost segments of %lld records exceed the exchange buffer limit", (long long)run); return P3M_ECAPACITY; }
    G->seg_total = run;
    HIP_TRY(hipStreamSynchronize(G->stream));
    for (int i = 0; i < nl; i++) {
      CoarseDist &d = G->cd[i];
      gfree(d.sb); gfree(d.rb);
      P3M_TRY(galloc(&d.sb, (size_t)run)); P3M_TRY(galloc(&d.rb, (size_t)run));
    }
    P3M_TRY(pack_all());
  }
  // 4. payloads: sender and receiver read the size of every message from the gathered counts
  std::vector<XMsg> pm;
  for (int r = 0; r < G->nodes; r++)
    for (int k = 2; k < 54; k++) {
      const int dst = shift_neighbour(G, r, k / 2), li = G->lidx[r], ld = G->lidx[dst];
      const size_t bytes = (size_t)hg[64 * r + k] * ((k & 1) ? 32 : 16);
      pm.push_back({r, dst, li >= 0 ? (const void *)(G->cd[li].sb + (size_t)G->seg_off[k]) : nullptr,
                    ld >= 0 ? (void *)(G->cd[ld].rb + (size_t)G->seg_off[k]) : nullptr, bytes});
    }
  P3M_TRY(do_exchange(G, pm));
  // 5. append: slot k of rank d is filled by the one rank whose shift k/2 lands on d
  for (int i = 0; i < nl; i++) {
    const int d = G->lrank[i];
    int cnt_in[64] = {0};
    for (int r = 0; r < G->nodes; r++)
      for (int k = 2; k < 54; k++) if (shift_neighbour(G, r, k / 2) == d) cnt_in[k] = hg[64 * r + k];
    P3M_TRY(particles_ghost_unpack(G->ctx[i], G->cd[i].rb, G->seg_off, cnt_in, G->ctx[i]->np_local));
    G->ctx[i]->np_all = G->ctx[i]->np_local + (int)in[d];
  }
  return P3M_OK;
}

// ================================================================== coarse mesh, distributed
// blocks [(j*nd+i)][zl][yy][xx] -> rows [zl][y][x], ny rows per plane (nc: slabs, blocks of nd x nd cubes; ncn: pencils, nd cubes along x)
__global__ __launch_bounds__(256) void k_blocks_to_rows(const float *__restrict__ blocks, float *__restrict__ rows, int s, int nc, int ny, int ncn, int nd, int rp) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)s * ny * rp) return;
  const int x = (int)(idx % rp), y = (int)((idx / rp) % ny), zl = (int)(idx / ((int64_t)rp * ny));
  float v = 0.f;
  if (x < nc) {
    const int i = x / ncn, j = y / ncn;
    v = blocks[(((int64_t)(j * nd + i) * s + zl) * ncn + (y - j * ncn)) * ncn + (x - i * ncn)];
  }
  rows[idx] = v;
}
// recv [comp][t][p][chunk][q][16] -> out [comp][p][chunk][t*s+q][16]   (complex)
__global__ __launch_bounds__(256) void k_a2a_permute(const float2 *__restrict__ recv, float2 *__restrict__ out, int s, int nchunk, int nc, int ncomp) {
  const int64_t per = (int64_t)s * nchunk * nc * 16;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per * ncomp) return;
  const int comp = (int)(idx / per); const int64_t r = idx - comp * per;
  const int l = (int)(r % 16), q = (int)((r / 16) % s), chunk = (int)((r / (16 * s)) % nchunk), p = (int)((r / ((int64_t)16 * s * nchunk)) % s);
  const int t = (int)(r / ((int64_t)16 * s * nchunk * s));
  out[comp * per + (((int64_t)p * nchunk + chunk) * nc + (t * s + q)) * 16 + l] = recv[idx];
}
// Index shuffle of 128-byte bundles (16 complex): source dims d[0..4] (outermost first, contiguous), destination strides
// st[0..4] in bundles.  Packs and unpacks the x<->y transpose of the pencil decomposition.
struct Perm5 { int d[5]; int64_t st[5]; };
__global__ __launch_bounds__(256) void k_permute5(const float4 *__restrict__ src, float4 *__restrict__ dst, Perm5 p, int64_t nbundle) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nbundle * 8) return;
  int64_t e = idx >> 3, off = 0;
#pragma unroll
  for (int k = 4; k >= 0; k--) { const int64_t q = e / p.d[k]; off += (e - q * p.d[k]) * p.st[k]; e = q; }
  dst[off * 8 + (idx & 7)] = src[idx];
}
static int permute5(hipStream_t st, const float *src, float *dst, const int (&d)[5], const int64_t (&s5)[5]) {
  Perm5 p; int64_t n = 1;
  for (int k = 0; k < 5; k++) { p.d[k] = d[k]; p.st[k] = s5[k]; n *= d[k]; }
  hipLaunchKernelGGL(k_permute5, dim3(cdiv(n * 8, 256)), dim3(256), 0, st, reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), p, n);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
// rows [comp][zl][y][x] (ny = nj*ncn rows per plane) -> blocks_out [(j*nd+i)][comp][zl][yy][xx]
__global__ __launch_bounds__(256) void k_rows_to_blocks(const float *__restrict__ rows, float *__restrict__ blocks, int s, int nj, int ncn, int nd, int rp) {
  const int64_t tot = (int64_t)nj * nd * 3 * s * ncn * ncn;
  const int nc = nj * ncn;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= tot) return;
  const int xx = (int)(idx % ncn), yy = (int)((idx / ncn) % ncn), zl = (int)((idx / ((int64_t)ncn * ncn)) % s);
  const int comp = (int)((idx / ((int64_t)ncn * ncn * s)) % 3), ji = (int)(idx / ((int64_t)ncn * ncn * s * 3));
  const int j = ji / nd, i = ji % nd;
  blocks[idx] = rows[(((int64_t)comp * s + zl) * nc + (j * ncn + yy)) * rp + (i * ncn + xx)];
}
// blocks_back [q][comp][zl][yy][xx] -> force_c[comp][1+q*s+zl][1+yy][1+xx]
__global__ __launch_bounds__(256) void k_blocks_to_force(const float *__restrict__ blocks, float *__restrict__ fc, int s, int ncn, int nq) {
  const int64_t tot = (int64_t)nq * 3 * s * ncn * ncn;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= tot) return;
  const int xx = (int)(idx % ncn), yy = (int)((idx / ncn) % ncn), zl = (int)((idx / ((int64_t)ncn * ncn)) % s);
  const int comp = (int)((idx / ((int64_t)ncn * ncn * s)) % 3), q = (int)(idx / ((int64_t)ncn * ncn * s * 3));
  const int m = ncn + 2;
  fc[(int64_t)comp * m * m * m + ((int64_t)(1 + q * s + zl) * m + (1 + yy)) * m + (1 + xx)] = blocks[idx];
}
// coarse_force_buffer.f90: face `pl` of axis (0=x,1=y,2=z), full extent (0..ncn+1) of the other two
__global__ __launch_bounds__(256) void k_halo_pack(const float *__restrict__ fc, float *__restrict__ buf, int ncn, int axis, int pl) {
  const int m = ncn + 2; const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)3 * m * m) return;
  const int a = (int)(idx % m), b = (int)((idx / m) % m), comp = (int)(idx / ((int64_t)m * m));
  int i, j, k;
  if (axis == 0) { i = pl; j = a; k = b; } else if (axis == 1) { i = a; j = pl; k = b; } else { i = a; j = b; k = pl; }
  buf[idx] = fc[(int64_t)comp * m * m * m + ((int64_t)k * m + j) * m + i];
}
__global__ __launch_bounds__(256) void k_halo_unpack(float *__restrict__ fc, const float *__restrict__ buf, int ncn, int axis, int pl) {
  const int m = ncn + 2; const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)3 * m * m) return;
  const int a = (int)(idx % m), b = (int)((idx / m) % m), comp = (int)(idx / ((int64_t)m * m));
  int i, j, k;
  if (axis == 0) { i = pl; j = a; k = b; } else if (axis == 1) { i = a; j = pl; k = b; } else { i = a; j = b; k = pl; }
  fc[(int64_t)comp * m * m * m + ((int64_t)k * m + j) * m + i] = buf[idx];
}
__global__ __launch_bounds__(256) void k_gmax_interior(const float *__restrict__ fc, int n, float *__restrict__ out) {
  const int m = n + 2; const int64_t cs = (int64_t)m * m * m;
  float mx = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)n * n * n; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx % n), j = (int)((idx / n) % n), k = (int)(idx / ((int64_t)n * n));
    const int64_t o = ((int64_t)(k + 1) * m + (j + 1)) * m + (i + 1);
    const float a = fc[o], b = fc[o + cs], d = fc[o + 2 * cs];
    mx = fmaxf(mx, sqrtf(a * a + b * b + d * d));                        // coarse_max_dt.f90:24-31
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) p3m_atomic_max_nonneg(out + p3m_slot() * 16, mx);
}
// real-space coarse kernel on this rank's cube, global coordinates (kernel_initialization.f90:293-336, :366-457)
__global__ __launch_bounds__(256) void k_ck_cube(float *__restrict__ cube, const float *__restrict__ table, int ncn, int nc, int ox, int oy, int oz, int ms,
                                                 int comp, int use_table) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)ncn * ncn * ncn) return;
  const int c3[3] = {(int)(idx % ncn) + ox, (int)((idx / ncn) % ncn) + oy, (int)(idx / ((int64_t)ncn * ncn)) + oz};
  float xs[3]; int t3[3]; bool in_table = use_table != 0; float sgn = 1.f;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const float w = (c3[d] < nc / 2 + 1) ? (float)c3[d] : (float)(c3[d] - nc);
    xs[d] = (float)ms * w;
    if (c3[d] < 4) t3[d] = c3[d];
    else if (c3[d] > nc - 4) { t3[d] = nc - c3[d]; if (d == comp) sgn = -sgn; }
    else in_table = false;
  }
  float v;
  if (in_table) v = sgn * table[(((int64_t)t3[2] * 4 + t3[1]) * 4 + t3[0]) * 3 + comp];
  else { const float rr = sqrtf(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]); v = (rr == 0.0f) ? 0.f : -xs[comp] / (rr * rr * rr); }
  cube[idx] = v;
}
__global__ __launch_bounds__(256) void k_take_imag_g(const float *__restrict__ hat, float *__restrict__ kern, int64_t ncomplex) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < ncomplex) kern[i] = hat[2 * i + 1];
}
// LRCKCORR (kernel_initialization.f90:562-591) on the local ky-slab: kern/uncorr in [yl][chunk][z][16]
__global__ __launch_bounds__(256) void k_lrck_slab(float *__restrict__ kern, const float *__restrict__ uncorr, int n, int s, int nchunk, int ky0, int kx0, int comp) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)s * nchunk * n * 16) return;
  const int l = (int)(idx % 16), k = (int)((idx / 16) % n), chunk = (int)((idx / (16 * (int64_t)n)) % nchunk), yl = (int)(idx / (16 * (int64_t)n * nchunk));
  const int kx = kx0 + chunk * 16 + l, j = ky0 + yl;
  if (kx > n / 2) return;
  const int ky = (j < n / 2 + 1) ? j : j - n, kz = (k < n / 2 + 1) ? k : k - n;
  const float kr = sqrtf((float)(kx * kx + ky * ky + kz * kz));
  if (!(kr <= 8.f)) return;
  const int kk = comp == 0 ? kx : (comp == 1 ? ky : kz);
  if (kk == 0) return;
  const float ka = 2 * sinf(P3M_PI_F * kx / (float)n), kb = 2 * sinf(P3M_PI_F * ky / (float)n), kc = 2 * sinf(P3M_PI_F * kz / (float)n);
  const float kq = comp == 0 ? ka : (comp == 1 ? kb : kc);
  const float wc = 4.f * P3M_PI_F * kq / (ka * ka + kb * kb + kc * kc) / 16.f;
  kern[idx] = kern[idx] * (wc / uncorr[idx]);
}

// ------------------------------------------------------------------ batched layout kernels (G->batched: slabs, ncn % 4 == 0)
// One launch serves every local rank; a wavefront moves whole rows with 16-byte accesses and does the index arithmetic once per
// row (the per-element kernels above spend a chain of 64-bit divisions on every float: 1.8 ms for a 1.6 GB array).
// blocks_in [rank][(j*nd+i)][zl][yy][xx] -> rows [rank][zl][y][x] (component 0 of the rows array), pad columns zero
__global__ __launch_bounds__(256) void k_blocks_to_rows_b(const float *__restrict__ blocks, float *__restrict__ rows, RowGeom q) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= (unsigned)q.nl * q.s * q.rpp) return;
  fdiv_t d_rpp{q.m_rpp, q.rpp}, d_s{q.m_s, q.s}, d_ncn{q.m_ncn, q.ncn};
  const unsigned plane = fdiv(row, d_rpp), y = row - plane * q.rpp, rank = fdiv(plane, d_s), zl = plane - rank * q.s;
  const unsigned j = fdiv(y, d_ncn), yy = y - j * q.ncn;
  const int64_t blk = (int64_t)q.s * q.ncn * q.ncn;
  float4 *dst = reinterpret_cast<float4 *>(rows + (int64_t)row * q.rp);
  for (int x4 = lane; x4 < q.rp / 4; x4 += 64) {
    const int x = 4 * x4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < q.nc) {
      const unsigned i = fdiv(x, d_ncn);
      v = *reinterpret_cast<const float4 *>(blocks + ((int64_t)rank * q.nd * q.nd + (j * q.nd + i)) * blk + ((int64_t)zl * q.ncn + yy) * q.ncn + (x - i * q.ncn));
    }
    dst[x4] = v;
  }
}
// the same from the ranks' cubes themselves (G->direct): block (j*nd+i) of rank r is z-slice r % nd^2 of the cube of rank
// layer(r) + j*nd + i (pack_slab, fftw3ds.f90:24-52)
__global__ __launch_bounds__(256) void k_cubes_to_rows_b(const float *__restrict__ cubes, float *__restrict__ rows, RowGeom q) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= (unsigned)q.nl * q.s * q.rpp) return;
  fdiv_t d_rpp{q.m_rpp, q.rpp}, d_s{q.m_s, q.s}, d_ncn{q.m_ncn, q.ncn};
  const unsigned plane = fdiv(row, d_rpp), y = row - plane * q.rpp, rank = fdiv(plane, d_s), zl = plane - rank * q.s;
  const unsigned j = fdiv(y, d_ncn), yy = y - j * q.ncn;
  const unsigned nd2 = q.nd * q.nd, layer = rank / nd2, qz = rank - layer * nd2;
  const int64_t n3 = (int64_t)q.ncn * q.ncn * q.ncn;
  float4 *dst = reinterpret_cast<float4 *>(rows + (int64_t)row * q.rp);
  for (int x4 = lane; x4 < q.rp / 4; x4 += 64) {
    const int x = 4 * x4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < q.nc) {
      const unsigned i = fdiv(x, d_ncn);
      v = *reinterpret_cast<const float4 *>(cubes + (int64_t)(layer * nd2 + j * q.nd + i) * n3 + ((int64_t)(qz * q.s + zl) * q.ncn + yy) * q.ncn + (x - i * q.ncn));
    }
    dst[x4] = v;
  }
}
// rows [comp][rank][zl][y][x] -> blocks_out [rank][(j*nd+i)][comp][zl][yy][xx]; one wavefront per (comp, rank, zl, y) row
__global__ __launch_bounds__(256) void k_rows_to_blocks_b(const float *__restrict__ rows, float *__restrict__ blocks, RowGeom q) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 3u * q.nl * q.s * q.rpp) return;
  fdiv_t d_rpp{q.m_rpp, q.rpp}, d_s{q.m_s, q.s}, d_ncn{q.m_ncn, q.ncn};
  const unsigned plane = fdiv(row, d_rpp), y = row - plane * q.rpp, pr = fdiv(plane, d_s), zl = plane - pr * q.s;   // pr = comp*nl + rank
  const unsigned comp = pr / (unsigned)q.nl, rank = pr - comp * q.nl;
  const unsigned j = fdiv(y, d_ncn), yy = y - j * q.ncn;
  const int64_t blk = (int64_t)q.s * q.ncn * q.ncn;
  const float4 *src = reinterpret_cast<const float4 *>(rows + (int64_t)row * q.rp);
  for (int x4 = lane; x4 < q.nc / 4; x4 += 64) {
    const int x = 4 * x4; const unsigned i = fdiv(x, d_ncn);
    *reinterpret_cast<float4 *>(blocks + (((int64_t)rank * q.nd * q.nd + (j * q.nd + i)) * 3 + comp) * blk + ((int64_t)zl * q.ncn + yy) * q.ncn + (x - i * q.ncn)) = src[x4];
  }
}
// blocks_back [rank][qz][comp][zl][yy][xx] -> force_c [rank][comp][1+qz*s+zl][1+yy][1+xx], and max |F| over the interior
// (coarse_max_dt.f90:24-31) on the way: one wavefront per (rank, qz, zl, yy) takes the three component rows
__global__ __launch_bounds__(256) void k_blocks_to_force_b(const float *__restrict__ blocks, float *__restrict__ fc, int nl, int nq, int s, int ncn, unsigned m_ncn,
                                                           unsigned m_s, RankPtrs red) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned nrow = (unsigned)nl * nq * s * ncn;
  float mx = 0.f; unsigned rank = 0;
  if (row < nrow) {
    fdiv_t d_ncn{m_ncn, ncn}, d_s{m_s, s};
    const unsigned t = fdiv(row, d_ncn), yy = row - t * ncn, t2 = fdiv(t, d_s), zl = t - t2 * s, qz = t2 % (unsigned)nq;
    rank = t2 / (unsigned)nq;
    const int m = ncn + 2; const int64_t fcs = (int64_t)m * m * m, blk = (int64_t)s * ncn * ncn;
    const float *b0 = blocks + (((int64_t)rank * nq + qz) * 3) * blk + ((int64_t)zl * ncn + yy) * ncn;
    float *f0 = fc + (int64_t)rank * 3 * fcs + ((int64_t)(1 + qz * s + zl) * m + (1 + yy)) * m + 1;
    for (int x4 = lane; x4 < ncn / 4; x4 += 64) {
      const float4 a = reinterpret_cast<const float4 *>(b0)[x4], b = reinterpret_cast<const float4 *>(b0 + blk)[x4], c = reinterpret_cast<const float4 *>(b0 + 2 * blk)[x4];
      float *pa = f0 + 4 * x4;
      pa[0] = a.x; pa[1] = a.y; pa[2] = a.z; pa[3] = a.w;
      pa[fcs] = b.x; pa[fcs + 1] = b.y; pa[fcs + 2] = b.z; pa[fcs + 3] = b.w;
      pa[2 * fcs] = c.x; pa[2 * fcs + 1] = c.y; pa[2 * fcs + 2] = c.z; pa[2 * fcs + 3] = c.w;
      mx = fmaxf(fmaxf(mx, sqrtf(a.x * a.x + b.x * b.x + c.x * c.x)), sqrtf(a.y * a.y + b.y * b.y + c.y * c.y));
      mx = fmaxf(fmaxf(mx, sqrtf(a.z * a.z + b.z * b.z + c.z * c.z)), sqrtf(a.w * a.w + b.w * b.w + c.w * c.w));
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
  if (lane == 0 && row < nrow) p3m_atomic_max_nonneg(red.p[rank] + p3m_slot() * 16, mx);
}
// G->direct: rows [comp][rank][zl][y][x] -> force_c of the OWNERS, and max |F| on the way (unpack_slab, fftw3ds.f90:69-99 +
// coarse_max_dt.f90:24-31): row (rank r, zl, y) holds, for i < nd, cells of the cube of rank layer(r) + (y/ncn)*nd + i, in its
// plane (r % nd^2)*s + zl.  One wavefront takes the three component rows.
__global__ __launch_bounds__(256) void k_rows_to_force_b(const float *__restrict__ rows, float *__restrict__ fc, RowGeom q, int64_t rstride, RankPtrs red) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= (unsigned)q.nl * q.s * q.rpp) return;
  fdiv_t d_rpp{q.m_rpp, q.rpp}, d_s{q.m_s, q.s}, d_ncn{q.m_ncn, q.ncn};
  const unsigned plane = fdiv(row, d_rpp), y = row - plane * q.rpp, rank = fdiv(plane, d_s), zl = plane - rank * q.s;
  const unsigned j = fdiv(y, d_ncn), yy = y - j * q.ncn;
  const unsigned nd2 = q.nd * q.nd, layer = rank / nd2, qz = rank - layer * nd2;
  const int m = q.ncn + 2; const int64_t fcs = (int64_t)m * m * m;
  const float *r0 = rows + (int64_t)row * q.rp;
  for (unsigned i = 0; i < (unsigned)q.nd; i++) {              // the nd owners along x
    const unsigned owner = layer * nd2 + j * q.nd + i;
    float *f0 = fc + (int64_t)owner * 3 * fcs + ((int64_t)(1 + qz * q.s + zl) * m + (1 + yy)) * m + 1;
    float mx = 0.f;
    for (int x4 = lane; x4 < q.ncn / 4; x4 += 64) {
      const float4 a = reinterpret_cast<const float4 *>(r0 + i * q.ncn)[x4], b = reinterpret_cast<const float4 *>(r0 + rstride + i * q.ncn)[x4],
                   c = reinterpret_cast<const float4 *>(r0 + 2 * rstride + i * q.ncn)[x4];
      float *pa = f0 + 4 * x4;
      pa[0] = a.x; pa[1] = a.y; pa[2] = a.z; pa[3] = a.w;
      pa[fcs] = b.x; pa[fcs + 1] = b.y; pa[fcs + 2] = b.z; pa[fcs + 3] = b.w;
      pa[2 * fcs] = c.x; pa[2 * fcs + 1] = c.y; pa[2 * fcs + 2] = c.z; pa[2 * fcs + 3] = c.w;
      mx = fmaxf(fmaxf(mx, sqrtf(a.x * a.x + b.x * b.x + c.x * c.x)), sqrtf(a.y * a.y + b.y * b.y + c.y * c.y));
      mx = fmaxf(fmaxf(mx, sqrtf(a.z * a.z + b.z * b.z + c.z * c.z)), sqrtf(a.w * a.w + b.w * b.w + c.w * c.w));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
    if (lane == 0) p3m_atomic_max_nonneg(red.p[owner] + p3m_slot() * 16, mx);
  }
}
// coarse_force_buffer.f90 for all local ranks: blockIdx.y = rank*2 + side; halo [rank][4][face]: slots 0,1 = send to -axis / +axis, 2,3 = received
__global__ __launch_bounds__(256) void k_halo_pack_b(const float *__restrict__ fc, float *__restrict__ halo, int ncn, int axis) {
  const int m = ncn + 2; const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)3 * m * m) return;
  const int rank = blockIdx.y >> 1, side = blockIdx.y & 1, pl = side ? ncn : 1;
  const int a = (int)(idx % m), b = (int)((idx / m) % m), comp = (int)(idx / ((int64_t)m * m));
  int i, j, k;
  if (axis == 0) { i = pl; j = a; k = b; } else if (axis == 1) { i = a; j = pl; k = b; } else { i = a; j = b; k = pl; }
  const int64_t fcs = (int64_t)m * m * m, face = (int64_t)3 * m * m;
  halo[((int64_t)rank * 4 + side) * face + idx] = fc[(int64_t)rank * 3 * fcs + (int64_t)comp * fcs + ((int64_t)k * m + j) * m + i];
}
__global__ __launch_bounds__(256) void k_halo_unpack_b(float *__restrict__ fc, const float *__restrict__ halo, int ncn, int axis) {
  const int m = ncn + 2; const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)3 * m * m) return;
  const int rank = blockIdx.y >> 1, side = blockIdx.y & 1, pl = side ? 0 : ncn + 1;   // slot 2 (from the + neighbour's plane 1) fills ncn+1, slot 3 fills 0
  const int a = (int)(idx % m), b = (int)((idx / m) % m), comp = (int)(idx / ((int64_t)m * m));
  int i, j, k;
  if (axis == 0) { i = pl; j = a; k = b; } else if (axis == 1) { i = a; j = pl; k = b; } else { i = a; j = b; k = pl; }
  const int64_t fcs = (int64_t)m * m * m, face = (int64_t)3 * m * m;
  fc[(int64_t)rank * 3 * fcs + (int64_t)comp * fcs + ((int64_t)k * m + j) * m + i] = halo[((int64_t)rank * 4 + 2 + side) * face + idx];
}

// ---- who exchanges with whom in the transposes of the coarse transform (rank = c1*nd^2 + c2*nd + c3, x <-> c3, z <-> c1)
// cube <-> x-lines.  Slabs (pack_slab, fftw3ds.f90:24-52): z-slice q of r's cube goes to the q-th rank of the own z-layer.
// Pencils (pack_pencils, p3dfft_coarse.f90:69-127 with pen_neighbor_to / pen_neighbor_fm of mpi_initialization_p3dfft.f90:48-51):
// z-slice q goes to rank (c1, q, c2), so that rank (c1, c2, c3) holds the x-pencil of y block c3 and z planes
// c1*ncn + c2*nc_pen ..: p3dfft's processor grid, rank = iy + nd*iz with iy = c3, iz = c1*nd + c2.  At the receiver the
// block sits at the sender's x position, xl_index
static inline int xl_peer(const p3m_group *G, int r, int q) {
  const int nd = G->nd, layer = (r / (nd * nd)) * nd * nd;
  return G->pencil ? layer + q * nd + (r / nd) % nd : layer + q;
}
static inline int xl_index(const p3m_group *G, int r) { return G->pencil ? r % G->nd : r % (G->nd * G->nd); }
// x <-> y transpose (pencils only): kx chunk range j goes to the j-th of the nd ranks with the same z planes (same iz);
// arrivals stack along y by the sender's iy
static inline int xy_peer(const p3m_group *G, int r, int j) { return (r / G->nd) * G->nd + j; }
static inline int xy_index(const p3m_group *G, int r) { return r % G->nd; }
// y <-> z transpose: ky block j goes to yz_peer(r, j).  Slabs: all ranks.  Pencils: the nd^2 ranks that hold the same kx
// chunks (same iy), in the order of their z planes
static inline int yz_count(const p3m_group *G) { return G->pencil ? G->nd * G->nd : G->nodes; }
static inline int yz_peer(const p3m_group *G, int r, int j) { return G->pencil ? j * G->nd + r % G->nd : j; }
static inline int yz_index(const p3m_group *G, int r) { return G->pencil ? r / G->nd : r; }

// the messages of one transpose among groups of `count` ranks: block j of r's `send` (bytes each) lands as block index(r) of
// peer(r, j)'s `recv`
template <typename P, typename I, typename S, typename R>
static void group_msgs(p3m_group *G, std::vector<XMsg> &m, int count, size_t bytes, P peer, I index, S sendp, R recvp) {
  for (int r = 0; r < G->nodes; r++)
    for (int j = 0; j < count; j++) {
      const int t = peer(r, j), lr = G->lidx[r], lt = G->lidx[t];
      m.push_back({r, t, lr >= 0 ? (const void *)(sendp(lr) + (size_t)j * bytes) : nullptr, lt >= 0 ? (void *)(recvp(lt) + (size_t)index(r) * bytes) : nullptr, bytes});
    }
}
template <typename P, typename I, typename S, typename R>
static int group_exchange(p3m_group *G, int count, size_t bytes, P peer, I index, S sendp, R recvp) {
  std::vector<XMsg> m;
  group_msgs(G, m, count, bytes, peer, index, sendp, recvp);
  return do_exchange(G, m);
}

static RowGeom row_geom(const p3m_group *G) {
  const Geometry &g = G->ctx[0]->g;
  RowGeom q; q.nl = (int)G->ctx.size(); q.s = G->s; q.nc = g.nc; q.ncn = g.ncn; q.nd = G->nd; q.rpp = G->rpp; q.rp = 2 * G->plan_c.px;
  q.m_rpp = fdiv_magic(q.rpp); q.m_s = fdiv_magic(q.s); q.m_ncn = fdiv_magic(q.ncn);
  return q;
}
// forward distributed transform of every rank's cube `cube_of(i)` (ncn^3 floats) into d.lz: rho-hat of the own ky slab and
// (pencils) the own kx chunks, all kz
template <typename F> static int dist_forward(p3m_group *G, F cube_of) {
  const Geometry &g = G->ctx[0]->g;
  const int nl = (int)G->ctx.size(), nd = G->nd, s = G->s, nc = g.nc, ncn = g.ncn, rp = 2 * G->plan_c.px, ncl = G->ncl, rpp = G->rpp;
  const size_t blk = (size_t)s * ncn * ncn;
  if (G->direct && cube_of(0) == G->a_rho_c) {   // no exchanges: gather from the cubes, store into the peers' receive blocks
    p3m_ctx *c0 = G->ctx[0];
    const RowGeom q = row_geom(G);
    if (fft_x_has_cubes(G->plan_c, q)) P3M_TRY(fft_x_forward_cubes(c0, G->plan_c, G->a_rho_c, G->a_ly, q));   // the x pass reads the cubes themselves
    else {
      hipLaunchKernelGGL(k_cubes_to_rows_b, dim3(cdiv((int64_t)nl * s * rpp, 4)), dim3(256), 0, G->stream, (const float *)G->a_rho_c, G->a_rows, q);
      HIP_TRY(hipGetLastError());
      P3M_TRY(fft_x_forward_rows(c0, G->plan_c, G->a_rows, G->a_ly, (int64_t)nl * s * rpp, rpp));
    }
    P3M_TRY(fft_slab_y_fwd(c0, G->plan_c, G->a_ly, G->a_recv, s, nl, true));
    return fft_slab_z_fwd(c0, G->plan_l, G->a_recv, G->a_lz, s, s, nl);
  }
  // cube -> x-lines
  P3M_TRY(group_exchange(G, G->nxb, blk * sizeof(float), [&](int r, int q) { return xl_peer(G, r, q); }, [&](int r) { return xl_index(G, r); },
                         [&](int li) { return (const char *)cube_of(li); }, [&](int li) { return (char *)G->cd[li].blocks_in; }));
  if (G->batched) {   // every stage once, over all local ranks
    p3m_ctx *c0 = G->ctx[0];
    const RowGeom q = row_geom(G);
    hipLaunchKernelGGL(k_blocks_to_rows_b, dim3(cdiv((int64_t)nl * s * rpp, 4)), dim3(256), 0, G->stream, (const float *)G->a_blocks_in, G->a_rows, q);
    HIP_TRY(hipGetLastError());
    P3M_TRY(fft_x_forward_rows(c0, G->plan_c, G->a_rows, G->a_ly, (int64_t)nl * s * rpp, rpp));
    P3M_TRY(fft_slab_y_fwd(c0, G->plan_c, G->a_ly, G->a_send, s, nl));
    P3M_TRY(group_exchange(G, yz_count(G), (size_t)s * ncl * s * 16 * sizeof(float2), [&](int r, int j) { return yz_peer(G, r, j); },
                           [&](int r) { return yz_index(G, r); }, [&](int li) { return (const char *)G->cd[li].send; }, [&](int li) { return (char *)G->cd[li].recv; }));
    return fft_slab_z_fwd(c0, G->plan_l, G->a_recv, G->a_lz, s, s, nl);   // reads the arrivals where they lie
  }
  const size_t NBc = (size_t)s * ncl * nc * 16;   // complex elements of one component's slab / pencil
  const size_t xyb = (size_t)s * ncl * ncn * 16 * sizeof(float2);   // pencils: one block of the x<->y transpose
  for (int i = 0; i < nl; i++) {
    p3m_ctx *c = G->ctx[i]; CoarseDist &d = G->cd[i];
    const int64_t tot = (int64_t)s * rpp * rp;
    hipLaunchKernelGGL(k_blocks_to_rows, dim3(cdiv(tot, 256)), dim3(256), 0, G->stream, (const float *)d.blocks_in, d.rows, s, nc, rpp, ncn, nd, rp);
    HIP_TRY(hipGetLastError());
    P3M_TRY(fft_x_forward_rows(c, G->plan_c, d.rows, d.ly, (int64_t)s * rpp, rpp));   // ROWS -> LY (local planes)
    if (!G->pencil) P3M_TRY(fft_slab_y_fwd(c, G->plan_c, d.ly, d.send, s));           // LY -> send layout [y][chunk][zl][16]
    else P3M_TRY(permute5(G->stream, d.ly, d.send, {s, nd, ncl, ncn, 1},              // [zl][iy'][cl][yl] -> [iy'][zl][cl][yl]
                          {(int64_t)ncl * ncn, (int64_t)s * ncl * ncn, ncn, 1, 0}));
  }
  if (G->pencil) {
    P3M_TRY(group_exchange(G, nd, xyb, [&](int r, int j) { return xy_peer(G, r, j); }, [&](int r) { return xy_index(G, r); },
                           [&](int li) { return (const char *)G->cd[li].send; }, [&](int li) { return (char *)G->cd[li].recv; }));
    for (int i = 0; i < nl; i++) {
      CoarseDist &d = G->cd[i];
      P3M_TRY(permute5(G->stream, d.recv, d.ly, {nd, s, ncl, ncn, 1}, {ncn, (int64_t)ncl * nc, nc, 1, 0}));   // [iy][zl][cl][yl] -> LY of the own chunks, all y
      P3M_TRY(fft_slab_y_fwd(G->ctx[i], G->plan_l, d.ly, d.send, s));
    }
  }
  // the y <-> z transpose: ky block j of every rank goes to the j-th rank of its group
  P3M_TRY(group_exchange(G, yz_count(G), (size_t)s * ncl * s * 16 * sizeof(float2), [&](int r, int j) { return yz_peer(G, r, j); },
                         [&](int r) { return yz_index(G, r); }, [&](int li) { return (const char *)G->cd[li].send; }, [&](int li) { return (char *)G->cd[li].recv; }));
  for (int i = 0; i < nl; i++) {
    p3m_ctx *c = G->ctx[i]; CoarseDist &d = G->cd[i];
    if (fft_has_segmented(G->plan_l)) { P3M_TRY(fft_slab_z_fwd(c, G->plan_l, d.recv, d.lz, s, s)); continue; }   // the z pass reads the arrivals where they lie
    hipLaunchKernelGGL(k_a2a_permute, dim3(cdiv((int64_t)NBc, 256)), dim3(256), 0, G->stream, reinterpret_cast<const float2 *>(d.recv),
                       reinterpret_cast<float2 *>(d.lz), s, ncl, nc, 1);
    HIP_TRY(hipGetLastError());
    P3M_TRY(fft_slab_z_fwd(c, G->plan_l, d.lz, d.lz, s, 0));                          // LZ in place: rho-hat(ky slab, all kz)
  }
  return P3M_OK;
}

static int build_coarse_kernel_dist(p3m_group *G, const float *table4_host) {
  const Geometry &g = G->ctx[0]->g;
  const int nl = (int)G->ctx.size(), s = G->s, nc = g.nc, ncn = g.ncn, nchunk = G->ncl;
  float *d_table = nullptr; P3M_TRY(galloc(&d_table, 192));
  HIP_TRY(hipMemcpyAsync(d_table, table4_host, sizeof(float) * 192, hipMemcpyHostToDevice, G->stream));
  const int64_t NBc = (int64_t)s * nchunk * nc * 16;
  const bool lr = (G->base.flags & P3M_FLAG_LRCKCORR) != 0;
  std::vector<float *> unc(nl, nullptr);
  if (lr) for (int i = 0; i < nl; i++) P3M_TRY(galloc(&unc[i], (size_t)NBc));
  for (int comp = 0; comp < 3; comp++) {
    for (int pass = lr ? 0 : 1; pass < 2; pass++) {   // pass 0: uncorrected analytic kernel (LRCKCORR only); pass 1: with the table
      for (int i = 0; i < nl; i++) {
        const Geometry &gi = G->ctx[i]->g;
        hipLaunchKernelGGL(k_ck_cube, dim3(cdiv((int64_t)ncn * ncn * ncn, 256)), dim3(256), 0, G->stream, G->ctx[i]->rho_c, (const float *)d_table, ncn, nc,
                           gi.cart[2] * ncn, gi.cart[1] * ncn, gi.cart[0] * ncn, g.ms, comp, pass);
        HIP_TRY(hipGetLastError());
      }
      P3M_TRY(dist_forward(G, [&](int li) { return G->ctx[li]->rho_c; }));
      for (int i = 0; i < nl; i++) {
        float *dst = pass == 0 ? unc[i] : G->cd[i].kern + (size_t)comp * (G->cstride / 2);
        hipLaunchKernelGGL(k_take_imag_g, dim3(cdiv(NBc, 256)), dim3(256), 0, G->stream, (const float *)G->cd[i].lz, dst, NBc);
        if (pass == 1 && lr)
          hipLaunchKernelGGL(k_lrck_slab, dim3(cdiv(NBc, 256)), dim3(256), 0, G->stream, dst, (const float *)unc[i], nc, s, nchunk, yz_index(G, G->lrank[i]) * s,
                             G->pencil ? xy_index(G, G->lrank[i]) * nchunk * 16 : 0, comp);
        HIP_TRY(hipGetLastError());
      }
    }
  }
  HIP_TRY(hipStreamSynchronize(G->stream));
  gfree(d_table);
  for (float *&u : unc) gfree(u);
  for (p3m_ctx *c : G->ctx) c->have_kc = true;
  return P3M_OK;
}

// coarse_force_buffer.f90 for all local ranks
static int coarse_force_halo(p3m_group *G) {
  const Geometry &g = G->ctx[0]->g;
  const int nl = (int)G->ctx.size(), nd = G->nd, ncn = g.ncn;
  const int m = ncn + 2; const size_t face = (size_t)3 * m * m;
  // one-cell halo: x, then y (carrying the x halo), then z (coarse_force_buffer.f90:19-63)
  for (int axis = 0; axis < 3; axis++) {
    if (G->batched) hipLaunchKernelGGL(k_halo_pack_b, dim3(cdiv((int64_t)face, 256), 2 * nl), dim3(256), 0, G->stream, (const float *)G->a_force_c, G->a_halo, ncn, axis);
    else
    for (int i = 0; i < nl; i++) {
      hipLaunchKernelGGL(k_halo_pack, dim3(cdiv((int64_t)face, 256)), dim3(256), 0, G->stream, (const float *)G->ctx[i]->force_c, G->cd[i].halo_s[0], ncn, axis, 1);    // to -axis
      hipLaunchKernelGGL(k_halo_pack, dim3(cdiv((int64_t)face, 256)), dim3(256), 0, G->stream, (const float *)G->ctx[i]->force_c, G->cd[i].halo_s[1], ncn, axis, ncn);  // to +axis
    }
    HIP_TRY(hipGetLastError());
    std::vector<XMsg> hm;
    for (int r = 0; r < G->nodes; r++) {
      const int c1 = r / (nd * nd), c2 = (r / nd) % nd, c3 = r % nd;
      int cp[3] = {c1, c2, c3}, cmn[3] = {c1, c2, c3};
      const int dim = 2 - axis;
      cp[dim] = (cp[dim] + 1) % nd; cmn[dim] = (cmn[dim] - 1 + nd) % nd;
      const int rpl = cp[0] * nd * nd + cp[1] * nd + cp[2], rmn = cmn[0] * nd * nd + cmn[1] * nd + cmn[2];
      const int li = G->lidx[r], lp = G->lidx[rpl], lm = G->lidx[rmn];
      hm.push_back({r, rmn, li >= 0 ? (const void *)G->cd[li].halo_s[0] : nullptr, lm >= 0 ? (void *)G->cd[lm].halo_r[0] : nullptr, face * sizeof(float)});  // plane 1 -> their ncn+1
      hm.push_back({r, rpl, li >= 0 ? (const void *)G->cd[li].halo_s[1] : nullptr, lp >= 0 ? (void *)G->cd[lp].halo_r[1] : nullptr, face * sizeof(float)});  // plane ncn -> their 0
    }
    P3M_TRY(do_exchange(G, hm));
    if (G->batched) hipLaunchKernelGGL(k_halo_unpack_b, dim3(cdiv((int64_t)face, 256), 2 * nl), dim3(256), 0, G->stream, G->a_force_c, (const float *)G->a_halo, ncn, axis);
    else
    for (int i = 0; i < nl; i++) {
      hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv((int64_t)face, 256)), dim3(256), 0, G->stream, G->ctx[i]->force_c, (const float *)G->cd[i].halo_r[0], ncn, axis, ncn + 1);
      hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv((int64_t)face, 256)), dim3(256), 0, G->stream, G->ctx[i]->force_c, (const float *)G->cd[i].halo_r[1], ncn, axis, 0);
    }
    HIP_TRY(hipGetLastError());
  }
  return P3M_OK;
}
// coarse_force.f90 + coarse_force_buffer.f90 + coarse_max_dt.f90 for all local ranks
static int coarse_force_dist(p3m_group *G) {
  const Geometry &g = G->ctx[0]->g;
  const int nl = (int)G->ctx.size(), nd = G->nd, s = G->s, nc = g.nc, ncn = g.ncn, rp = 2 * G->plan_c.px, ncl = G->ncl, rpp = G->rpp;
  const size_t NBc = (size_t)s * ncl * nc * 16, blk = (size_t)s * ncn * ncn;
  P3M_TRY(dist_forward(G, [&](int li) { return G->ctx[li]->rho_c; }));                  // coarse_force.f90:18
  const int64_t ccs = (int64_t)(G->cstride / 2);                                         // complex elements between two components of a rank
  if (G->direct) {   // no exchanges (see p3m_group::direct)
    p3m_ctx *c0 = G->ctx[0];
    P3M_TRY(fft_slab_z_inv3(c0, G->plan_l, G->a_lz, G->a_recv, G->a_kern, s, ccs, ccs, nl, (int64_t)NBc, true));
    P3M_TRY(fft_slab_y_inv(c0, G->plan_l, G->a_recv, G->a_ly, s, 3 * nl, s));
    RankPtrs red;
    for (int i = 0; i < nl; i++) red.p[i] = G->ctx[i]->d_red + 2 * P3M_RED_SPAN;
    const RowGeom q = row_geom(G);
    if (fft_x_has_cubes(G->plan_c, q)) P3M_TRY(fft_x_inverse_cubes(c0, G->plan_c, G->a_ly, G->a_force_c, q, red));   // the x pass stores into the owners' force arrays
    else {
      P3M_TRY(fft_x_inverse(c0, G->plan_c, G->a_ly, G->a_rows, -(3 * nl * s * rpp), 0, nullptr, 0, 0, 1, 0, rpp));
      hipLaunchKernelGGL(k_rows_to_force_b, dim3(cdiv((int64_t)nl * s * rpp, 4)), dim3(256), 0, G->stream, (const float *)G->a_rows, G->a_force_c, q, (int64_t)G->rstride, red);
      HIP_TRY(hipGetLastError());
    }
    return coarse_force_halo(G);
  }
  if (G->batched) P3M_TRY(fft_slab_z_inv3(G->ctx[0], G->plan_l, G->a_lz, G->a_send, G->a_kern, s, ccs, ccs, nl, (int64_t)NBc));
  else for (int i = 0; i < nl; i++)                                                      // :37-50 x3, fused multiply
    P3M_TRY(fft_slab_z_inv3(G->ctx[i], G->plan_l, G->cd[i].lz, G->cd[i].send, G->cd[i].kern, s, (int64_t)NBc, (int64_t)NBc));
  const size_t ab = (size_t)s * ncl * s * 16 * sizeof(float2);
  std::vector<XMsg> m2;                                                                  // transpose back, 3 components in one exchange
  for (int comp = 0; comp < 3; comp++) {
    const size_t co = (size_t)comp * ccs * sizeof(float2);
    group_msgs(G, m2, yz_count(G), ab, [&](int r, int j) { return yz_peer(G, r, j); }, [&](int r) { return yz_index(G, r); },
               [&](int li) { return (const char *)G->cd[li].send + co; }, [&](int li) { return (char *)G->cd[li].recv + co; });
  }
  P3M_TRY(do_exchange(G, m2));
  if (G->batched) {
    p3m_ctx *c0 = G->ctx[0];
    P3M_TRY(fft_slab_y_inv(c0, G->plan_l, G->a_recv, G->a_ly, s, 3 * nl, s));
    P3M_TRY(fft_x_inverse(c0, G->plan_c, G->a_ly, G->a_rows, -(3 * nl * s * rpp), 0, nullptr, 0, 0, 1, 0, rpp));
    hipLaunchKernelGGL(k_rows_to_blocks_b, dim3(cdiv((int64_t)3 * nl * s * rpp, 4)), dim3(256), 0, G->stream, (const float *)G->a_rows, G->a_blocks_out, row_geom(G));
    HIP_TRY(hipGetLastError());
  } else
  for (int i = 0; i < nl; i++) {
    p3m_ctx *c = G->ctx[i]; CoarseDist &d = G->cd[i];
    if (fft_has_segmented(G->plan_l)) P3M_TRY(fft_slab_y_inv(c, G->plan_l, d.recv, d.ly, s, 3, s));
    else {
      hipLaunchKernelGGL(k_a2a_permute, dim3(cdiv((int64_t)3 * NBc, 256)), dim3(256), 0, G->stream, reinterpret_cast<const float2 *>(d.recv),
                         reinterpret_cast<float2 *>(d.ly), s, ncl, nc, 3);
      HIP_TRY(hipGetLastError());
      P3M_TRY(fft_slab_y_inv(c, G->plan_l, d.ly, d.ly, s, 3, 0));
    }
    if (G->pencil) P3M_TRY(permute5(G->stream, d.ly, d.send, {3 * s, ncl, nd, ncn, 1},  // [comp,zl][cl][iy'][yl] -> [iy'][comp,zl][cl][yl]
                                    {(int64_t)ncl * ncn, ncn, (int64_t)3 * s * ncl * ncn, 1, 0}));
  }
  if (G->pencil) {                                                                       // y -> x transpose of the three components
    P3M_TRY(group_exchange(G, nd, (size_t)3 * s * ncl * ncn * 16 * sizeof(float2), [&](int r, int j) { return xy_peer(G, r, j); },
                           [&](int r) { return xy_index(G, r); }, [&](int li) { return (const char *)G->cd[li].send; }, [&](int li) { return (char *)G->cd[li].recv; }));
    for (int i = 0; i < nl; i++)                                                         // [iy][comp,zl][cl][yl] -> LY [comp,zl][chunk = iy*ncl + cl][yl]
      P3M_TRY(permute5(G->stream, G->cd[i].recv, G->cd[i].ly, {nd, 3 * s, ncl, ncn, 1}, {(int64_t)ncl * ncn, (int64_t)G->nchunk * ncn, ncn, 1, 0}));
  }
  if (!G->batched)
  for (int i = 0; i < nl; i++) {
    p3m_ctx *c = G->ctx[i]; CoarseDist &d = G->cd[i];
    P3M_TRY(fft_x_inverse(c, G->plan_c, d.ly, d.rows, -(3 * s * rpp), 0, nullptr, 0, 0, 1, 0, rpp));   // incl. /nc^3 (fftw3ds.f90:161, p3dfft_coarse.f90:57)
    const int64_t tot = (int64_t)G->nxb * 3 * blk;
    hipLaunchKernelGGL(k_rows_to_blocks, dim3(cdiv(tot, 256)), dim3(256), 0, G->stream, (const float *)d.rows, d.blocks_out, s, rpp / ncn, ncn, nd, rp);
    HIP_TRY(hipGetLastError());
  }
  // x-lines -> cube (unpack_slab, fftw3ds.f90:69-99; unpack_pencils, p3dfft_coarse.f90:129-183): the pack exchange backwards
  {
    std::vector<XMsg> m3;
    for (int r = 0; r < G->nodes; r++)
      for (int q = 0; q < G->nxb; q++) {
        const int t = xl_peer(G, r, q), lt = G->lidx[t], lr = G->lidx[r];   // t holds z-slice q of r's cube as block xl_index(r)
        m3.push_back({t, r, lt >= 0 ? (const void *)(G->cd[lt].blocks_out + (size_t)xl_index(G, r) * 3 * blk) : nullptr,
                      lr >= 0 ? (void *)(G->cd[lr].blocks_back + (size_t)q * 3 * blk) : nullptr, 3 * blk * sizeof(float)});
      }
    P3M_TRY(do_exchange(G, m3));
  }
  if (G->batched) {   // the maximum (coarse_max_dt.f90) rides on this copy
    RankPtrs red;
    for (int i = 0; i < nl; i++) red.p[i] = G->ctx[i]->d_red + 2 * P3M_RED_SPAN;
    hipLaunchKernelGGL(k_blocks_to_force_b, dim3(cdiv((int64_t)nl * G->nxb * s * ncn, 4)), dim3(256), 0, G->stream, (const float *)G->a_blocks_back, G->a_force_c, nl, G->nxb, s,
                       ncn, fdiv_magic(ncn), fdiv_magic(s), red);
    HIP_TRY(hipGetLastError());
  } else
  for (int i = 0; i < nl; i++) {
    const int64_t tot = (int64_t)G->nxb * 3 * blk;
    hipLaunchKernelGGL(k_blocks_to_force, dim3(cdiv(tot, 256)), dim3(256), 0, G->stream, (const float *)G->cd[i].blocks_back, G->ctx[i]->force_c, s, ncn, G->nxb);
    HIP_TRY(hipGetLastError());
  }
  P3M_TRY(coarse_force_halo(G));
  if (!G->batched)
  for (int i = 0; i < nl; i++) {
    hipLaunchKernelGGL(k_gmax_interior, dim3(std::min<int64_t>(1024, cdiv((int64_t)ncn * ncn * ncn, 256))), dim3(256), 0, G->stream, (const float *)G->ctx[i]->force_c, ncn,
                       G->ctx[i]->d_red + 2 * P3M_RED_SPAN);
    HIP_TRY(hipGetLastError());
  }
  return P3M_OK;
}

// ================================================================== public group API
extern "C" int p3m_hip_group_set_kernel_tables(p3m_group *G, const float *fine_table, const float *coarse_table) {
  const bool coarse_only = G && (G->base.flags & P3M_FLAG_COARSE_ONLY);
  if (!G || (!fine_table && !coarse_only) || !coarse_table) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(G->device));
  if (G->nodes == 1) return p3m_hip_set_kernel_tables(G->ctx[0], fine_table, coarse_table);
  if (!coarse_only) {
    // kern_f is identical on every rank: build once, copy
    P3M_TRY(build_fine_kernel(G->ctx[0], fine_table));
    const Geometry &g = G->ctx[0]->g;
    for (size_t i = 1; i < G->ctx.size(); i++) {
      HIP_TRY(hipMemcpyAsync(G->ctx[i]->kern_f, G->ctx[0]->kern_f, sizeof(float) * 3 * g.nf * g.nf * g.px, hipMemcpyDeviceToDevice, G->stream));
      G->ctx[i]->have_kf = true; G->ctx[i]->kf_zmirror = G->ctx[0]->kf_zmirror;
    }
  }
  P3M_TRY(build_coarse_kernel_dist(G, coarse_table));
  G->have_k = true;
  return P3M_OK;
}
// kern_c z-slabs of the reference (kern_c(3, nc/2+1, nc, nc_slab), kernel_checkpoint.f90) -> every rank's ky slab in the bundle layout:
// recv [q][comp][kyl][kzl][px] (q: the rank whose z-slab the block came from) -> kern [comp][kyl][chunk][q*s + kzl][16]
__global__ __launch_bounds__(256) void k_kern_scatter(const float *__restrict__ recv, float *__restrict__ kern, int nodes, int s, int px, int nc, int64_t kcs) {
  const int64_t per = (int64_t)3 * s * s * px, idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per * nodes) return;
  const int q = (int)(idx / per); int64_t r = idx - q * per;
  const int x = (int)(r % px); r /= px; const int kzl = (int)(r % s); r /= s; const int kyl = (int)(r % s); const int comp = (int)(r / s);
  kern[comp * kcs + (((int64_t)kyl * (px / 16) + x / 16) * nc + (q * s + kzl)) * 16 + x % 16] = recv[idx];
}
int kernels_set_fine_raw(p3m_ctx *c, const float *kern_f);   // p3m_api.hip
extern "C" int p3m_hip_group_set_kernels_raw(p3m_group *G, const float *kern_f, const float *const *kern_c_slabs) {
  if (!G || !kern_f || !kern_c_slabs) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(G->device));
  if (G->nodes == 1) return p3m_hip_set_kernels_raw(G->ctx[0], kern_f, kern_c_slabs[0]);
  if (G->pencil) { p3m_set_error("set_kernels_raw: kern_c z-slabs do not map onto the pencil decomposition (build the kernels from the tables)"); return P3M_EINVAL; }
  if (!(G->base.flags & P3M_FLAG_COARSE_ONLY)) {
    P3M_TRY(kernels_set_fine_raw(G->ctx[0], kern_f));
    const Geometry &g0 = G->ctx[0]->g;
    for (size_t i = 1; i < G->ctx.size(); i++) {
      HIP_TRY(hipMemcpyAsync(G->ctx[i]->kern_f, G->ctx[0]->kern_f, sizeof(float) * 3 * g0.nf * g0.nf * g0.px, hipMemcpyDeviceToDevice, G->stream));
      G->ctx[i]->have_kf = true; G->ctx[i]->kf_zmirror = G->ctx[0]->kf_zmirror;
    }
  }
  const Geometry &g = G->ctx[0]->g;
  const int nl = (int)G->ctx.size(), s = G->s, nc = g.nc, hx = nc / 2 + 1, px = G->plan_c.px;
  const size_t blkf = (size_t)3 * s * s * px;                        // floats of one (source slab, destination slab) block
  if ((size_t)G->nodes * blkf > 3 * (size_t)s * G->ncl * nc * 16 * 2) { p3m_set_error("set_kernels_raw: staging too small"); return P3M_EINVAL; }
  std::vector<float> host((size_t)G->nodes * blkf);
  for (int i = 0; i < nl; i++) {
    const float *slab = kern_c_slabs[i];
    if (!slab) return P3M_EINVAL;
    std::fill(host.begin(), host.end(), 0.f);
    for (int t = 0; t < G->nodes; t++)
      for (int comp = 0; comp < 3; comp++) for (int kyl = 0; kyl < s; kyl++) for (int kzl = 0; kzl < s; kzl++) {
        float *dst = host.data() + (size_t)t * blkf + (((size_t)comp * s + kyl) * s + kzl) * px;
        const float *src = slab + (((size_t)kzl * nc + (size_t)(t * s + kyl)) * hx) * 3 + comp;   // kern_c(comp, kx, ky, kz local), component fastest
        for (int x = 0; x < hx; x++) dst[x] = src[(size_t)x * 3];
      }
    HIP_TRY(hipMemcpyAsync(G->cd[i].send, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice, G->stream));
    HIP_TRY(hipStreamSynchronize(G->stream));
  }
  P3M_TRY(group_exchange(G, G->nodes, blkf * sizeof(float), [&](int r, int j) { (void)r; return j; }, [&](int r) { return r; },
                         [&](int li) { return (const char *)G->cd[li].send; }, [&](int li) { return (char *)G->cd[li].recv; }));
  for (int i = 0; i < nl; i++) {
    hipLaunchKernelGGL(k_kern_scatter, dim3(cdiv((int64_t)G->nodes * blkf, 256)), dim3(256), 0, G->stream, (const float *)G->cd[i].recv, G->cd[i].kern, G->nodes, s, px, nc,
                       (int64_t)(G->cstride / 2));
    HIP_TRY(hipGetLastError());
    G->ctx[i]->have_kc = true;
  }
  HIP_TRY(hipStreamSynchronize(G->stream));
  G->have_k = true;
  return P3M_OK;
}
extern "C" int p3m_hip_group_upload_particles(p3m_group *G, int32_t i, const float *xv6, const int64_t *pid, int32_t np) {
  if (!G || i < 0 || i >= (int)G->ctx.size()) return P3M_EINVAL;
  return p3m_hip_upload_particles(G->ctx[i], xv6, pid, np);
}
extern "C" int p3m_hip_group_download_particles(p3m_group *G, int32_t i, float *xv6, int64_t *pid, int32_t *np) {
  if (!G || i < 0 || i >= (int)G->ctx.size()) return P3M_EINVAL;
  return p3m_hip_download_particles(G->ctx[i], xv6, pid, np);
}

static int reduce_step_out(p3m_group *G, float a_mid, p3m_step_out *out, bool downloaded = false) {
  const Geometry &g = G->ctx[0]->g;
  float v[4] = {0, 0, 0, 0}; double sums[3] = {0, 0, 0}; int ng = 0, ndel = 0;
  if (!downloaded) {
    for (p3m_ctx *c : G->ctx) {
      P3M_TRY(reductions_download(c));   // (with the per-tile PP maxima: one block)
    }
    HIP_TRY(hipStreamSynchronize(G->stream));
  }
  for (p3m_ctx *c : G->ctx) {
    reductions_fold(c);
    P3M_TRY(rho_u8_check(c));
    v[0] = std::max(v[0], sqrtf(c->h_red[0])); v[1] = std::max(v[1], c->h_red[1]); v[3] = std::max(v[3], c->h_red[2]);
    if (c->p.flags & P3M_FLAG_PP_EXT) {   // per-thread "last tile" overwrite, particle_mesh_threaded.f90:617
      const int cores = std::max(1, c->p.cores), nt = std::min(cores, g.ntiles), base = g.ntiles / nt, rem = g.ntiles % nt;
      int pos = 0;
      for (int t = 0; t < nt; t++) { pos += base + (t < rem ? 1 : 0); v[2] = std::max(v[2], c->h_tile_ext[pos - 1]); }
    }
    sums[0] += c->h_sums[0]; sums[1] += c->h_sums[1]; sums[2] += (double)c->np_local; ng += c->np_ghost; ndel += c->np_deleted;
  }
  if (G->nprocs > 1 && !G->comm && G->have_tr) {   // host transport: the values are on the host already
    if (G->tr.allreduce_max_f32(G->tr.user, v, 4) || G->tr.allreduce_sum_f64(G->tr.user, sums, 3)) { p3m_set_error("host transport: all-reduce callback failed"); return P3M_ECOMM; }
  } else if (G->nprocs > 1) {   // mpi_reduce + mpi_bcast pairs (:646-696, coarse_max_dt.f90:34-37) as two all-reduces
    if (!G->comm) { p3m_set_error("group reduction between processes needs RCCL or a host transport"); return P3M_ECOMM; }
    memcpy(G->h_red4, v, sizeof(v)); memcpy(G->h_sum3, sums, sizeof(sums));
    HIP_TRY(hipMemcpyAsync(G->d_red4, G->h_red4, 4 * sizeof(float), hipMemcpyHostToDevice, G->stream));
    HIP_TRY(hipMemcpyAsync(G->d_sum3, G->h_sum3, 3 * sizeof(double), hipMemcpyHostToDevice, G->stream));
    NCCL_TRY(ncclAllReduce(G->d_red4, G->d_red4, 4, ncclFloat, ncclMax, G->comm, G->stream));
    NCCL_TRY(ncclAllReduce(G->d_sum3, G->d_sum3, 3, ncclDouble, ncclSum, G->comm, G->stream));
    HIP_TRY(hipMemcpyAsync(G->h_red4, G->d_red4, 4 * sizeof(float), hipMemcpyDeviceToHost, G->stream));
    HIP_TRY(hipMemcpyAsync(G->h_sum3, G->d_sum3, 3 * sizeof(double), hipMemcpyDeviceToHost, G->stream));
    HIP_TRY(hipStreamSynchronize(G->stream));
    memcpy(v, G->h_red4, sizeof(v)); memcpy(sums, G->h_sum3, sizeof(sums));
  }
  p3m_step_out o; memset(&o, 0, sizeof(o));
  const uint32_t fl = G->base.flags;
  o.f_force_max = v[0]; o.pp_force_max = v[1]; o.pp_ext_force_max = v[2]; o.c_force_max = v[3];
  o.dt_f_acc = 1.0f / sqrtf(fmaxf(0.0001f, v[0]) * a_mid * P3M_G_F);
  o.dt_pp_acc = (fl & P3M_FLAG_PPINT) ? sqrtf(G->base.dt_pp_scale * G->base.rsoft) / fmaxf(sqrtf(v[1] * a_mid * P3M_G_F), 1e-3f) : G->last.dt_pp_acc;
  o.dt_pp_ext_acc = (fl & P3M_FLAG_PP_EXT) ? sqrtf(G->base.dt_pp_scale * G->base.rsoft) / fmaxf(sqrtf(v[2] * a_mid * P3M_G_F), 1e-3f) : G->last.dt_pp_ext_acc;
  o.dt_c_acc = sqrtf((float)g.ms / (v[3] * a_mid * P3M_G_F));
  o.sum_rho_f = sums[0]; o.sum_rho_c = sums[1]; o.np_total = (int64_t)llround(sums[2]);
  o.np_local = G->ctx[0]->np_local; o.np_ghost = ng; o.np_deleted = ndel;
  G->last = o; *out = o;
  return P3M_OK;
}

// per-phase GPU times of the last whole step of this process's ranks (timers.f90:68-77; the reference prints max / avg / min over ranks)
extern "C" int p3m_hip_group_phase_timing(p3m_group *G, int32_t on) {
  if (!G) return P3M_EINVAL;
  if (G->nodes == 1) return p3m_hip_phase_timing(G->ctx[0], on);
  G->pt.on = on != 0; G->pt.reset();
  for (p3m_ctx *c : G->ctx) c->pt = &G->pt;
  return P3M_OK;
}
extern "C" int p3m_hip_group_last_phase_ms(p3m_group *G, float *ms12) {
  if (!G || !ms12) return P3M_EINVAL;
  if (G->nodes == 1) return p3m_hip_last_phase_ms(G->ctx[0], ms12);
  if (!G->pt.on) { p3m_set_error("p3m_hip_group_last_phase_ms: phase timing is off (p3m_hip_group_phase_timing)"); return P3M_ESTATE; }
  for (int k = 0; k < P3M_NPHASE; k++) ms12[k] = G->pt.ms[k];
  return P3M_OK;
}

extern "C" int p3m_hip_group_update_position(p3m_group *G, float dt, float dt_old, const float *offset) {
  if (!G) return P3M_EINVAL;
  P3M_TRY(need_particles(G->ctx[0], "p3m_hip_group_update_position"));
  HIP_TRY(hipSetDevice(G->device));
  for (p3m_ctx *c : G->ctx) P3M_TRY(particles_drift(c, dt, dt_old, offset));
  return P3M_OK;
}

// subroutine particle_mesh on every local rank (particle_mesh_threaded.f90:2-726)
static int group_particle_mesh_step(p3m_group *G, float a_mid, float dt, float dt_old, float mass_p, const float *offset, const float *move_back, p3m_step_out *out);
extern "C" int p3m_hip_group_particle_mesh(p3m_group *G, float a_mid, float dt, float dt_old, float mass_p, const float *offset,
                                           const float *move_back, p3m_step_out *out) {
  if (!G) return P3M_EINVAL;
  const int r = group_particle_mesh_step(G, a_mid, dt, dt_old, mass_p, offset, move_back, out);
  if (r != P3M_OK && r != P3M_ESTATE && G->nodes > 1) {   // e.g. P3M_ECAPACITY in the ghost pass: images may already be counted into the row histogram
    for (p3m_ctx *c : G->ctx) particles_reset_after_error(c);
    if (G->stream2) (void)hipStreamSynchronize(G->stream2);
  }
  return r;
}
static int group_particle_mesh_step(p3m_group *G, float a_mid, float dt, float dt_old, float mass_p, const float *offset, const float *move_back, p3m_step_out *out) {
  HIP_TRY(hipSetDevice(G->device));
  if (G->nodes == 1) return p3m_hip_particle_mesh(G->ctx[0], a_mid, dt, dt_old, mass_p, offset, move_back, out);
  if (G->base.flags & P3M_FLAG_COARSE_ONLY) { p3m_set_error("particle_mesh on a P3M_FLAG_COARSE_ONLY group (it holds the coarse mesh only)"); return P3M_ESTATE; }
  for (p3m_ctx *c : G->ctx) if (!c->have_kf || !c->have_kc) { p3m_set_error("particle_mesh before the Green's functions were set"); return P3M_ESTATE; }
  G->pt.reset();
  { PhaseScope ps(&G->pt, P3M_PH_DRIFT, G->stream); for (p3m_ctx *c : G->ctx) P3M_TRY(particles_drift(c, dt, dt_old, offset)); }                      // :56
  { PhaseScope ps(&G->pt, P3M_PH_GHOST, G->stream); P3M_TRY(ghost_pass(G)); }                                                                          // :61-63
  // ---- fork: from here to the step's one host wait every local rank queues on its own stream (see p3m_group::rstream)
  const bool multi = G->multi && !G->pt.on;
  hipStream_t const main_stream = G->stream;
  struct Rejoin {   // whatever happens below: the contexts are back on the group's stream and nothing of this step is still running on theirs
    p3m_group *G; hipStream_t main; bool armed;
    ~Rejoin() { if (!armed) return; for (size_t i = 0; i < G->ctx.size(); i++) { (void)hipStreamSynchronize(G->rstream[i % G->nrs]); G->ctx[i]->stream = main; } G->stream = main; }
  } rejoin{G, main_stream, multi};
  if (multi) {
    HIP_TRY(hipEventRecord(G->ev_fork, main_stream));
    for (size_t i = 0; i < G->ctx.size(); i++) { HIP_TRY(hipStreamWaitEvent(G->rstream[i % G->nrs], G->ev_fork, 0)); G->ctx[i]->stream = G->rstream[i % G->nrs]; }
  }
  { PhaseScope ps(&G->pt, P3M_PH_SORT, G->stream); for (p3m_ctx *c : G->ctx) { P3M_TRY(step_prezero(c)); P3M_TRY(particles_sort_enqueue(c, mass_p)); } }   // every rank's sort queued ...
  for (p3m_ctx *c : G->ctx) P3M_TRY(particles_sort_finish(c, false));                                      // ... and nobody waits: the counters come in with the step's results
  // The coarse force depends on positions only: it is formed right after the sort, on a second stream underneath the
  // fine-mesh force sweeps (many small kernels and the all-to-all exchanges against bandwidth-bound FFT passes).  Its kick
  // is then applied inside the fine kick's pass (coarse_kick_rides_on_fine, p3m_api.hip).
  const bool ride = !G->ctx.empty() && coarse_kick_rides_on_fine(G->ctx[0]);
  { PhaseScope ps(&G->pt, P3M_PH_COARSE_DEPOSIT, G->stream); for (p3m_ctx *c : G->ctx) P3M_TRY(coarse_deposit(c, mass_p)); }                          // coarse_mass
  if (G->stream2) {
    if (multi) { for (size_t i = 0; i < G->ctx.size(); i++) HIP_TRY(hipEventRecord(G->rev_dep[i], G->rstream[i % G->nrs])); }
    else HIP_TRY(hipEventRecord(G->ev_dep, G->stream));
    for (p3m_ctx *c : G->ctx) P3M_TRY(fine_mesh_force_phase(c, mass_p, false));                     // :72-204 of every tile, queued first
    if (multi) { for (size_t i = 0; i < G->ctx.size(); i++) HIP_TRY(hipStreamWaitEvent(G->stream2, G->rev_dep[i], 0)); }   // the coarse force needs every rank's coarse density
    else HIP_TRY(hipStreamWaitEvent(G->stream2, G->ev_dep, 0));
    G->stream = G->stream2; for (p3m_ctx *c : G->ctx) c->stream = G->stream2;
    int r;
    { PhaseScope ps(&G->pt, P3M_PH_COARSE_FORCE, G->stream2); r = coarse_force_dist(G); }                                         // coarse_force, _buffer, max
    if (r == P3M_OK && hipEventRecord(G->ev_cf, G->stream2) != hipSuccess) r = P3M_EDEVICE;
    G->stream = main_stream; for (size_t i = 0; i < G->ctx.size(); i++) G->ctx[i]->stream = multi ? G->rstream[i % G->nrs] : main_stream;
    if (r != P3M_OK) { (void)hipStreamSynchronize(G->stream2); return r; }
  } else {
    { PhaseScope ps(&G->pt, P3M_PH_COARSE_FORCE, G->stream); P3M_TRY(coarse_force_dist(G)); }
    for (p3m_ctx *c : G->ctx) P3M_TRY(fine_mesh_force_phase(c, mass_p, false));
  }
  auto wait_cf = [&]() {   // the kicks need the coarse force
    if (!G->stream2) return P3M_OK;
    if (multi) { for (size_t i = 0; i < G->ctx.size(); i++) HIP_TRY(hipStreamWaitEvent(G->rstream[i % G->nrs], G->ev_cf, 0)); }
    else HIP_TRY(hipStreamWaitEvent(G->stream, G->ev_cf, 0));
    return P3M_OK;
  };
  if (ride) {
    P3M_TRY(wait_cf());
    for (p3m_ctx *c : G->ctx) {
      c->coarse_first = true;
      const int r = fine_mesh_kick_phase(c, a_mid, dt, mass_p);                                     // :208-319 + coarse_velocity
      c->coarse_first = false;
      P3M_TRY(r);
    }
  } else {
    for (p3m_ctx *c : G->ctx) P3M_TRY(fine_mesh_kick_phase(c, a_mid, dt, mass_p));                  // :208-628
    P3M_TRY(wait_cf());
    { PhaseScope ps(&G->pt, P3M_PH_COARSE_KICK, G->stream); for (p3m_ctx *c : G->ctx) P3M_TRY(coarse_kick(c, a_mid, dt)); }                          // coarse_velocity
  }
  { PhaseScope ps(&G->pt, P3M_PH_DELETE, G->stream); for (p3m_ctx *c : G->ctx) P3M_TRY(particles_finalize_enqueue(c, (G->base.flags & P3M_FLAG_MOVE_GRID_BACK) ? move_back : nullptr)); }  // :716-720
  // ONE host wait for everything the host reads back: survivor counts, the sort's counters, maxima and sums
  for (p3m_ctx *c : G->ctx) P3M_TRY(reductions_download(c));   // (with the per-tile PP maxima: one block)
  for (p3m_ctx *c : G->ctx) c->step_zeroed = false;
  if (multi) {   // ---- join: the group's stream waits for every rank's
    for (size_t i = 0; i < G->ctx.size(); i++) { HIP_TRY(hipEventRecord(G->rev_done[i], G->rstream[i % G->nrs])); HIP_TRY(hipStreamWaitEvent(main_stream, G->rev_done[i], 0)); G->ctx[i]->stream = main_stream; }
    rejoin.armed = false;
  }
  HIP_TRY(hipStreamSynchronize(G->stream));
  if (G->pt.on) { if (G->stream2) HIP_TRY(hipStreamSynchronize(G->stream2)); G->pt.collect(); }
  for (p3m_ctx *c : G->ctx) P3M_TRY(particles_finalize_finish(c, false));
  p3m_step_out o;
  P3M_TRY(reduce_step_out(G, a_mid, &o, true));
  if (out) *out = o;
  return P3M_OK;
}

// projection.f90 at a projection step (cubepm.f90:193-228): link_list + particle_pass, projection, delete_particles
int projection_rank(p3m_ctx *c, float mass_p, float *d_pxy, float *d_pxz, float *d_pyz, double *rho_node);   // p3m_api.hip
extern "C" int p3m_hip_group_projection(p3m_group *G, float mass_p, float *pxy, float *pxz, float *pyz, double *rho_tot) {
  if (!G || !pxy || !pxz || !pyz) return P3M_EINVAL;
  P3M_TRY(need_particles(G->ctx[0], "p3m_hip_group_projection"));
  HIP_TRY(hipSetDevice(G->device));
  const Geometry &g = G->ctx[0]->g;
  const size_t n2 = (size_t)g.Nn * g.nodes_dim * g.Nn * g.nodes_dim;
  float *d = nullptr;
  HIP_TRY(hipMalloc(&d, 3 * n2 * sizeof(float)));
  auto body = [&]() -> int {
    HIP_TRY(hipMemsetAsync(d, 0, 3 * n2 * sizeof(float), G->stream));
    if (G->nodes == 1) P3M_TRY(p3m_hip_link_list_and_pass(G->ctx[0]));
    else { P3M_TRY(ghost_pass(G)); for (p3m_ctx *c : G->ctx) P3M_TRY(particles_sort(c, -1.f)); }
    double tot = 0.0;
    for (p3m_ctx *c : G->ctx) { double t = 0.0; P3M_TRY(projection_rank(c, mass_p, d, d + n2, d + 2 * n2, &t)); tot += t; }
    for (p3m_ctx *c : G->ctx) P3M_TRY(particles_finalize(c, nullptr));
    if (G->nodes > 1 && G->nprocs > 1) {
      if (!G->comm && G->have_tr) { if (G->tr.allreduce_sum_f64(G->tr.user, &tot, 1)) return P3M_ECOMM; }
      else if (!G->comm) { p3m_set_error("group reduction between processes needs RCCL or a host transport"); return P3M_ECOMM; }
      else {
        HIP_TRY(hipMemcpyAsync(G->d_sum3, &tot, sizeof(double), hipMemcpyHostToDevice, G->stream));
        NCCL_TRY(ncclAllReduce(G->d_sum3, G->d_sum3, 1, ncclDouble, ncclSum, G->comm, G->stream));
        HIP_TRY(hipMemcpyAsync(&tot, G->d_sum3, sizeof(double), hipMemcpyDeviceToHost, G->stream));
        HIP_TRY(hipStreamSynchronize(G->stream));
      }
    }
    if (rho_tot) *rho_tot = tot;
    HIP_TRY(hipMemcpy(pxy, d, n2 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pxz, d + n2, n2 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pyz, d + 2 * n2, n2 * sizeof(float), hipMemcpyDeviceToHost));
    return P3M_OK;
  };
  const int r = body();
  (void)hipFree(d);
  return r;
}

// probes for the parity tests: local coarse density / force of local rank i in the reference layout
extern "C" int p3m_hip_group_probe_coarse(p3m_group *G, float mass_p, int32_t i, float *rho_c, float *force_c) {
  if (!G || i < 0 || i >= (int)G->ctx.size()) return P3M_EINVAL;
  P3M_TRY(need_particles(G->ctx[0], "p3m_hip_group_probe_coarse"));   // it deposits the records; a coarse-only group takes p3m_hip_group_set_coarse_density
  HIP_TRY(hipSetDevice(G->device));
  const Geometry &g = G->ctx[0]->g;
  if (G->nodes == 1) return p3m_hip_probe_coarse(G->ctx[0], mass_p, rho_c, force_c);
  for (p3m_ctx *c : G->ctx) { P3M_TRY(reductions_clear(c)); P3M_TRY(coarse_deposit(c, mass_p)); }
  if (rho_c) HIP_TRY(hipMemcpyAsync(rho_c, G->ctx[i]->rho_c, sizeof(float) * (size_t)g.ncn * g.ncn * g.ncn, hipMemcpyDeviceToHost, G->stream));
  if (force_c) {
    P3M_TRY(coarse_force_dist(G));
    const size_t fcs = (size_t)(g.ncn + 2) * (g.ncn + 2) * (g.ncn + 2);
    std::vector<float> tmp(3 * fcs);
    HIP_TRY(hipMemcpyAsync(tmp.data(), G->ctx[i]->force_c, sizeof(float) * 3 * fcs, hipMemcpyDeviceToHost, G->stream));
    HIP_TRY(hipStreamSynchronize(G->stream));
    for (int comp = 0; comp < 3; comp++) for (size_t k = 0; k < fcs; k++) force_c[k * 3 + comp] = tmp[comp * fcs + k];
  }
  HIP_TRY(hipStreamSynchronize(G->stream));
  return P3M_OK;
}


// ------------------------------------------------------------------ the distributed coarse transform on its own
static int need_dist(p3m_group *G, int32_t i) {
  if (!G || i < 0 || i >= (int)G->ctx.size()) return P3M_EINVAL;
  if (G->nodes == 1) { p3m_set_error("the distributed coarse transform needs a multi-rank group (nodes_dim > 1)"); return P3M_EINVAL; }
  return P3M_OK;
}
extern "C" int p3m_hip_group_set_coarse_density(p3m_group *G, int32_t i, const float *rho_c) {
  P3M_TRY(need_dist(G, i));
  if (!rho_c) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(G->device));
  const Geometry &g = G->ctx[0]->g;
  HIP_TRY(hipMemcpyAsync(G->ctx[i]->rho_c, rho_c, sizeof(float) * (size_t)g.ncn * g.ncn * g.ncn, hipMemcpyHostToDevice, G->stream));
  HIP_TRY(hipStreamSynchronize(G->stream));
  return P3M_OK;
}
extern "C" int p3m_hip_group_coarse_transform(p3m_group *G, int32_t what, int32_t reps, float *ms) {
  P3M_TRY(need_dist(G, 0));
  if (what < 0 || what > 1 || reps < 0) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(G->device));
  if (what == 1) for (p3m_ctx *c : G->ctx) if (!c->have_kc) { p3m_set_error("coarse force before the coarse kernel was set"); return P3M_ESTATE; }
  auto once = [&]() -> int {
    if (what == 0) return dist_forward(G, [&](int li) { return G->ctx[li]->rho_c; });
    for (p3m_ctx *c : G->ctx) HIP_TRY(hipMemsetAsync(c->d_red + 2 * P3M_RED_SPAN, 0, P3M_RED_SPAN * sizeof(float), G->stream));
    return coarse_force_dist(G);
  };
  P3M_TRY(once());
  if (reps > 0) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int r = hipEventRecord(e0, G->stream) == hipSuccess ? P3M_OK : P3M_EDEVICE;
    for (int k = 0; k < reps && r == P3M_OK; k++) r = once();
    if (r == P3M_OK && (hipEventRecord(e1, G->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) r = P3M_EDEVICE;
    float t = 0.f;
    if (r == P3M_OK && hipEventElapsedTime(&t, e0, e1) != hipSuccess) r = P3M_EDEVICE;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    P3M_TRY(r);
    if (ms) *ms = t / (float)reps;
  }
  HIP_TRY(hipStreamSynchronize(G->stream));
  return P3M_OK;
}
extern "C" int p3m_hip_group_get_coarse_hat(p3m_group *G, int32_t i, float *hat, int64_t nfloats) {
  P3M_TRY(need_dist(G, i));
  const Geometry &g = G->ctx[0]->g;
  const int64_t want = (int64_t)G->s * G->ncl * g.nc * 16 * 2;
  if (!hat || nfloats != want) { p3m_set_error("get_coarse_hat: the local rho-hat holds %lld floats ([s=%d][ncl=%d][nc=%d][16] complex)", (long long)want, G->s, G->ncl, g.nc); return P3M_EINVAL; }
  HIP_TRY(hipSetDevice(G->device));
  HIP_TRY(hipMemcpyAsync(hat, G->cd[i].lz, sizeof(float) * (size_t)want, hipMemcpyDeviceToHost, G->stream));
  HIP_TRY(hipStreamSynchronize(G->stream));
  return P3M_OK;
}
extern "C" int p3m_hip_group_get_coarse_force(p3m_group *G, int32_t i, float *force_c) {
  P3M_TRY(need_dist(G, i));
  if (!force_c) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(G->device));
  const Geometry &g = G->ctx[0]->g;
  const size_t fcs = (size_t)(g.ncn + 2) * (g.ncn + 2) * (g.ncn + 2);
  std::vector<float> tmp(3 * fcs);
  HIP_TRY(hipMemcpyAsync(tmp.data(), G->ctx[i]->force_c, sizeof(float) * 3 * fcs, hipMemcpyDeviceToHost, G->stream));
  HIP_TRY(hipStreamSynchronize(G->stream));
  for (int comp = 0; comp < 3; comp++) for (size_t k = 0; k < fcs; k++) force_c[k * 3 + comp] = tmp[comp * fcs + k];
  return P3M_OK;
}
extern "C" int64_t p3m_hip_group_coarse_exchange_bytes(const p3m_group *G) {
  if (!G || G->nodes == 1) return 0;
  return (int64_t)G->s * G->ncl * G->s * 16 * (int64_t)sizeof(float2);
}

extern "C" int32_t p3m_hip_coarse_fft_schedule(int32_t nodes_dim, uint32_t flags, int32_t rank, int32_t which, int32_t j, int32_t *peer, int32_t *index) {
  if (nodes_dim < 1 || !peer || !index) return P3M_EINVAL;
  p3m_group G;
  G.nd = nodes_dim; G.nodes = nodes_dim * nodes_dim * nodes_dim; G.pencil = (flags & P3M_FLAG_PENCIL) != 0;
  G.nxb = G.pencil ? G.nd : G.nd * G.nd;
  const int count = which == 0 ? G.nxb : (which == 1 ? (G.pencil ? G.nd : 0) : (which == 2 ? yz_count(&G) : -1));
  if (count < 0 || rank < 0 || rank >= G.nodes) return P3M_EINVAL;
  if (count == 0) return 0;
  if (j < 0 || j >= count) return P3M_EINVAL;
  if (which == 0) { *peer = xl_peer(&G, rank, j); *index = xl_index(&G, rank); }
  else if (which == 1) { *peer = xy_peer(&G, rank, j); *index = xy_index(&G, rank); }
  else { *peer = yz_peer(&G, rank, j); *index = yz_index(&G, rank); }
  return count;
}

// ------------------------------------------------------------------ coarse_power.f90 on the distributed rho-hat
// every local rank bins its own ky slab (d.lz, left by the last step's forward transform); the weights and sums of all
// processes are added by one all-reduce (the reference's mpi_reduce, :109), then every process holds the spectrum.
int coarse_power_accumulate(p3m_ctx *c, const float *lz, int planes, int ky0, int kx0, int nc, int nchunk, float rho_c_mean, double *d_ps);
void coarse_power_finish(const double *acc, int nc, float box, float *ps);
extern "C" int p3m_hip_group_coarse_power(p3m_group *G, float mass_p, float box, float *ps) {
  if (!G || !ps) return P3M_EINVAL;
  if (G->nodes == 1) return p3m_hip_coarse_power(G->ctx[0], mass_p, box, ps);
  HIP_TRY(hipSetDevice(G->device));
  const Geometry &g = G->ctx[0]->g;
  const int nc = g.nc, nb = nc + 2;
  double *d_ps = nullptr;
  HIP_TRY(hipMalloc(&d_ps, sizeof(double) * 2 * nb));
  std::vector<double> acc(2 * nb);
  auto body = [&]() -> int {
    HIP_TRY(hipMemsetAsync(d_ps, 0, sizeof(double) * 2 * nb, G->stream));
    const float nfp = (float)(g.Nn * g.nodes_dim / 2), fnc = (float)nc;
    const float rho_c_mean = nfp * nfp * nfp * mass_p / (fnc * fnc * fnc);   // coarse_power.f90:24
    for (size_t i = 0; i < G->ctx.size(); i++) {
      p3m_ctx *c = G->ctx[i];
      hipStream_t keep = c->stream; c->stream = G->stream;
      const int r = coarse_power_accumulate(c, G->cd[i].lz, G->s, yz_index(G, G->lrank[i]) * G->s, G->pencil ? xy_index(G, G->lrank[i]) * G->ncl * 16 : 0, nc, G->ncl,
                                            rho_c_mean, d_ps);
      c->stream = keep;
      P3M_TRY(r);
    }
    // the same choice as reduce_step_out and do_exchange: RCCL when there is a communicator, the host transport otherwise
    const bool use_tr = G->nprocs > 1 && !G->comm && G->have_tr;
    if (G->nprocs > 1 && G->comm) NCCL_TRY(ncclAllReduce(d_ps, d_ps, 2 * nb, ncclDouble, ncclSum, G->comm, G->stream));
    HIP_TRY(hipMemcpyAsync(acc.data(), d_ps, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, G->stream));
    HIP_TRY(hipStreamSynchronize(G->stream));
    if (use_tr) { if (G->tr.allreduce_sum_f64(G->tr.user, acc.data(), 2 * nb)) { p3m_set_error("host transport: all-reduce callback failed"); return P3M_ECOMM; } }
    else if (G->nprocs > 1 && !G->comm) { p3m_set_error("group reduction between processes needs RCCL or a host transport"); return P3M_ECOMM; }
    return P3M_OK;
  };
  const int r = body();
  (void)hipFree(d_ps);
  if (r != P3M_OK) return r;
  coarse_power_finish(acc.data(), nc, box, ps);
  return P3M_OK;
}
