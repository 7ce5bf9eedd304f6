// p3m_api.hip -- the C ABI of include/p3m_hip.h: context lifecycle, particle upload/download,
// the `particle_mesh` sequence (particle_mesh_threaded.f90:2-726) and the probes/timers.
#include "p3m_internal.h"
#include "kick_fused.h"
#include <stdlib.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>


static thread_local char g_err[1024] = "";
void p3m_set_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char *p3m_hip_last_error(void) { return g_err; }

template <typename T> static int dalloc(T **p, size_t n) {
  *p = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) { p3m_set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e)); return P3M_ENOMEM; }
  return P3M_OK;
}
template <typename T> static void dfree(T *&p) { if (p) (void)hipFree(p); p = nullptr; }

static int fill_geometry(const p3m_params *p, Geometry *g) {
  if (p->nodes_dim < 1 || p->tiles_node_dim < 1 || p->mesh_scale < 2 || p->nf_buf < 0) { p3m_set_error("bad parameters"); return P3M_EINVAL; }
  g->nodes_dim = p->nodes_dim; g->nodes = p->nodes_dim * p->nodes_dim * p->nodes_dim;
  g->T = p->tiles_node_dim; g->ntiles = g->T * g->T * g->T;
  g->nf = p->nf_tile; g->nb = p->nf_buf; g->ncut = p->nf_cutoff; g->ms = p->mesh_scale; g->pp_range = p->pp_range;
  g->pt = g->nf - 2 * g->nb;
  if (g->pt <= 0 || g->pt % g->ms || g->nb % g->ms || (g->nf & 1)) {
    p3m_set_error("nf_tile=%d, nf_buf=%d: nf_tile-2*nf_buf must be a positive multiple of mesh_scale (parameters.example:25-31)", g->nf, g->nb);
    return P3M_EINVAL;
  }
  if (g->ncut > g->nf / 2) { p3m_set_error("nf_cutoff too large for nf_tile"); return P3M_EINVAL; }
  if (g->pp_range + 1 > g->ncut || g->pp_range > g->nb) { p3m_set_error("pp_range too large"); return P3M_EINVAL; }
  g->Nn = g->pt * g->T; g->E = g->Nn + 2 * g->nb;
  g->nc_buf = g->nb / g->ms; g->nct = g->pt / g->ms; g->ncn = g->nct * g->T; g->nc = g->ncn * g->nodes_dim;
  if (p->flags & P3M_FLAG_PENCIL) {
    if (g->ncn % g->nodes_dim) { p3m_set_error("cannot evenly decompose mesh into pencils: mod(nc_dim, nodes_dim**2) != 0 (mpi_initialization_p3dfft.f90:27)"); return P3M_EINVAL; }
  } else if (g->nc % g->nodes) { p3m_set_error("cannot evenly decompose mesh into slabs (mpi_initialization.f90:26)"); return P3M_EINVAL; }
  g->nc_slab = g->nc / g->nodes;
  g->hx = g->nf / 2 + 1; g->fb = g->pt + 3;
  g->fbp = (g->fb + 3) / 4 * 4;
  g->px = (g->hx + 15) / 16 * 16; g->pxc = ((g->nc / 2 + 1) + 15) / 16 * 16;
  const bool coarse_only = (p->flags & P3M_FLAG_COARSE_ONLY) != 0;   // no particle store, no fine mesh: their int32 limits do not apply
  if (!coarse_only && (int64_t)g->E * g->E * g->E > 2000000000LL) { p3m_set_error("extended fine domain %d^3 exceeds int32 cell indices", g->E); return P3M_EINVAL; }
  const int nd = g->nodes_dim, rk = p->rank;
  if (rk < 0 || rk >= g->nodes) { p3m_set_error("rank %d out of range", rk); return P3M_EINVAL; }
  g->cart[0] = rk / (nd * nd); g->cart[1] = (rk / nd) % nd; g->cart[2] = rk % nd;  // mpi_initialization.f90:60-64
  for (int d = 0; d < 3; d++) {
    int cm[3] = {g->cart[0], g->cart[1], g->cart[2]}, cp[3] = {g->cart[0], g->cart[1], g->cart[2]};
    cm[d] = (cm[d] - 1 + nd) % nd; cp[d] = (cp[d] + 1) % nd;
    g->nbr[2 * d] = cm[0] * nd * nd + cm[1] * nd + cm[2]; g->nbr[2 * d + 1] = cp[0] * nd * nd + cp[1] * nd + cp[2];
  }
  // cubepm.par:170-172
  const double Nn = g->Nn, nb = g->nb;
  const double inner = (double)((g->Nn / 2) * (int64_t)(g->Nn / 2)) * (g->Nn / 2) + (8.0 * nb * nb * nb + 6.0 * nb * Nn * Nn + 12.0 * nb * nb * Nn) / 8.0;
  g->max_np = coarse_only ? 0 : (int64_t)((double)p->density_buffer * inner);
  if (g->max_np > 2000000000LL) { p3m_set_error("max_np exceeds int32"); return P3M_EINVAL; }
  return P3M_OK;
}

extern "C" int32_t p3m_hip_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

int p3m_ctx_share_hint = 1;   // contexts still to be created on this device, set by p3m_hip_group_create around its p3m_hip_create calls
extern "C" int p3m_hip_create(const p3m_params *params, p3m_ctx **out) {
  if (!params || !out) return P3M_EINVAL;
  *out = nullptr;
  p3m_ctx *c = new p3m_ctx();
  c->p = *params;
  int r = fill_geometry(params, &c->g);
  if (r) { delete c; return r; }
  const Geometry &g = c->g;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { p3m_set_error("no HIP device (this library has no CPU fallback)"); delete c; return P3M_EDEVICE; }
  if (params->device >= 0) { c->device = params->device; } else { (void)hipGetDevice(&c->device); }
  auto fail = [&](int code) { p3m_hip_destroy(c); return code; };
  if (hipSetDevice(c->device) != hipSuccess) { p3m_set_error("hipSetDevice(%d) failed", c->device); return fail(P3M_EDEVICE); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { p3m_set_error("stream creation failed"); return fail(P3M_EDEVICE); }
  if (g.nodes == 1 && !(getenv("P3M_ONE_STREAM") && getenv("P3M_ONE_STREAM")[0] == '1')) {
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_cf, hipEventDisableTiming) != hipSuccess) { p3m_set_error("stream creation failed"); return fail(P3M_EDEVICE); }
  }
  c->cap = g.max_np;
#define A(x) do { int _r = (x); if (_r) return fail(_r); } while (0)
  const bool coarse_only = (params->flags & P3M_FLAG_COARSE_ONLY) != 0;
  if (coarse_only && g.nodes == 1) { p3m_set_error("P3M_FLAG_COARSE_ONLY is for multi-rank groups"); return fail(P3M_EINVAL); }
  if (!coarse_only) {
  A(dalloc(&c->pos, c->cap)); A(dalloc(&c->vel, c->cap)); A(dalloc(&c->vel_alt, c->cap)); A(dalloc(&c->pid_home, c->cap));
  A(dalloc(&c->spos, c->cap)); A(dalloc(&c->tpos, c->cap));
  A(dalloc(&c->flags, c->cap + 8));
  c->cand_seg = (int)std::max<int64_t>(1024, c->cap / 16);   // a list holds 4x its share of ALL records; beyond that the fix-up scans everything
  if (getenv("P3M_CAND_SEG")) c->cand_seg = std::max(1, atoi(getenv("P3M_CAND_SEG")));   // tests of the overflow path
  A(dalloc(&c->cand, (size_t)c->cand_seg * P3M_CAND_SLOTS)); A(dalloc(&c->cand_cnt, 16 * P3M_CAND_SLOTS + 16));
  A(dalloc(&c->gl_cnt, 16 * P3M_GL_SLOTS)); c->gl_cap = (int)std::min<int64_t>(0x7fffffff, (4 * c->cap) / P3M_GL_SLOTS);   // the lists live in a float4-per-record buffer
  const int64_t ncell = (int64_t)g.E * g.E * g.E;
  int *raw = nullptr; A(dalloc(&raw, ncell + 16)); c->cell_end = raw + 3;  // every entry of [0, ncell] is rewritten by each sort; the pads stay zero
  if (hipMemset(raw, 0, (size_t)(ncell + 16) * sizeof(int)) != hipSuccess) return fail(P3M_EDEVICE);
  raw = nullptr; A(dalloc(&raw, (size_t)g.E * g.E + 16)); c->row_end = raw + 3;   // (row_end+1) is 16-byte aligned for the scan
  if (particles_preload() != P3M_OK) return fail(P3M_EDEVICE);
  if (scan_reserve(c, std::max<int64_t>(c->cap, (int64_t)g.E * g.E) + 8) != P3M_OK) return fail(P3M_ENOMEM);
  c->crow_w = (g.ncn + 2 + 2 * g.T + 3) & ~3; A(dalloc(&c->crow, (size_t)g.E * g.E * c->crow_w));
  A(dalloc(&c->d_counters, 16));
  { const int64_t ec = g.E / g.ms; A(dalloc(&c->cflag, (size_t)(ec * ec * ec + 16))); }
  if (hipHostMalloc(reinterpret_cast<void **>(&c->h_counters), 16 * sizeof(int)) != hipSuccess) return fail(P3M_ENOMEM);
  // fine mesh: as many tiles per sweep as fit the budget for rho+work: 64 GiB, or this context's share (a group places
  // several logical ranks on one device: p3m_ctx_share_hint) of 60 % of what is free on the device now, whichever is less
  const size_t S = (size_t)(2 * g.px) * g.nf * g.nf;
  size_t budget = (size_t)64 << 30;
  { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) budget = std::min(budget, (size_t)(0.6 * (double)fr) / (size_t)std::max(1, p3m_ctx_share_hint)); }
  c->tile_batch = (int)std::max<size_t>(1, std::min<size_t>(g.ntiles, budget / (4 * S * sizeof(float))));
  A(dalloc(&c->rho, S * c->tile_batch)); A(dalloc(&c->work, 3 * S * c->tile_batch));
  if (hipMemset(c->rho, 0, S * c->tile_batch * sizeof(float)) != hipSuccess || hipMemset(c->work, 0, 3 * S * c->tile_batch * sizeof(float)) != hipSuccess) return fail(P3M_EDEVICE);
  A(dalloc(&c->fbox, (size_t)3 * g.ntiles * g.fb * g.fb * g.fbp));
  // the fused inverse-x + kick pass of NGP steps (kick_fused.hip): rows per batch (0: this tile size has none) and the per-row flags, all zero between steps
  c->fuse_nr = kick_fused_rows(g.nf, g.fbp, g.nb - 2);
  if (c->fuse_nr > 0) { A(dalloc(&c->rowflag, (size_t)g.ntiles * g.fb * g.fb + 16)); if (hipMemset(c->rowflag, 0, (size_t)g.ntiles * g.fb * g.fb + 16) != hipSuccess) return fail(P3M_EDEVICE); }
  A(dalloc(&c->kern_f, (size_t)3 * g.nf * g.nf * g.px));
  if (hipMemset(c->kern_f, 0, (size_t)3 * g.nf * g.nf * g.px * sizeof(float)) != hipSuccess) return fail(P3M_EDEVICE);
  A(fft_plan_create(&c->plan_f, g.nf));
  A(dalloc(&c->cmom, (size_t)8 * (g.ncn + 1) * (g.ncn + 1) * (g.ncn + 1)));
  }   // !coarse_only
  // coarse mesh
  A(dalloc(&c->rho_c, (size_t)g.ncn * g.ncn * g.ncn));
  A(dalloc(&c->force_c, (size_t)3 * (g.ncn + 2) * (g.ncn + 2) * (g.ncn + 2)));
  if (g.nodes == 1) {
    const size_t Sc = (size_t)g.nc * g.nc * (2 * g.pxc);
    A(dalloc(&c->slab, Sc)); A(dalloc(&c->slab_w, Sc)); A(dalloc(&c->slab_o, Sc));
    A(dalloc(&c->kern_c, (size_t)3 * g.nc * g.nc * g.pxc));
    if (hipMemset(c->slab, 0, Sc * sizeof(float)) != hipSuccess || hipMemset(c->slab_w, 0, Sc * sizeof(float)) != hipSuccess ||
        hipMemset(c->kern_c, 0, (size_t)3 * g.nc * g.nc * g.pxc * sizeof(float)) != hipSuccess) return fail(P3M_EDEVICE);
    A(fft_plan_create(&c->plan_c, g.nc));
  }
  // the step's reduced scalars in ONE block (sums | maxima | per-tile PP maxima) with one pinned mirror: one download per rank and step
  {
    const size_t nb_sums = 4 * P3M_SUM_SPAN * sizeof(double), nb_red = 8 * P3M_RED_SPAN * sizeof(float), nb_ext = ((size_t)g.ntiles * sizeof(float) + 15) & ~(size_t)15;
    c->red_bytes = nb_sums + nb_red + nb_ext;
    char *blk = nullptr; A(dalloc(&blk, c->red_bytes)); c->d_redblk = blk;
    c->d_sums = reinterpret_cast<double *>(blk); c->d_red = reinterpret_cast<float *>(blk + nb_sums); c->d_tile_ext = reinterpret_cast<float *>(blk + nb_sums + nb_red);
    char *hb = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&hb), c->red_bytes) != hipSuccess) return fail(P3M_ENOMEM);
    c->h_redblk = hb;
    c->h_sums_raw = reinterpret_cast<double *>(hb); c->h_red_raw = reinterpret_cast<float *>(hb + nb_sums); c->h_tile_ext = reinterpret_cast<float *>(hb + nb_sums + nb_red);
  }
#undef A
  // variable_initialization.f90:22-29
  c->last.dt_f_acc = c->last.dt_pp_acc = c->last.dt_pp_ext_acc = c->last.dt_c_acc = 1000.f;
  *out = c;
  return P3M_OK;
}

extern "C" void p3m_hip_destroy(p3m_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  dfree(c->pos); dfree(c->vel); dfree(c->vel_alt); dfree(c->pid_home); dfree(c->spos);
  dfree(c->tpos); dfree(c->flags); dfree(c->cflag); dfree(c->cand); dfree(c->cand_cnt); dfree(c->gl_cnt); dfree(c->scan_tmp); dfree(c->scan_state); dfree(c->d_counters); dfree(c->pp_plan); dfree(c->pp_task_group); dfree(c->pp_counter); dfree(c->pp_htask); dfree(c->pp_slow); dfree(c->pp_intra_done);
  if (c->cell_end) { int *raw = c->cell_end - 3; (void)hipFree(raw); c->cell_end = nullptr; }
  if (c->row_end) { int *raw = c->row_end - 3; (void)hipFree(raw); c->row_end = nullptr; }
  dfree(c->crow);
  dfree(c->rho); dfree(c->work); dfree(c->fbox); dfree(c->rowflag); dfree(c->kern_f);
  dfree(c->rho_c); dfree(c->cmom); dfree(c->force_c); dfree(c->slab); dfree(c->slab_w); dfree(c->slab_o); dfree(c->kern_c);
  dfree(c->d_redblk); c->d_red = nullptr; c->d_tile_ext = nullptr; c->d_sums = nullptr;
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  if (c->h_redblk) (void)hipHostFree(c->h_redblk);
  fft_plan_destroy(&c->plan_f); fft_plan_destroy(&c->plan_c);
  if (c->own_pt) { delete c->pt; } c->pt = nullptr;
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->ev_dep) (void)hipEventDestroy(c->ev_dep);
  if (c->ev_cf) (void)hipEventDestroy(c->ev_cf);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int64_t p3m_hip_derived(const p3m_ctx *c, int32_t what) {
  if (!c) return -1;
  switch (what) {
    case 0: return c->g.max_np; case 1: return c->g.nc; case 2: return c->g.ncn; case 3: return c->g.Nn; case 4: return c->g.nc_slab; case 5: return c->g.pt;
    case 6: return c->tile_batch; case 7: return c->g.E;
  }
  return -1;
}
extern "C" void *p3m_hip_stream(p3m_ctx *c) { return c ? (void *)c->stream : nullptr; }

// ------------------------------------------------------------------ kernels
extern "C" int p3m_hip_set_kernel_tables(p3m_ctx *c, const float *fine_table, const float *coarse_table) {
  if (!c || !fine_table || !coarse_table) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(build_fine_kernel(c, fine_table));
  P3M_TRY(build_coarse_kernel(c, coarse_table));
  return P3M_OK;
}

// reference layout (3, hx, n, n) component fastest  <->  device SoA planes [3][n][n][hx]
static void aos_to_planes(const float *aos, float *planes, size_t ncx) {
  for (int comp = 0; comp < 3; comp++) for (size_t i = 0; i < ncx; i++) planes[comp * ncx + i] = aos[i * 3 + comp];
}
static void planes_to_aos(const float *planes, float *aos, size_t ncx) {
  for (int comp = 0; comp < 3; comp++) for (size_t i = 0; i < ncx; i++) aos[i * 3 + comp] = planes[comp * ncx + i];
}
// kernels: reference rows hold hx values, device rows px (zero padded)
// kernels: reference (3, hx, n, n) component fastest  <->  device SoA planes in the FFT's bundle layout
// LZ [y][chunk][z][16] (fft.hip "memory layouts"), zero in the pad columns
static inline size_t lz_index(int n, int px, int x, int y, int z) {
  return ((((size_t)y * (px / 16) + x / 16) * n + z) * 16) + x % 16;
}
static void kern_to_device(const float *aos, float *planes, int n, int hx, int px) {
  const size_t plane = (size_t)n * n * px;
  for (size_t i = 0; i < 3 * plane; i++) planes[i] = 0.f;
  for (int comp = 0; comp < 3; comp++) for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < hx; x++)
    planes[comp * plane + lz_index(n, px, x, y, z)] = aos[((((size_t)z * n + y) * hx) + x) * 3 + comp];
}
static void kern_from_device(const float *planes, float *aos, int n, int hx, int px) {
  const size_t plane = (size_t)n * n * px;
  for (int comp = 0; comp < 3; comp++) for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < hx; x++)
    aos[((((size_t)z * n + y) * hx) + x) * 3 + comp] = planes[comp * plane + lz_index(n, px, x, y, z)];
}

// kern_f in the reference's layout -> this context's device planes (also used by p3m_hip_group_set_kernels_raw)
int kernels_set_fine_raw(p3m_ctx *c, const float *kern_f) {
  const Geometry &g = c->g;
  const size_t nf = (size_t)g.nf * g.nf * g.px;
  std::vector<float> tmp(3 * nf);
  kern_to_device(kern_f, tmp.data(), g.nf, g.hx, g.px);
  HIP_TRY(hipMemcpy(c->kern_f, tmp.data(), sizeof(float) * 3 * nf, hipMemcpyHostToDevice));
  c->have_kf = true; c->kf_zmirror = false;   // (a table from outside is read as it is)
  return P3M_OK;
}
extern "C" int p3m_hip_set_kernels_raw(p3m_ctx *c, const float *kern_f, const float *kern_c) {
  if (!c || !kern_f || !kern_c) return P3M_EINVAL;
  if (c->g.nodes != 1) { p3m_set_error("set_kernels_raw: single-rank contexts only (groups: p3m_hip_group_set_kernels_raw)"); return P3M_EINVAL; }
  HIP_TRY(hipSetDevice(c->device));
  const Geometry &g = c->g;
  const size_t nf = (size_t)g.nf * g.nf * g.px, ncx = (size_t)g.nc * g.nc * g.pxc;
  std::vector<float> tmp(3 * std::max(nf, ncx));
  kern_to_device(kern_f, tmp.data(), g.nf, g.hx, g.px);
  HIP_TRY(hipMemcpy(c->kern_f, tmp.data(), sizeof(float) * 3 * nf, hipMemcpyHostToDevice));
  kern_to_device(kern_c, tmp.data(), g.nc, g.nc / 2 + 1, g.pxc);
  HIP_TRY(hipMemcpy(c->kern_c, tmp.data(), sizeof(float) * 3 * ncx, hipMemcpyHostToDevice));
  c->have_kf = c->have_kc = true; c->kf_zmirror = false;
  return P3M_OK;
}

extern "C" int p3m_hip_get_kernels(p3m_ctx *c, float *kern_f, float *kern_c) {
  if (!c) return P3M_EINVAL;
  if (kern_f) P3M_TRY(need_particles(c, "p3m_hip_get_kernels (fine kernel)"));
  if ((kern_f && !c->have_kf) || !c->have_kc) { p3m_set_error("kernels not set"); return P3M_ESTATE; }
  HIP_TRY(hipSetDevice(c->device));
  const Geometry &g = c->g;
  const size_t nf = (size_t)g.nf * g.nf * g.px, ncx = (size_t)g.nc * g.nc * g.pxc;
  std::vector<float> tmp(3 * std::max(nf, ncx));
  if (kern_f) { HIP_TRY(hipMemcpy(tmp.data(), c->kern_f, sizeof(float) * 3 * nf, hipMemcpyDeviceToHost)); kern_from_device(tmp.data(), kern_f, g.nf, g.hx, g.px); }
  if (kern_c && c->kern_c) { HIP_TRY(hipMemcpy(tmp.data(), c->kern_c, sizeof(float) * 3 * ncx, hipMemcpyDeviceToHost)); kern_from_device(tmp.data(), kern_c, g.nc, g.nc / 2 + 1, g.pxc); }
  return P3M_OK;
}

// ------------------------------------------------------------------ particle store
__global__ __launch_bounds__(256) void k_unpack_xv(const float *__restrict__ xv6, float4 *__restrict__ pos, float4 *__restrict__ vel, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *r = xv6 + (int64_t)i * 6;
  pos[i] = make_float4(r[0], r[1], r[2], 0.f); vel[i] = with_index(r[3], r[4], r[5], i);   // PID slot = upload index
}
__global__ __launch_bounds__(256) void k_pack_xv(const float4 *__restrict__ pos, const float4 *__restrict__ vel, float *__restrict__ xv6, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = pos[i]; const float4 v = vel[i];
  float *r = xv6 + (int64_t)i * 6;
  r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = v.x; r[4] = v.y; r[5] = v.z;
}
__global__ __launch_bounds__(256) void k_gather_pid(const float4 *__restrict__ vel, const int64_t *__restrict__ pid_home, int64_t *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = pid_home[rec_index(vel[i])];
}
__global__ __launch_bounds__(256) void k_iota_pid(int64_t *pid, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) pid[i] = (int64_t)i + 1;
}

extern "C" int p3m_hip_upload_particles(p3m_ctx *c, const float *xv6, const int64_t *pid, int32_t np_local) {
  if (!c || np_local < 0 || (np_local > 0 && !xv6)) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_upload_particles"));
  if (np_local > c->cap) { p3m_set_error("np_local %d exceeds max_np %lld", np_local, (long long)c->cap); return P3M_ECAPACITY; }
  HIP_TRY(hipSetDevice(c->device));
  c->np_local = np_local; c->np_all = 0; c->pending_compact = false; c->hist_done = false; c->gl_valid = false; c->cnt_from_kick = 0; c->n_home = 0; c->cell_max_known = false;
  if (np_local == 0) return P3M_OK;
  float *tmp = nullptr; P3M_TRY(dalloc(&tmp, (size_t)np_local * 6));
  HIP_TRY(hipMemcpyAsync(tmp, xv6, sizeof(float) * 6 * (size_t)np_local, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_unpack_xv, dim3(cdiv(np_local, 256)), dim3(256), 0, c->stream, (const float *)tmp, c->pos, c->vel, np_local);
  if (pid) HIP_TRY(hipMemcpyAsync(c->pid_home, pid, sizeof(int64_t) * (size_t)np_local, hipMemcpyHostToDevice, c->stream));
  else hipLaunchKernelGGL(k_iota_pid, dim3(cdiv(np_local, 256)), dim3(256), 0, c->stream, c->pid_home, np_local);
  c->n_home = np_local;
  HIP_TRY(hipStreamSynchronize(c->stream));
  dfree(tmp);
  return P3M_OK;
}

extern "C" int p3m_hip_download_particles(p3m_ctx *c, float *xv6, int64_t *pid, int32_t *np_local) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_download_particles"));
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(particles_resolve(c));
  if (np_local) *np_local = c->np_local;
  const int n = c->np_local;
  if (n == 0) return P3M_OK;
  if (xv6) {
    float *tmp = nullptr; P3M_TRY(dalloc(&tmp, (size_t)n * 6));
    hipLaunchKernelGGL(k_pack_xv, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const float4 *)c->pos, (const float4 *)c->vel, tmp, n);
    HIP_TRY(hipMemcpyAsync(xv6, tmp, sizeof(float) * 6 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dfree(tmp);
  }
  if (pid) {   // PIDs live in pid_home; the records carry their slot (p3m_internal.h)
    int64_t *tmp = nullptr; P3M_TRY(dalloc(&tmp, (size_t)n));
    hipLaunchKernelGGL(k_gather_pid, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const float4 *)c->vel, (const int64_t *)c->pid_home, tmp, n);
    HIP_TRY(hipMemcpyAsync(pid, tmp, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream));
    dfree(tmp);
  }
  return P3M_OK;
}

// ------------------------------------------------------------------ phases
static int need_kernels(p3m_ctx *c) {
  if (!c->have_kf || !c->have_kc) { p3m_set_error("particle_mesh before the Green's functions were set (fine_kernel/coarse_kernel, cubepm.f90:42-46)"); return P3M_ESTATE; }
  return P3M_OK;
}

extern "C" int p3m_hip_update_position(p3m_ctx *c, float dt, float dt_old, const float *offset) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_update_position"));
  HIP_TRY(hipSetDevice(c->device));
  return particles_drift(c, dt, dt_old, offset);
}

extern "C" int p3m_hip_link_list_and_pass(p3m_ctx *c) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_link_list_and_pass"));
  HIP_TRY(hipSetDevice(c->device));
  const int r = particles_pass_and_sort(c);
  if (r != P3M_OK) particles_reset_after_error(c);   // e.g. P3M_ECAPACITY after the images were counted into the row histogram
  return r;
}

static int fine_sweep(p3m_ctx *c, float mass_p, bool kick_follows) {
  const Geometry &g = c->g;
  // NGP, and the kick phase follows: the last pass of the force (inverse x) runs with the kick (kick_fused.hip); fine_max_and_kick picks
  // it up.  (The timing hook p3m_hip_time_fine_sweep runs the stand-alone sweep: five launches and a force box.)
  c->xinv_deferred = kick_follows && fine_kick_fusable(c);
  for (int t0 = 0; t0 < g.ntiles; t0 += c->tile_batch) {
    const int nt = std::min(c->tile_batch, g.ntiles - t0);
    { PhaseScope ps(c->pt, P3M_PH_FINE_DEPOSIT, c->stream); P3M_TRY(fine_deposit(c, t0, nt, mass_p, c->xinv_deferred)); }
    { PhaseScope ps(c->pt, P3M_PH_FINE_FFT, c->stream); P3M_TRY(fine_force(c, t0, nt, c->xinv_deferred)); }
  }
  return P3M_OK;
}

int reductions_clear(p3m_ctx *c) {
  if (c->step_zeroed) return P3M_OK;   // whole steps: step_prezero
  HIP_TRY(hipMemsetAsync(c->d_red, 0, 8 * P3M_RED_SPAN * sizeof(float), c->stream));
  HIP_TRY(hipMemsetAsync(c->d_sums, 0, 4 * P3M_SUM_SPAN * sizeof(double), c->stream));
  return P3M_OK;
}
// Whole steps: everything the step accumulates into, cleared in ONE launch right after the drift (the drift's compaction still reads the
// previous step's block offsets in c->flags; nothing queued later in the step needs an old value of any of these): the reduction slots,
// the candidate lists' lengths, the per-tile PP maxima, the coarse density and the kick's survivor counts.  The phase-level entry
// points clear what they need themselves (step_zeroed == false).
int step_prezero(p3m_ctx *c) {
  const Geometry &g = c->g;
  P3M_TRY(zero_add(c, c->d_redblk, c->red_bytes));   // sums | maxima | per-tile PP maxima
  P3M_TRY(zero_add(c, c->cand_cnt, sizeof(int) * (16 * P3M_CAND_SLOTS + 16)));
  P3M_TRY(zero_add(c, c->rho_c, sizeof(float) * (size_t)g.ncn * g.ncn * g.ncn));
  if ((c->p.flags & P3M_FLAG_NGP) && !(c->p.flags & P3M_FLAG_MOVE_GRID_BACK)) P3M_TRY(zero_add(c, c->flags, sizeof(int) * (size_t)(cdiv(c->cap, 256) + 1)));   // fine_max_and_kick's cnt256
  P3M_TRY(zero_flush(c));
  c->step_zeroed = true;
  return P3M_OK;
}
int reductions_download(p3m_ctx *c) {
  HIP_TRY(hipMemcpyAsync(c->h_redblk, c->d_redblk, c->red_bytes, hipMemcpyDeviceToHost, c->stream));   // sums, maxima and the per-tile PP maxima: one block
  return P3M_OK;
}
// a step whose NGP density went through bytes (RowDep::rho8) met a cell of more than 255 records: the density is wrong, the step fails
int rho_u8_check(p3m_ctx *c) {
  const bool bad = c->rho_u8_step && c->h_red[3] >= 255.5f;
  c->rho_u8_step = false;
  if (bad) { c->cell_max_known = false; p3m_set_error("a fine cell holds more than 255 particles, more than twice what it held a step ago: the byte-per-cell NGP density of this step is saturated (P3M_RHO_F32=1 keeps floats)"); return P3M_ECAPACITY; }
  return P3M_OK;
}
void reductions_fold(p3m_ctx *c) {
  for (int k = 0; k < 8; k++) { float m = 0.f; for (int sl = 0; sl < P3M_NSLOT; sl++) m = std::max(m, c->h_red_raw[k * P3M_RED_SPAN + sl * 16]); c->h_red[k] = m; }
  // slot 3: the largest count of a fine cell the sort of this step saw (0: below 64) -- what decides whether the NEXT step's NGP density is
  // written as one byte per cell (particles_sort_enqueue)
  if (c->cell_max_reported) { c->cell_max = c->h_red[3]; c->cell_max_known = true; c->cell_max_reported = false; }
  for (int k = 0; k < 4; k++) { double t = 0.0; for (int sl = 0; sl < P3M_NSLOT; sl++) t += c->h_sums_raw[k * P3M_SUM_SPAN + sl * 8]; c->h_sums[k] = t; }
}

// whole-step calls only: the fine kick carries the coarse kick (NGP: k_fine_kick_rows<true>; CIC fine mesh: k_fine_kick_cic<true>).
// With PPINT / PP_EXT the reference adds the PP kicks between the two mesh kicks (fine, PP, coarse); here the sum is formed as
// fine, coarse, PP -- the same terms, each formed as in the reference, added in another order: 1e-7 relative on a velocity against
// the 1e-5 bar, for one pass over the records less (295 us per rank at the bench's size).  P3M_SEPARATE_COARSE_KICK=1 keeps the
// reference's order (a pass of its own after the PP kicks)
bool coarse_kick_rides_on_fine(const p3m_ctx *c) {
  static const bool off = getenv("P3M_SEPARATE_COARSE_KICK") && getenv("P3M_SEPARATE_COARSE_KICK")[0] == '1';
  return !off && !(c->p.flags & P3M_FLAG_COARSE_NGP);   // -DCOARSE_NGP: k_coarse_kick has the switch
}

// the two halves of the fine mesh step: density + force of every tile (positions only), then everything that moves velocities
int fine_mesh_force_phase(p3m_ctx *c, float mass_p, bool may_clear) {
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(need_kernels(c));
  if (may_clear && !c->rho_from_sort) P3M_TRY(reductions_clear(c));   // else cleared before the sort, which already added the NGP mass sum
  if (!c->step_zeroed) HIP_TRY(hipMemsetAsync(c->d_tile_ext, 0, c->g.ntiles * sizeof(float), c->stream));
  return fine_sweep(c, mass_p, true);
}
int fine_mesh_kick_phase(p3m_ctx *c, float a_mid, float dt, float mass_p) {
  // (NGP whole steps: the inverse x pass of the force runs inside the kick, kick_fused.hip -- it is timed with the kick)
  { PhaseScope ps(c->pt, P3M_PH_FINE_KICK, c->stream); P3M_TRY(fine_max_and_kick(c, a_mid, dt)); }
  // -DPPINT + -DPP_EXT: the extended pass first -- where its lean light pass runs it sums the bucket pairs of the records it stages anyway
  // (:324-361 inside :378-624's pass) and flags them; pp_intra then works the rest (flagged coarse cells, heavy records, crowded patches)
  const bool ppint = (c->p.flags & P3M_FLAG_PPINT) && (c->p.flags & P3M_FLAG_NGP), ppext = (c->p.flags & P3M_FLAG_PP_EXT) != 0;
  if (ppext) { PhaseScope ps(c->pt, P3M_PH_PP_EXT, c->stream); P3M_TRY(pp_extended(c, a_mid, dt, mass_p, ppint)); }
  if (ppint) { PhaseScope ps(c->pt, P3M_PH_PP_INTRA, c->stream); P3M_TRY(pp_intra(c, a_mid, dt, mass_p)); }
  return P3M_OK;
}
extern "C" int p3m_hip_fine_mesh(p3m_ctx *c, float a_mid, float dt, float mass_p) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_fine_mesh"));
  P3M_TRY(fine_mesh_force_phase(c, mass_p, true));
  return fine_mesh_kick_phase(c, a_mid, dt, mass_p);
}

extern "C" int p3m_hip_coarse_mesh(p3m_ctx *c, float a_mid, float dt, float mass_p) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_coarse_mesh"));
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(need_kernels(c));
  P3M_TRY(coarse_deposit(c, mass_p));
  P3M_TRY(coarse_force(c));
  P3M_TRY(coarse_kick(c, a_mid, dt));   // coarse_vel_update = .true. (cubepm.par:87)
  return P3M_OK;
}

extern "C" int p3m_hip_delete_particles(p3m_ctx *c, const float *move_back) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_delete_particles"));
  HIP_TRY(hipSetDevice(c->device));
  return particles_finalize(c, (c->p.flags & P3M_FLAG_MOVE_GRID_BACK) ? move_back : nullptr);
}

// shared by the group variant: maps on the device, zeroed by the caller; returns the rank's projected mass
int projection_rank(p3m_ctx *c, float mass_p, float *d_pxy, float *d_pxz, float *d_pyz, double *rho_node) {
  HIP_TRY(hipMemsetAsync(c->d_sums + 3 * P3M_SUM_SPAN, 0, P3M_SUM_SPAN * sizeof(double), c->stream));
  P3M_TRY(fine_projection(c, mass_p, d_pxy, d_pxz, d_pyz));
  HIP_TRY(hipMemcpyAsync(c->h_sums_raw + 3 * P3M_SUM_SPAN, c->d_sums + 3 * P3M_SUM_SPAN, P3M_SUM_SPAN * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  double t = 0.0;
  for (int sl = 0; sl < P3M_NSLOT; sl++) t += c->h_sums_raw[3 * P3M_SUM_SPAN + sl * 8];
  if (rho_node) *rho_node = t;
  return P3M_OK;
}
extern "C" int p3m_hip_projection(p3m_ctx *c, float mass_p, float *pxy, float *pxz, float *pyz, double *rho_node) {
  if (!c || !pxy || !pxz || !pyz) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_projection"));
  HIP_TRY(hipSetDevice(c->device));
  const size_t n2 = (size_t)c->g.Nn * c->g.nodes_dim * c->g.Nn * c->g.nodes_dim;
  float *d = nullptr;
  HIP_TRY(hipMalloc(&d, 3 * n2 * sizeof(float)));
  int r = P3M_OK;
  if (hipMemsetAsync(d, 0, 3 * n2 * sizeof(float), c->stream) != hipSuccess) r = P3M_EDEVICE;
  if (!r) r = projection_rank(c, mass_p, d, d + n2, d + 2 * n2, rho_node);
  if (!r && (hipMemcpy(pxy, d, n2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(pxz, d + n2, n2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
             hipMemcpy(pyz, d + 2 * n2, n2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) { p3m_set_error("projection: download failed"); r = P3M_EDEVICE; }
  (void)hipFree(d);
  return r;
}

// per-phase GPU times of the last whole step (timers.f90:68-77 prints such a table under -DMPI_TIME)
extern "C" int p3m_hip_phase_timing(p3m_ctx *c, int32_t on) {
  if (!c) return P3M_EINVAL;
  if (!c->pt) { c->pt = new PhaseTimer(); c->own_pt = true; }
  c->pt->on = on != 0; c->pt->reset();
  return P3M_OK;
}
extern "C" int p3m_hip_last_phase_ms(p3m_ctx *c, float *ms12) {
  if (!c || !ms12) return P3M_EINVAL;
  if (!c->pt || !c->pt->on) { p3m_set_error("p3m_hip_last_phase_ms: phase timing is off (p3m_hip_phase_timing)"); return P3M_ESTATE; }
  for (int k = 0; k < P3M_NPHASE; k++) ms12[k] = c->pt->ms[k];
  return P3M_OK;
}

extern "C" int p3m_hip_get_step_out(p3m_ctx *c, float a_mid, p3m_step_out *out) {
  if (!c || !out) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  const Geometry &g = c->g;
  P3M_TRY(reductions_download(c));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->finalize_queued) P3M_TRY(particles_finalize_finish(c, false));   // a ghost removal queued by the whole-step call
  particles_collect_counters(c);
  reductions_fold(c);
  P3M_TRY(rho_u8_check(c));
  p3m_step_out o; memset(&o, 0, sizeof(o));
  float fmax = sqrtf(c->h_red[0]);                         // :643
  float ppmax = c->h_red[1], cmax = c->h_red[2];
  // :617 assigns pp_ext_force_max(thread) per tile: every OpenMP thread keeps its LAST tile of the
  // static `!$omp do` chunk (:84-85); reproduce with `cores`.
  float emax = 0.f;
  if (c->p.flags & P3M_FLAG_PP_EXT) {
    const int cores = std::max(1, c->p.cores), nt = std::min(cores, g.ntiles), base = g.ntiles / nt, rem = g.ntiles % nt;
    int pos = 0;
    for (int t = 0; t < nt; t++) { pos += base + (t < rem ? 1 : 0); emax = std::max(emax, c->h_tile_ext[pos - 1]); }
  }
  double sums[2] = {c->h_sums[0], c->h_sums[1]}; int64_t npt = c->np_local;
  if (g.nodes > 1) {
    if (!c->have_transport) { p3m_set_error("multi-rank context without transport"); return P3M_ECOMM; }
    float v[4] = {fmax, ppmax, emax, cmax};
    if (c->transport.allreduce_max_f32(c->transport.user, v, 4)) return P3M_ECOMM;
    fmax = v[0]; ppmax = v[1]; emax = v[2]; cmax = v[3];
    double s[3] = {sums[0], sums[1], (double)npt};
    if (c->transport.allreduce_sum_f64(c->transport.user, s, 3)) return P3M_ECOMM;
    sums[0] = s[0]; sums[1] = s[1]; npt = (int64_t)s[2];
  }
  o.f_force_max = fmax; o.pp_force_max = ppmax; o.pp_ext_force_max = emax; o.c_force_max = cmax;
  o.dt_f_acc = 1.0f / sqrtf(fmaxf(0.0001f, fmax) * a_mid * P3M_G_F);                                  // :652
  o.dt_pp_acc = (c->p.flags & P3M_FLAG_PPINT) ? sqrtf(c->p.dt_pp_scale * c->p.rsoft) / fmaxf(sqrtf(ppmax * a_mid * P3M_G_F), 1e-3f) : c->last.dt_pp_acc;      // :668
  o.dt_pp_ext_acc = (c->p.flags & P3M_FLAG_PP_EXT) ? sqrtf(c->p.dt_pp_scale * c->p.rsoft) / fmaxf(sqrtf(emax * a_mid * P3M_G_F), 1e-3f) : c->last.dt_pp_ext_acc;  // :692
  o.dt_c_acc = sqrtf((float)g.ms / (cmax * a_mid * P3M_G_F));                                          // coarse_max_dt.f90:36
  o.sum_rho_f = sums[0]; o.sum_rho_c = sums[1];
  o.np_total = npt; o.np_local = c->np_local; o.np_ghost = c->np_ghost; o.np_deleted = c->np_deleted;
  c->last = o; *out = o;
  return P3M_OK;
}

// after an error in the middle of a step: nothing that was queued for "later" may be trusted by the next call -- the next sort
// counts its rows itself (k_row_hist), no survivor counts, no deferred counters, no half-finished ghost removal
void particles_reset_after_error(p3m_ctx *c) {
  c->hist_done = false; c->gl_valid = false; c->lazy_counters = false; c->finalize_queued = false; c->cnt_from_kick = 0; c->rho_from_sort = false; c->rho_u8 = false; c->rho_u8_force = false; c->cell_max_known = false; c->cell_max_reported = false; c->rho_u8_step = false; c->coarse_first = false; c->xinv_deferred = false; c->step_zeroed = false; c->zl.cnt = 0;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
}
static int particle_mesh_step(p3m_ctx *c, float a_mid, float dt, float dt_old, float mass_p, const float *offset, const float *move_back, p3m_step_out *out);
extern "C" int p3m_hip_particle_mesh(p3m_ctx *c, float a_mid, float dt, float dt_old, float mass_p, const float *offset,
                                     const float *move_back, p3m_step_out *out) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_particle_mesh"));
  c->step_begun = false;
  const int r = particle_mesh_step(c, a_mid, dt, dt_old, mass_p, offset, move_back, out);
  // an error behind the first state change (whatever its code: P3M_EINVAL can come from the coarse deposit, the fused kick or the CIC kick,
  // after step_prezero and the deferred inverse x pass were set up) leaves nothing queued for "later" to be trusted
  if (r != P3M_OK && c->step_begun) particles_reset_after_error(c);
  c->step_begun = false;
  return r;
}
static int particle_mesh_step(p3m_ctx *c, float a_mid, float dt, float dt_old, float mass_p, const float *offset, const float *move_back, p3m_step_out *out) {
  P3M_TRY(need_kernels(c));
  // every parameter / state check comes before the first state change (the drift)
  if (c->g.nodes != 1) { p3m_set_error("multi-rank contexts are stepped through a p3m_group (p3m_hip_group_*)"); return P3M_ECOMM; }
  if (c->pt) c->pt->reset();
  c->step_begun = true;
  { PhaseScope ps(c->pt, P3M_PH_DRIFT, c->stream); P3M_TRY(p3m_hip_update_position(c, dt, dt_old, offset)); }   // :56
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(step_prezero(c));
  { PhaseScope ps(c->pt, P3M_PH_GHOST, c->stream); P3M_TRY(particles_pass_self(c)); }                           // :61-63
  { PhaseScope ps(c->pt, P3M_PH_SORT, c->stream); P3M_TRY(particles_sort_enqueue(c, mass_p)); }
  P3M_TRY(particles_sort_finish(c, false));                  // no host wait: the sort's counters come in with the step's results
  // The coarse force depends on positions only: it is formed right after the sort, on a second stream underneath the
  // fine-mesh force sweep.  Its kick is then applied inside the fine kick's pass
  // (coarse_kick_rides_on_fine).
  const bool ride = coarse_kick_rides_on_fine(c);
  { PhaseScope ps(c->pt, P3M_PH_COARSE_DEPOSIT, c->stream); P3M_TRY(coarse_deposit(c, mass_p)); }   // coarse_mass
  if (c->stream2) {
    HIP_TRY(hipEventRecord(c->ev_dep, c->stream));
    P3M_TRY(fine_mesh_force_phase(c, mass_p, false));        // :72-204 of every tile, queued first
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_dep, 0));
    hipStream_t main = c->stream;
    c->stream = c->stream2;
    int r;
    { PhaseScope ps(c->pt, P3M_PH_COARSE_FORCE, c->stream2); r = coarse_force(c); }   // coarse_force, _buffer, max
    if (r == P3M_OK && hipEventRecord(c->ev_cf, c->stream2) != hipSuccess) r = P3M_EDEVICE;
    c->stream = main;
    if (r != P3M_OK) { (void)hipStreamSynchronize(c->stream2); return r; }
  } else {
    { PhaseScope ps(c->pt, P3M_PH_COARSE_FORCE, c->stream); P3M_TRY(coarse_force(c)); }
    P3M_TRY(fine_mesh_force_phase(c, mass_p, false));
  }
  if (ride) {
    if (c->stream2) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_cf, 0));
    c->coarse_first = true;
    const int r = fine_mesh_kick_phase(c, a_mid, dt, mass_p);   // :208-319 + coarse_velocity
    c->coarse_first = false;
    P3M_TRY(r);
  } else {
    P3M_TRY(fine_mesh_kick_phase(c, a_mid, dt, mass_p));     // :208-628
    if (c->stream2) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_cf, 0));
    { PhaseScope ps(c->pt, P3M_PH_COARSE_KICK, c->stream); P3M_TRY(coarse_kick(c, a_mid, dt)); }   // coarse_velocity (coarse_vel_update = .true., cubepm.par:87)
  }
  // :716-720; the survivor count, the sort's counters and the step's maxima and sums reach the host behind ONE wait
  { PhaseScope ps(c->pt, P3M_PH_DELETE, c->stream); P3M_TRY(particles_finalize_enqueue(c, (c->p.flags & P3M_FLAG_MOVE_GRID_BACK) ? move_back : nullptr)); }
  c->step_zeroed = false;
  p3m_step_out o;
  P3M_TRY(p3m_hip_get_step_out(c, a_mid, &o));               // :643-706 (synchronises the stream)
  if (c->pt && c->pt->on) { if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2)); c->pt->collect(); }
  if (out) *out = o;
  return P3M_OK;
}

// ------------------------------------------------------------------ probes
extern "C" int p3m_hip_probe_tile_density(p3m_ctx *c, int32_t tx, int32_t ty, int32_t tz, float mass_p, float *rho_f) {
  if (!c || !rho_f) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_probe_tile_density"));
  const Geometry &g = c->g;
  if (tx < 0 || ty < 0 || tz < 0 || tx >= g.T || ty >= g.T || tz >= g.T) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  const int tile = (tz * g.T + ty) * g.T + tx;
  P3M_TRY(fine_deposit(c, tile, 1, mass_p));
  HIP_TRY(hipMemcpy2DAsync(rho_f, sizeof(float) * (g.nf + 2), c->rho, sizeof(float) * 2 * g.px, sizeof(float) * (g.nf + 2), (size_t)g.nf * g.nf,
                           hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return P3M_OK;
}

extern "C" int p3m_hip_probe_tile_force(p3m_ctx *c, const float *rho_f, float *force_f, float *force_max2) {
  if (!c || !rho_f || !force_f) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_probe_tile_force"));
  if (!c->have_kf) { p3m_set_error("fine kernel not set"); return P3M_ESTATE; }
  const Geometry &g = c->g;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy2DAsync(c->rho, sizeof(float) * 2 * g.px, rho_f, sizeof(float) * (g.nf + 2), sizeof(float) * (g.nf + 2), (size_t)g.nf * g.nf,
                           hipMemcpyHostToDevice, c->stream));
  P3M_TRY(fine_force(c, 0, 1));
  const size_t boxsz = (size_t)g.fb * g.fb * g.fb, boxp = (size_t)g.fb * g.fb * g.fbp;
  std::vector<float> tmp(3 * boxsz);
  for (int comp = 0; comp < 3; comp++)
    HIP_TRY(hipMemcpy2DAsync(tmp.data() + comp * boxsz, sizeof(float) * g.fb, c->fbox + (size_t)comp * g.ntiles * boxp, sizeof(float) * g.fbp,
                             sizeof(float) * g.fb, (size_t)g.fb * g.fb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float m2 = 0.f;
  for (size_t i = 0; i < boxsz; i++) {
    const float a = tmp[i], b = tmp[boxsz + i], d = tmp[2 * boxsz + i];
    force_f[3 * i] = a; force_f[3 * i + 1] = b; force_f[3 * i + 2] = d;
    m2 = std::max(m2, a * a + b * b + d * d);
  }
  if (force_max2) *force_max2 = m2;
  return P3M_OK;
}

extern "C" int p3m_hip_probe_coarse(p3m_ctx *c, float mass_p, float *rho_c, float *force_c) {
  if (!c) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_probe_coarse"));
  const Geometry &g = c->g;
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(reductions_clear(c));
  P3M_TRY(coarse_deposit(c, mass_p));
  if (rho_c) HIP_TRY(hipMemcpyAsync(rho_c, c->rho_c, sizeof(float) * (size_t)g.ncn * g.ncn * g.ncn, hipMemcpyDeviceToHost, c->stream));
  if (force_c) {
    P3M_TRY(need_kernels(c));
    P3M_TRY(coarse_force(c));
    const size_t fcs = (size_t)(g.ncn + 2) * (g.ncn + 2) * (g.ncn + 2);
    std::vector<float> tmp(3 * fcs);
    HIP_TRY(hipMemcpyAsync(tmp.data(), c->force_c, sizeof(float) * 3 * fcs, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    planes_to_aos(tmp.data(), force_c, fcs);
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return P3M_OK;
}

extern "C" int p3m_hip_fft3d(p3m_ctx *c, float *data, int32_t n, int32_t dir) {
  if (!c || !data) return P3M_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  FftPlan pl; P3M_TRY(fft_plan_create(&pl, n));
  float *d = nullptr, *t1 = nullptr, *t2 = nullptr; const size_t S = (size_t)(2 * pl.px) * n * n;
  int r = dalloc(&d, S); if (!r) r = dalloc(&t1, S); if (!r) r = dalloc(&t2, S);
  if (!r) {
    hipError_t e = hipMemsetAsync(d, 0, sizeof(float) * S, c->stream);
    if (e == hipSuccess) e = hipMemcpy2DAsync(d, sizeof(float) * 2 * pl.px, data, sizeof(float) * (n + 2), sizeof(float) * (n + 2), (size_t)n * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      if (dir > 0) { r = fft3d_forward(c, pl, d, t1, 1); if (!r) r = fft_lz_to_rows(c, pl, d, t2); }               // reference rows <- LZ
      else { r = fft_rows_to_lz(c, pl, d, t1); if (!r) r = fft3d_inverse(c, pl, t1, d, t2, 1, nullptr); }
    }
    if (!r) e = hipMemcpy2DAsync(data, sizeof(float) * (n + 2), t2, sizeof(float) * 2 * pl.px, sizeof(float) * (n + 2), (size_t)n * n, hipMemcpyDeviceToHost, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess || e != hipSuccess) { p3m_set_error("fft3d probe: HIP error"); r = r ? r : P3M_EDEVICE; }
  }
  dfree(d); dfree(t1); dfree(t2); fft_plan_destroy(&pl);
  return r;
}

extern "C" int p3m_hip_time_fine_sweep(p3m_ctx *c, float mass_p, int32_t reps, float *ms_per_sweep) {
  if (!c || reps < 1 || !ms_per_sweep) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_time_fine_sweep"));
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(need_kernels(c));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  P3M_TRY(fine_sweep(c, mass_p, false));  // warm-up
  HIP_TRY(hipEventRecord(e0, c->stream));
  for (int i = 0; i < reps; i++) P3M_TRY(fine_sweep(c, mass_p, false));
  HIP_TRY(hipEventRecord(e1, c->stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *ms_per_sweep = ms / reps;
  return P3M_OK;
}

int fft_single_pass(p3m_ctx *c, const FftPlan &pl, int which, float *data, float *work, const float *kern, int batch, float *box, int fb, int lo,
                    int64_t bcs);

extern "C" int p3m_hip_time_fine_gather(p3m_ctx *c, int32_t reps, float *ms_per_pass) {
  if (!c || reps < 1 || !ms_per_pass) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_time_fine_gather"));
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(need_kernels(c));
  if (c->np_all == 0) { p3m_set_error("p3m_hip_time_fine_gather: no sorted records (run a step or link_list_and_pass first)"); return P3M_ESTATE; }
  const bool cf = c->coarse_first;
  c->coarse_first = false;                      // the fine kick alone (the coarse force of a finished step may be gone)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  auto body = [&]() -> int {
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    P3M_TRY(fine_max_and_kick(c, 0.5f, 0.0f, false));  // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; i++) P3M_TRY(fine_max_and_kick(c, 0.5f, 0.0f, false));
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    return P3M_OK;
  };
  const int rc = body();
  c->coarse_first = cf;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  P3M_TRY(rc);
  *ms_per_pass = ms / reps;
  return P3M_OK;
}
extern "C" int p3m_hip_time_fft_pass(p3m_ctx *c, int32_t which, int32_t reps, float *ms_per_launch, int32_t *batch) {
  if (!c || reps < 1 || !ms_per_launch) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_time_fft_pass"));
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(need_kernels(c));
  const Geometry &g = c->g;
  const int nt = std::min(c->tile_batch, g.ntiles);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  const int64_t bcs = (int64_t)g.ntiles * g.fb * g.fb * g.fbp;
  auto one = [&]() { return which == 7 ? fine_time_fused_kick(c) : fft_single_pass(c, c->plan_f, which, c->rho, c->work, c->kern_f, nt, c->fbox, g.fb, g.nb - 2, bcs); };
  P3M_TRY(one());  // warm-up
  HIP_TRY(hipEventRecord(e0, c->stream));
  for (int i = 0; i < reps; i++) P3M_TRY(one());
  HIP_TRY(hipEventRecord(e1, c->stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *ms_per_launch = ms / reps;
  if (batch) *batch = nt;
  return P3M_OK;
}

// ------------------------------------------------------------------ F77-ABI wrapper (style of pp_force_c_, nbody-ueli.cu:368)
extern "C" void particle_mesh_hip_(int64_t *handle, const p3m_params *params, const float *fine_table, const float *coarse_table,
                                   float *xv, int64_t *pid, int32_t *np_local, const float *a_mid, const float *dt, const float *dt_old,
                                   const float *mass_p, const float *offset, const float *move_back, float *dt_f_acc, float *dt_pp_acc,
                                   float *dt_pp_ext_acc, float *dt_c_acc, int32_t *ierr) {
  int r = P3M_OK;
  p3m_ctx *c = reinterpret_cast<p3m_ctx *>(static_cast<intptr_t>(*handle));
  if (!c) {
    r = p3m_hip_create(params, &c);
    if (!r) r = p3m_hip_set_kernel_tables(c, fine_table, coarse_table);
    if (r) { if (c) p3m_hip_destroy(c); *ierr = r; return; }
    *handle = static_cast<int64_t>(reinterpret_cast<intptr_t>(c));
  }
  p3m_step_out o;
  r = p3m_hip_upload_particles(c, xv, pid, *np_local);
  if (!r) r = p3m_hip_particle_mesh(c, *a_mid, *dt, *dt_old, *mass_p, offset, move_back, &o);
  if (!r) r = p3m_hip_download_particles(c, xv, pid, np_local);
  if (!r) { *dt_f_acc = o.dt_f_acc; *dt_pp_acc = o.dt_pp_acc; *dt_pp_ext_acc = o.dt_pp_ext_acc; *dt_c_acc = o.dt_c_acc; }
  *ierr = r;
}
