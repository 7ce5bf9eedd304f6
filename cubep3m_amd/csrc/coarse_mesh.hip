// coarse_mesh.hip -- long-range force on the mesh_scale-times coarser global mesh (coarse_mesh.f90):
//   coarse_mass.f90 + coarse_cic_mass(.f90|_buffer.f90) -> k_coarse_moments + k_coarse_collect (no atomics)
//   coarse_force.f90 + fftw3ds.f90 (single rank: cube == slab)      -> coarse_force
//   coarse_force_buffer.f90 (periodic 1-cell halo)                   -> k_coarse_unpack
//   coarse_max_dt.f90                                                -> k_coarse_max
//   coarse_velocity.f90:137-179                                      -> k_coarse_kick
//   coarse_kernel (kernel_initialization.f90:272-732)                -> build_coarse_kernel
#include "p3m_internal.h"
#include <algorithm>

struct CGeo { int nb, E, Nn, ms, ncn, nc, cngp; };   // cngp: -DCOARSE_NGP (whole weight on cell i2; coarse_cic_mass.f90:21-24, coarse_velocity.f90:146-149)

// CIC deposit without atomics.  A particle's footprint is the cells i1 = floor(x/ms - 0.5) + 1 and i1+1 per
// axis (coarse_cic_mass.f90:18-21), clipped to 1..ncn (coarse_cic_mass_buffer.f90:59-113).  All particles with
// the same (i1,j1,k1) share their 8 target cells, and they are exactly the records of ms^3 fine cells = ms^2
// contiguous record ranges of the sorted store.  Pass 1: one thread per (i1,j1,k1) in [0,ncn]^3 sums the 8
// corner weights of its records in registers (every record is read once, in sorted order) and stores them
// SoA in mom[corner][k1][j1][i1].  Pass 2: one thread per coarse cell adds the 8 moments that land on it.
// (LDS float atomics cost ~2.5 clocks per lane on this part: a scatter into an LDS tile spent 80 % of its time
// in ds_add_f32, and a gather per output cell re-reads cell_end 1.5 times.)
// The chain window hoc(0:ncn+1) of coarse_mass.f90:85-87 is implied: i1 in [0,ncn] <=> x in [-ms/2, Nn+ms/2).
#ifndef CM_FLY
#define CM_FLY 4
#endif
#define CM_CAP 16   // listed record indices per lane
#define CM_HEAVY 24 // records of a cube from which the whole wavefront sums it: this, or three times the mean per cube if that is more
                    // (measured at a mean of 8: 12 -> the uniform case 60 % slower, 16 / 24 / 48 / 96 -> clustered 251 / 258 / 287 / 380 us)
__global__ __launch_bounds__(256) void k_coarse_moments(const float4 *__restrict__ spos, const int *__restrict__ cs, float *__restrict__ mom, CGeo G,
                                                       float mass_p, float *__restrict__ rho_c, const int *__restrict__ crow, int crow_w, int heavy_min) {
  const int m1 = G.ncn + 1;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, tot = (int64_t)m1 * m1 * m1;
  // (threads past the last cube stay: the heavy cubes below are wavefront operations)
  const int64_t tc = t < tot ? t : tot - 1;
  const int ci = (int)(tc % m1), cj = (int)((tc / m1) % m1), ck = (int)(tc / ((int64_t)m1 * m1));   // i1, j1, k1
  const int h = G.ms / 2;
  const float inv = 1.0f / (float)G.ms;
  const int x0 = G.ms * ci - h + G.nb, y0 = G.ms * cj - h + G.nb, z0 = G.ms * ck - h + G.nb;      // first fine cell of the cube, extended index
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) acc[c] = 0.f;
  // the sums of one record into a[8], for the cube (ci_, cj_, ck_)
  auto add_to = [&](const float4 &p, int ci_, int cj_, int ck_, float (&a)[8]) {
    const float x = inv * p.x - 0.5f, y = inv * p.y - 0.5f, z = inv * p.z - 0.5f;     // coarse_cic_mass.f90:18
    const int i1 = (int)floorf(x) + 1, j1 = (int)floorf(y) + 1, k1 = (int)floorf(z) + 1;  // 1-based
    float dx1 = (float)i1 - x, dy1 = (float)j1 - y, dz1 = (float)k1 - z;
    float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1, dz2 = 1.0f - dz1;
    if (G.cngp) { dx1 = dy1 = dz1 = 0.0f; dx2 = dy2 = dz2 = 1.0f; }                     // :21-24
    dx1 = mass_p * dx1; dx2 = mass_p * dx2;                                             // :32-33
    if (i1 == ci_ && j1 == cj_ && k1 == ck_) {
      a[0] += dx1 * dy1 * dz1; a[1] += dx2 * dy1 * dz1; a[2] += dx1 * dy2 * dz1; a[3] += dx2 * dy2 * dz1;
      a[4] += dx1 * dy1 * dz2; a[5] += dx2 * dy1 * dz2; a[6] += dx1 * dy2 * dz2; a[7] += dx2 * dy2 * dz2;
    } else {
      // the fp32 expression put the record into a neighbouring cube (possible when 1/ms is inexact): rare, exact, slow
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int I = i1 + (c & 1), J = j1 + ((c >> 1) & 1), K = k1 + (c >> 2);
        if (I >= 1 && I <= G.ncn && J >= 1 && J <= G.ncn && K >= 1 && K <= G.ncn)
          atomicAdd(&rho_c[((int64_t)(K - 1) * G.ncn + (J - 1)) * G.ncn + (I - 1)], ((c & 1) ? dx2 : dx1) * (((c >> 1) & 1) ? dy2 : dy1) * ((c >> 2) ? dz2 : dz1));
      }
    }
  };
  auto add = [&](const float4 &p) { add_to(p, ci, cj, ck, acc); };
  const int nrow = G.ms * G.ms;
  auto row_range = [&](int ci_, int y0_, int z0_, int x0_, int r, int &a0, int &a1) {
    const int zz = r / G.ms, yy = r - zz * G.ms;
    // ONE pair of loads for both tables (the table, its pitch and the distance of the range's end are uniform selections): as two branches
    // the sixteen ranges of a cube went out one after the other on the full-table path -- each load waited for the other branch's load into
    // the same register (k_coarse_moments 255 us with the full table, 177 with the compact one)
    const int *tab = crow ? crow + ci_ : cs + x0_;
    const int pitch = crow ? crow_w : G.E, last = crow ? 1 : G.ms;
    const int *row = tab + ((int64_t)(z0_ + zz) * G.E + (y0_ + yy)) * pitch; a0 = row[0]; a1 = row[last];
  };
  // A cube inside a blob holds hundreds of records where its 63 neighbours in the wavefront hold eight: the lane that owns it would
  // walk them alone (clustered input: 645 us per rank against 160 us).  Such cubes (more than `heavy_min` records: CM_HEAVY) are taken one at a
  // time by the whole wavefront -- the ms^2 ranges flattened over the lanes, eight partial sums per lane, a butterfly at the end --
  // and their owner only receives the totals.  (The order of the sums inside such a cube differs from the serial walk: rounding
  // of the last bit, as with any other order of the records of a cell.)
  bool mine_done = false;
  constexpr int NRK = 16;                                // rows whose ranges are kept in registers between the count and the listing (ms <= 4)
  int ka0[NRK], ka1[NRK];
  const bool kept = nrow <= NRK;
  if (nrow <= 64) {
    const int lane = threadIdx.x & 63;
    int total = 0;
    if (kept) {
#pragma unroll
      for (int r = 0; r < NRK; r++) { ka0[r] = 0; ka1[r] = 0; if (r < nrow && t < tot) row_range(ci, y0, z0, x0, r, ka0[r], ka1[r]); total += ka1[r] - ka0[r]; }
    } else if (t < tot) for (int r = 0; r < nrow; r++) { int a0, a1; row_range(ci, y0, z0, x0, r, a0, a1); total += a1 - a0; }
    bool heavy = t < tot && total > heavy_min;
    unsigned long long pending;
    while ((pending = __ballot(heavy)) != 0ull) {
      const int lead = __ffsll((long long)pending) - 1;
      const int lci = __builtin_amdgcn_readlane(ci, lead), lcj = __builtin_amdgcn_readlane(cj, lead), lck = __builtin_amdgcn_readlane(ck, lead);
      const int lx0 = G.ms * lci - h + G.nb, ly0 = G.ms * lcj - h + G.nb, lz0 = G.ms * lck - h + G.nb;
      int ra0 = 0, ra1 = 0;                              // lane r < nrow: the range of row r of the lead's cube
      if (lane < nrow) row_range(lci, ly0, lz0, lx0, lane, ra0, ra1);
      const int inc = wave_scan_incl_i(ra1 - ra0);       // inclusive prefix of the lengths over the lanes
      const int ltot = __builtin_amdgcn_readlane(inc, 63);
      float pa[8];
#pragma unroll
      for (int c = 0; c < 8; c++) pa[c] = 0.f;
      const int rlen = ra1 - ra0;                        // formed at full EXEC: the readlanes below take it from lanes that hold no record
      for (int q0 = 0; q0 < ltot; q0 += 64) {            // uniform trip count: every readlane runs with all lanes active
        const int q = q0 + lane;
        int idx = 0;                                     // record q of the cube: in the row whose prefix interval holds q
        for (int r = 0; r < nrow; r++) {
          const int hi = __builtin_amdgcn_readlane(inc, r), len = __builtin_amdgcn_readlane(rlen, r), st = __builtin_amdgcn_readlane(ra0, r);
          if (q >= hi - len && q < hi) idx = st + (q - (hi - len));
        }
        if (q < ltot) add_to(spos[idx], lci, lcj, lck, pa);
      }
#pragma unroll
      for (int c = 0; c < 8; c++) {
        float v = pa[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == lead) acc[c] = v;
      }
      if (lane == lead) { heavy = false; mine_done = true; }
    }
  }
  if (!mine_done && t < tot) {
    // Listed: a lane's records come from ms^2 short ranges (half a record each at the reference's density).  Walking them
    // row by row makes the wavefront run `add` once per row and again for every extra record any lane has in that row --
    // several times the work of its busiest lane (PMC: 58 % VALU-busy at 34 % active lanes).  Instead the row loop only
    // lists the record indices in LDS (coalesced table loads, no arithmetic), and the lanes then take their records four at
    // a time, loads first: the wavefront runs `add` about as often as its busiest lane has records, with four loads in
    // flight per lane.  A full list is worked off before the row loop goes on, so the order of the sums never changes.
    extern __shared__ int lst[];                         // [CM_CAP][256]
    int cnt = 0;
    auto drain = [&]() {
      for (int k = 0; k < cnt; k += CM_FLY) {
        float4 q[CM_FLY];
#pragma unroll
        for (int u = 0; u < CM_FLY; u++) { q[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (k + u < cnt) q[u] = spos[lst[(k + u) * 256 + threadIdx.x]]; }
#pragma unroll
        for (int u = 0; u < CM_FLY; u++) if (k + u < cnt) add(q[u]);
      }
      cnt = 0;
    };
    if (kept) {
#pragma unroll
      for (int r = 0; r < NRK; r++)
        for (int sidx = ka0[r]; sidx < ka1[r]; sidx++) {
          if (cnt == CM_CAP) drain();
          lst[cnt * 256 + threadIdx.x] = sidx; cnt++;
        }
    } else
    for (int r = 0; r < nrow; r++) {
      int a0, a1;
      row_range(ci, y0, z0, x0, r, a0, a1);
      for (int sidx = a0; sidx < a1; sidx++) {
        if (cnt == CM_CAP) drain();
        lst[cnt * 256 + threadIdx.x] = sidx; cnt++;
      }
    }
    drain();
  }
  if (t >= tot) return;
#pragma unroll
  for (int c = 0; c < 8; c++) mom[c * tot + t] = acc[c];
}
// rho_c(I,J,K) (1-based) = sum over corners c = (cx,cy,cz) of mom[c](I-cx, J-cy, K-cz); added to what the rare
// path of pass 1 left in rho_c (zeroed before pass 1)
__global__ __launch_bounds__(256) void k_coarse_collect(const float *__restrict__ mom, float *__restrict__ rho_c, int ncn, double *__restrict__ sum_out) {
  __shared__ float sh[4];
  const int m1 = ncn + 1;
  const int64_t n3 = (int64_t)ncn * ncn * ncn, tot = (int64_t)m1 * m1 * m1;
  float part = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n3; t += (int64_t)gridDim.x * 256) {
    const int I = (int)(t % ncn) + 1, J = (int)((t / ncn) % ncn) + 1, K = (int)(t / ((int64_t)ncn * ncn)) + 1;
    float v = rho_c[t];
#pragma unroll
    for (int c = 0; c < 8; c++) v += mom[c * tot + ((int64_t)(K - (c >> 2)) * m1 + (J - ((c >> 1) & 1))) * m1 + (I - (c & 1))];
    rho_c[t] = v;
    part += v;
  }
  // one double atomic per workgroup (a per-wave atomic on one address serialises)
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0 && sum_out) { const float s4 = (sh[0] + sh[1]) + (sh[2] + sh[3]); if (s4 != 0.f) atomicAdd(sum_out, (double)s4); }  // coarse_mesh.f90:31-43
}

int coarse_deposit(p3m_ctx *c, float mass_p) {
  const Geometry &g = c->g;
  CGeo G{g.nb, g.E, g.Nn, g.ms, g.ncn, g.nc, (c->p.flags & P3M_FLAG_COARSE_NGP) ? 1 : 0};
  if (g.ms / 2 > g.nb) { p3m_set_error("coarse_deposit: mesh_scale/2 > nf_buf"); return P3M_EINVAL; }
  const int64_t m1 = g.ncn + 1, tot = m1 * m1 * m1, n3 = (int64_t)g.ncn * g.ncn * g.ncn;
  if (!c->step_zeroed) HIP_TRY(hipMemsetAsync(c->rho_c, 0, sizeof(float) * n3, c->stream));   // (whole steps: step_prezero)
  const int *crow = c->cells_compact ? (const int *)c->crow : (const int *)nullptr;
  hipLaunchKernelGGL(k_coarse_moments, dim3((unsigned)cdiv(tot, 256)), dim3(256), sizeof(int) * CM_CAP * 256, c->stream, (const float4 *)c->spos, (const int *)c->cell_end, c->cmom, G,
                     mass_p, c->rho_c, crow, c->crow_w, (int)std::max<int64_t>(CM_HEAVY, 3 * (int64_t)c->np_all / tot));
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_coarse_collect, dim3((unsigned)std::min<int64_t>(1024, cdiv(n3, 256))), dim3(256), 0, c->stream, (const float *)c->cmom, c->rho_c, g.ncn, c->d_sums + 1 * P3M_SUM_SPAN);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// cube -> slab (pack_slab, fftw3ds.f90:4-54); single rank: a pitch change only
__global__ __launch_bounds__(256) void k_cube_to_slab(const float *__restrict__ cube, float *__restrict__ slab, int n, int rp) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tot = (int64_t)n * n * rp;
  if (idx >= tot) return;
  const int i = (int)(idx % rp); const int64_t r = idx / rp;
  slab[idx] = (i < n) ? cube[r * n + i] : 0.f;
}
// slab -> force_c(comp, 0:ncn+1, ...) with the periodic 1-cell halo (unpack_slab + coarse_force_buffer.f90)
__global__ __launch_bounds__(256) void k_slab_to_force(const float *__restrict__ slab, float *__restrict__ fc, int n, int rp) {
  const int m = n + 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)m * m * m) return;
  const int i = (int)(idx % m), j = (int)((idx / m) % m), k = (int)(idx / ((int64_t)m * m));
  const int gi = (i - 1 + n) % n, gj = (j - 1 + n) % n, gk = (k - 1 + n) % n;
  fc[idx] = slab[((int64_t)gk * n + gj) * rp + gi];
}
// coarse_max_dt.f90:24-31
__global__ __launch_bounds__(256) void k_coarse_max(const float *__restrict__ fc, int n, float *__restrict__ out) {
  const int m = n + 2; const int64_t cs = (int64_t)m * m * m;
  float mx = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)n * n * n; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx % n), j = (int)((idx / n) % n), k = (int)(idx / ((int64_t)n * n));
    const int64_t o = ((int64_t)(k + 1) * m + (j + 1)) * m + (i + 1);
    const float a = fc[o], b = fc[o + cs], d = fc[o + 2 * cs];
    mx = fmaxf(mx, sqrtf(a * a + b * b + d * d));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
  if ((threadIdx.x & 63) == 0) p3m_atomic_max_nonneg(out + p3m_slot() * 16, mx);
}

int coarse_force(p3m_ctx *c) {
  const Geometry &g = c->g;
  if (g.nodes != 1) { p3m_set_error("coarse_force: multi-rank path needs the slab transport"); return P3M_ECOMM; }
  const int n = g.nc;
  const int rp = 2 * g.pxc;
  const int64_t tot = (int64_t)n * n * rp;
  hipLaunchKernelGGL(k_cube_to_slab, dim3(cdiv(tot, 256)), dim3(256), 0, c->stream, (const float *)c->rho_c, c->slab, n, rp);
  HIP_TRY(hipGetLastError());
  P3M_TRY(fft3d_forward(c, c->plan_c, c->slab, c->slab_w, 1));                               // coarse_force.f90:18
  const size_t kplane = (size_t)n * n * g.pxc;
  const int64_t fcs = (int64_t)(n + 2) * (n + 2) * (n + 2);
  for (int comp = 0; comp < 3; comp++) {
    P3M_TRY(fft3d_inverse(c, c->plan_c, c->slab, c->slab_w, c->slab_o, 1, c->kern_c + comp * kplane));  // :37-50
    hipLaunchKernelGGL(k_slab_to_force, dim3(cdiv(fcs, 256)), dim3(256), 0, c->stream, (const float *)c->slab_o, c->force_c + comp * fcs, n, rp);
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(k_coarse_max, dim3(std::min<int64_t>(1024, cdiv((int64_t)n * n * n, 256))), dim3(256), 0, c->stream, (const float *)c->force_c,
                     n, c->d_red + 2 * P3M_RED_SPAN);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// coarse_velocity.f90:137-179: CIC gather of force_c (with halo) at x/ms - 0.5, particles of hoc(1..ncn)
__global__ __launch_bounds__(256) void k_coarse_kick(const float4 *__restrict__ spos, float4 *__restrict__ vel, int n, CGeo G,
                                                     const float *__restrict__ fc, float a_mid, float dt) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n) return;
  const float4 p = spos[s];
  const float fNn = (float)G.Nn;
  if (!(p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn)) return;
  const float inv = 1.0f / (float)G.ms;
  const float x = inv * p.x - 0.5f, y = inv * p.y - 0.5f, z = inv * p.z - 0.5f;             // :143
  const int i1 = (int)floorf(x) + 1, j1 = (int)floorf(y) + 1, k1 = (int)floorf(z) + 1;
  float dx1 = (float)i1 - x, dy1 = (float)j1 - y, dz1 = (float)k1 - z;
  float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1, dz2 = 1.0f - dz1;
  if (G.cngp) { dx1 = dy1 = dz1 = 0.0f; dx2 = dy2 = dz2 = 1.0f; }                           // :146-149
  const int m = G.ncn + 2; const int64_t cs = (int64_t)m * m * m;
  const int vi = rec_index(p);   // the velocity stays in arrival order (p3m_internal.h)
  float4 v = vel[vi];
#pragma unroll
  for (int cz = 0; cz < 2; cz++)
#pragma unroll
    for (int cy = 0; cy < 2; cy++)
#pragma unroll
      for (int cx = 0; cx < 2; cx++) {                                                        // :153-168
        const float dV = a_mid * P3M_G_F * dt * (cx ? dx2 : dx1) * (cy ? dy2 : dy1) * (cz ? dz2 : dz1);
        const int64_t o = ((int64_t)(k1 + cz) * m + (j1 + cy)) * m + (i1 + cx);
        v.x = v.x + fc[o] * dV; v.y = v.y + fc[o + cs] * dV; v.z = v.z + fc[o + 2 * cs] * dV;
      }
  vel[vi] = v;
}

int coarse_kick(p3m_ctx *c, float a_mid, float dt) {
  const Geometry &g = c->g;
  if (c->np_all == 0) return P3M_OK;
  CGeo G{g.nb, g.E, g.Nn, g.ms, g.ncn, g.nc, (c->p.flags & P3M_FLAG_COARSE_NGP) ? 1 : 0};
  hipLaunchKernelGGL(k_coarse_kick, dim3(cdiv(c->np_all, 256)), dim3(256), 0, c->stream, (const float4 *)c->spos, c->vel, c->np_all, G,
                     (const float *)c->force_c, a_mid, dt);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// ------------------------------------------------------------------ coarse_kernel (kernel_initialization.f90:272-732), single rank
// ck = -r/r^3 on the mesh_scale-spaced periodic lattice (:302-336), the 4^3 corner and its signed
// mirror images from the table (:366-406): component d flips sign when axis d is mirrored.
__global__ __launch_bounds__(256) void k_coarse_kernel_real(float *__restrict__ slab, const float *__restrict__ table, int n, int rp, int ms, int comp,
                                                            int use_table) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tot = (int64_t)n * n * rp;
  if (idx >= tot) return;
  const int i = (int)(idx % rp); const int64_t r = idx / rp; const int j = (int)(r % n), k = (int)(r / n);
  float v = 0.f;
  if (i < n) {
    const int c3[3] = {i, j, k};
    float xs[3]; int t3[3]; bool in_table = use_table != 0; float sgn = 1.f;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const float w = (c3[d] < n / 2 + 1) ? (float)c3[d] : (float)(c3[d] - n);                // :304-308
      xs[d] = (float)ms * w;
      if (c3[d] < 4) t3[d] = c3[d];
      else if (c3[d] > n - 4) { t3[d] = n - c3[d]; if (d == comp) sgn = -sgn; }
      else in_table = false;
    }
    if (in_table) v = sgn * table[(((int64_t)t3[2] * 4 + t3[1]) * 4 + t3[0]) * 3 + comp];
    else {
      const float rr = sqrtf(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
      v = (rr == 0.0f) ? 0.f : -xs[comp] / (rr * rr * rr);                                    // :327-333
    }
  }
  slab[idx] = v;
}
// LRCKCORR, :562-591: Im K_c <- Im K_c^{corr} * (wc / Im K_c^{uncorr}) for integer |k| <= 8, k_c != 0
__global__ __launch_bounds__(256) void k_lrck(float *__restrict__ kern, const float *__restrict__ uncorr, int n, int px, int comp) {
  // kern / uncorr are in the bundle layout LZ: [y][chunk][z][16]
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * n * px) return;
  const int l = (int)(idx % 16), k = (int)((idx / 16) % n), nchunk = px / 16;
  const int chunk = (int)((idx / (16 * (int64_t)n)) % nchunk), j = (int)(idx / (16 * (int64_t)n * nchunk));
  const int kx = chunk * 16 + l;
  if (kx > n / 2) return;  // pad columns
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
__global__ __launch_bounds__(256) void k_take_imag_c(const float *__restrict__ hat, float *__restrict__ kern, int64_t ncomplex) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < ncomplex) kern[i] = hat[2 * i + 1];
}

int build_coarse_kernel(p3m_ctx *c, const float *table4_host) {
  const Geometry &g = c->g;
  if (g.nodes != 1) { p3m_set_error("build_coarse_kernel: multi-rank path needs the slab transport"); return P3M_ECOMM; }
  const int n = g.nc;
  float *d_table = nullptr;
  HIP_TRY(hipMalloc(&d_table, sizeof(float) * 192));
  HIP_TRY(hipMemcpyAsync(d_table, table4_host, sizeof(float) * 192, hipMemcpyHostToDevice, c->stream));
  const int rp = 2 * g.pxc;
  const int64_t tot = (int64_t)n * n * rp, ncx = (int64_t)n * n * g.pxc;
  float *unc = nullptr;
  const bool lr = (c->p.flags & P3M_FLAG_LRCKCORR) != 0;
  if (lr) HIP_TRY(hipMalloc(&unc, sizeof(float) * ncx));
  for (int comp = 0; comp < 3; comp++) {
    if (lr) {
      hipLaunchKernelGGL(k_coarse_kernel_real, dim3(cdiv(tot, 256)), dim3(256), 0, c->stream, c->slab, (const float *)d_table, n, rp, g.ms, comp, 0);
      P3M_TRY(fft3d_forward(c, c->plan_c, c->slab, c->slab_w, 1));
      hipLaunchKernelGGL(k_take_imag_c, dim3(cdiv(ncx, 256)), dim3(256), 0, c->stream, (const float *)c->slab, unc, ncx);
    }
    hipLaunchKernelGGL(k_coarse_kernel_real, dim3(cdiv(tot, 256)), dim3(256), 0, c->stream, c->slab, (const float *)d_table, n, rp, g.ms, comp, 1);
    HIP_TRY(hipGetLastError());
    P3M_TRY(fft3d_forward(c, c->plan_c, c->slab, c->slab_w, 1));
    hipLaunchKernelGGL(k_take_imag_c, dim3(cdiv(ncx, 256)), dim3(256), 0, c->stream, (const float *)c->slab, c->kern_c + comp * ncx, ncx);
    if (lr) hipLaunchKernelGGL(k_lrck, dim3(cdiv(ncx, 256)), dim3(256), 0, c->stream, c->kern_c + comp * ncx, (const float *)unc, n, g.pxc, comp);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  (void)hipFree(d_table);
  if (unc) (void)hipFree(unc);
  c->have_kc = true;
  return P3M_OK;
}


// ------------------------------------------------------------------ coarse_power.f90:2-139: mass power spectrum of the coarse density
// The reference transforms rho-hat back, forms the overdensity rho_c / rho_c_mean - 1 and transforms again (:27-35): away
// from k = 0 that is rho-hat / rho_c_mean, which is what is binned here (the k = 0 mode is skipped by :62 anyway), straight
// from the rho-hat the coarse force pass left in LZ order ([ky][chunk][kz][16 kx]).  Per mode as the reference writes it
// (:60-98): the kx = 0 plane counts each conjugate pair once, bin k1 = ceiling(|k|) with weight 1, and the sinc^4
// deconvolution divides the imaginary part's square only.  Sums in double (the reference adds in real(4)).
// ps_acc: [2][nc+2] doubles (weights, power); `planes` ky rows starting at ky0 and nchunk*16 kx columns starting at kx0 (a
// whole mesh, one rank's ky slab, or one rank's ky slab of its own kx chunks in the pencil decomposition).
__global__ __launch_bounds__(256) void k_coarse_power(const float2 *__restrict__ lz, int planes, int ky0, int kx0, int nc, int nchunk, float inv_mean,
                                                      double *__restrict__ ps_acc) {
  extern __shared__ double bins[];   // [2][nc+2]
  const int nb = nc + 2;
  for (int i = threadIdx.x; i < 2 * nb; i += 256) bins[i] = 0.0;
  __syncthreads();
  const int64_t tot = (int64_t)planes * nchunk * nc * 16;
  const int hc = nc / 2;
  const float fnc = (float)nc, n3 = fnc * fnc * fnc;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (int64_t)gridDim.x * 256) {
    const int l = (int)(idx & 15), kzi = (int)((idx >> 4) % nc), chunk = (int)((idx / (16 * (int64_t)nc)) % nchunk), o = (int)(idx / (16 * (int64_t)nc * nchunk));
    const int kxi = kx0 + chunk * 16 + l, kyi = ky0 + o;
    if (kxi > hc) continue;                                                  // pad columns
    const float kx = (float)kxi, ky = (kyi < hc + 1) ? (float)kyi : (float)(kyi - nc), kz = (kzi < hc + 1) ? (float)kzi : (float)(kzi - nc);   // :45-56
    if (kxi == 0 && ky <= 0.f && kz <= 0.f) continue;                        // :60
    if (kxi == 0 && ky > 0.f && kz < 0.f) continue;                          // :61
    const float kr = sqrtf(kx * kx + ky * ky + kz * kz);
    if (kr == 0.0f) continue;
    const int k1 = (int)ceilf(kr);
    const float x = P3M_PI_F * kx / fnc, y = P3M_PI_F * ky / fnc, z = P3M_PI_F * kz / fnc;
    const float sx = (x == 0.f) ? 1.f : sinf(x) / x, sy = (y == 0.f) ? 1.f : sinf(y) / y, sz = (z == 0.f) ? 1.f : sinf(z) / z;
    const float kernel = sx * sy * sz, k2 = kernel * kernel;
    const float2 h = lz[idx];
    const float re = (h.x * inv_mean) / n3, im = (h.y * inv_mean) / n3;
    const float pw = re * re + im * im / (k2 * k2);                          // :96
    atomicAdd(&bins[k1 - 1], 1.0);
    atomicAdd(&bins[nb + k1 - 1], (double)pw);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * nb; i += 256) if (bins[i] != 0.0) atomicAdd(&ps_acc[i], bins[i]);
}
int coarse_power_accumulate(p3m_ctx *c, const float *lz, int planes, int ky0, int kx0, int nc, int nchunk, float rho_c_mean, double *d_ps) {
  const int64_t tot = (int64_t)planes * nchunk * nc * 16;
  const int grid = (int)std::min<int64_t>(1024, cdiv(tot, 256));
  hipLaunchKernelGGL(k_coarse_power, dim3(grid), dim3(256), sizeof(double) * 2 * (nc + 2), c->stream, reinterpret_cast<const float2 *>(lz), planes, ky0, kx0, nc, nchunk,
                     1.0f / rho_c_mean, d_ps);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
// the last step of coarse_power.f90 (:112-119): weights and sums -> the (k, Delta^2) rows of <z>ps.dat
void coarse_power_finish(const double *acc, int nc, float box, float *ps) {
  const int nb = nc + 2;
  for (int k = 1; k <= nc; k++) {
    const float w = (float)acc[k - 1], pwr = (float)acc[nb + k - 1];
    ps[2 * (k - 1)] = w; ps[2 * (k - 1) + 1] = pwr;
    if (acc[k - 1] != 0.0) {
      const float km = (float)k - 1.f;
      ps[2 * (k - 1) + 1] = (float)(4.0 * (double)P3M_PI_F * (double)(km * km * km) * acc[nb + k - 1] / acc[k - 1]);
      ps[2 * (k - 1)] = 2.0f * P3M_PI_F * km / box;
    }
  }
}
extern "C" int p3m_hip_coarse_power(p3m_ctx *c, float mass_p, float box, float *ps) {
  if (!c || !ps) return P3M_EINVAL;
  const Geometry &g = c->g;
  if (g.nodes != 1) { p3m_set_error("multi-rank contexts: p3m_hip_group_coarse_power"); return P3M_ECOMM; }
  HIP_TRY(hipSetDevice(c->device));
  const int nc = g.nc, nb = nc + 2;
  double *d_ps = nullptr;
  HIP_TRY(hipMalloc(&d_ps, sizeof(double) * 2 * nb));
  HIP_TRY(hipMemsetAsync(d_ps, 0, sizeof(double) * 2 * nb, c->stream));
  const float nfp = (float)(g.Nn * g.nodes_dim / 2), fnc = (float)nc;
  const float rho_c_mean = nfp * nfp * nfp * mass_p / (fnc * fnc * fnc);   // :24
  int r = coarse_power_accumulate(c, c->slab, nc, 0, 0, nc, g.pxc / 16, rho_c_mean, d_ps);
  std::vector<double> acc(2 * nb);
  if (r == P3M_OK && hipMemcpyAsync(acc.data(), d_ps, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, c->stream) != hipSuccess) r = P3M_EDEVICE;
  if (r == P3M_OK && hipStreamSynchronize(c->stream) != hipSuccess) r = P3M_EDEVICE;
  (void)hipFree(d_ps);
  if (r != P3M_OK) return r;
  coarse_power_finish(acc.data(), nc, box, ps);
  return P3M_OK;
}
