// fft.hip -- bespoke batched 3-D real FFT for gfx950 (no rocFFT/hipFFT).
//
// Replaces FFTW 2.1.5's rfftwnd_f77_one_real_to_complex / _complex_to_real as the reference
// calls them (source_threads/fftw2.f90:19-22): r2c, unnormalised, sign -1, half-complex on the
// fastest axis, array (n+2, n, n) per tile; c2r followed by the division by n^3.
//
// Structure: three 1-D passes, each a Stockham autosort FFT done entirely in LDS by one
// workgroup over a bundle of lines (fft_core.h: mixed radix 2,4,8 and odd primes up to 19 so that
// the tile sizes nf = pt + 48 = 2^4 * {5,7,11,19,35} are covered):
//   x : packed-real trick, one half-length complex FFT per row + split post/pre-processing
//   y,z: complex FFTs over bundles of 16 adjacent x-columns (one 128-byte line per element).
// Between passes the data is kept BUNDLE-MAJOR so that every pass has a fully contiguous side
// (see "memory layouts").  The k-space multiply  F^_c = i K_c rho^
// (particle_mesh_threaded.f90:183-192) is fused into the load of the first inverse pass; the 1/n^3
// and the force-box extraction (:202) into the store of the last one; the inverse passes only
// produce the planes/rows of the force box.  Bound: HBM (no MFMA: nothing here is a dense contraction).
#include "p3m_internal.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include "fft_core.h"

// ------------------------------------------------------------------ memory layouts
// px = n/2+1 rounded up to 16 complex, nchunk = px/16.  Three layouts of one n^3 array, all of
// n*n*px complex:
//   ROWS  [z][y][px]                 real rows of 2*px floats (deposit output, force output)
//   LY    [z][chunk][y][16]          y-bundles: the n line elements of 16 columns are contiguous
//   LZ    [y][chunk][z][16]          z-bundles (rho-hat and the kernels K_c live in LZ)
// LY and LZ share one address formula: bundle (o,chunk), element idx -> (((o*nchunk+chunk)*n+idx)*16).
// forward : x (ROWS -> LY, scratch) ; y (LY -> LZ, back) ; z (LZ in place)
// inverse : z (LZ * iK -> LY, 3 components) ; y (LY in place) ; x (LY -> force box / ROWS)
// Pad columns (kx > n/2) are written as zeros by the x pass and stay zero.
#include "fft_x2.h"   // BXC, P3M_X2_SIZES, X2Cfg: shared with kick_fused.hip
// line lengths n = R1*R2 with a two-register-stage y/z kernel (k_fft_lines2, k_fft_lines3r); anything else runs the LDS Stockham kernels
#define P3M_LINES2_SIZES(X) X(64, 8, 8) X(80, 10, 8) X(96, 12, 8) X(112, 14, 8) X(128, 16, 8) X(160, 16, 10) X(176, 16, 11) X(192, 16, 12) X(208, 16, 13) X(224, 16, 14) \
  X(256, 16, 16) X(304, 19, 16) X(320, 20, 16) X(352, 22, 16) X(384, 24, 16) X(448, 28, 16) X(512, 32, 16) X(560, 28, 20) X(608, 32, 19) \
  X(640, 32, 20) X(704, 32, 22) X(768, 32, 24) X(832, 32, 26) X(896, 32, 28) X(1024, 32, 32)
// Lines longer than P3M_STOCKHAM_MAX no longer fit a 16-column bundle twice into the 160 KiB LDS: they exist with register stages only
#define P3M_STOCKHAM_MAX 608

__device__ __forceinline__ int64_t bundle_off(int64_t b, int n, int nchunk, int o, int chunk) {
  return (((b * n + o) * nchunk + chunk) * (int64_t)n) * BXC;
}
// general form: `planes` bundles-planes per batch element, lines of `line` elements
__device__ __forceinline__ int64_t bundle_off2(int64_t b, int planes, int line, int nchunk, int o, int chunk) {
  return (((b * planes + o) * nchunk + chunk) * (int64_t)line) * BXC;
}

// P3M_FFT_STOCKHAM=1 in the environment forces the LDS Stockham kernels for every size (A/B measurements)
static bool lines2_off(int n) { static const bool off = getenv("P3M_FFT_STOCKHAM") && getenv("P3M_FFT_STOCKHAM")[0] == '1'; return off && n <= P3M_STOCKHAM_MAX; }

// ------------------------------------------------------------------ x pass, forward (r2c): ROWS -> LY
// RB consecutive rows per batch (a power of two, compile-time, so that all LDS index arithmetic is shifts
// and adds); LDS holds element m of row r at m*(RB+1) + r (odd pitch: conflict-free both along rows and
// along the transform).  Persistent workgroups walk a grid-stride list of batches, software-pipelined
// like k_fft_lines: the next batch's rows are in flight (XLU 16-byte loads per lane, whose (row, quad)
// split is loop-invariant) while the butterflies of this one run.  A short last batch is padded with zero rows.
constexpr int XTB = 256, XLU = 6;
template <int RSET, int RB>
__global__ __launch_bounds__(XTB) void k_fft_x_fwd(const float *__restrict__ src, float2 *__restrict__ dst, int n, int px, int rows_total,
                                                   Factors fac, const float2 *__restrict__ tw_g, int rpp) {   // rpp: rows per plane of the LY output (n; fewer for a y-split pencil)
  extern __shared__ float2 lds[];
  constexpr int RBP = RB + 1, LRB = __builtin_ctz(RB);
  const int h = n >> 1, q4 = n >> 2, nchunk = px / BXC;
  float2 *A = lds, *B = A + h * RBP, *tw = B + h * RBP;
  for (int i = threadIdx.x; i < n; i += XTB) tw[i] = tw_g[i];
  const int nbatch = (rows_total + RB - 1) / RB, ne = RB * q4;
  __shared__ int64_t drow[RB];
  int rq[XLU];   // row | quad << 8, or -1
#pragma unroll
  for (int u = 0; u < XLU; u++) { const int e = (int)threadIdx.x + u * XTB; const int r = e / q4; rq[u] = e < ne ? (r | ((e - r * q4) << 8)) : -1; }
  float4 v[XLU];
  auto fetch = [&](int w) {
    const int64_t row0 = (int64_t)w * RB;
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
#pragma unroll
    for (int u = 0; u < XLU; u++) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = rq[u] & 255, m = rq[u] >> 8;
      if (rq[u] >= 0 && r < nrows) v[u] = reinterpret_cast<const float4 *>(src + (row0 + r) * (int64_t)(2 * px))[m];
    }
  };
  int w = blockIdx.x;
  if (w < nbatch) fetch(w);
  for (; w < nbatch; w += gridDim.x) {
    const int64_t row0 = (int64_t)w * RB;
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
#pragma unroll
    for (int u = 0; u < XLU; u++)
      if (rq[u] >= 0) { const int r = rq[u] & 255, m = rq[u] >> 8; float2 *pa = A + (2 * m) * RBP + r; pa[0] = make_float2(v[u].x, v[u].y); pa[RBP] = make_float2(v[u].z, v[u].w); }
    __syncthreads();
    { const int wn = w + gridDim.x; if (wn < nbatch) fetch(wn); }
    if ((int)threadIdx.x < nrows) {   // LY offset of (row, chunk 0, column 0); read after the barriers of fft_lds
      const unsigned row = (unsigned)row0 + threadIdx.x, bz = row / (unsigned)rpp;   // bz = b*n + z; rows_total is an int
      drow[threadIdx.x] = (((int64_t)bz * nchunk) * rpp + (row - bz * rpp)) * BXC;
    }
    const float2 *Z = fft_lds<false, RSET, RB>(A, B, h, RB, RBP, 1, fac, tw, 2);
    // X[k] = E + W_n^k O,  E = (Z[k]+conj Z[h-k])/2,  O = (Z[k]-conj Z[h-k])/(2i); lanes run over the 16
    // columns of a chunk, then over rows: consecutive rows of one chunk are contiguous in LY
    const int tot = RB * px;
    for (int e = threadIdx.x; e < tot; e += XTB) {
      const int l = e & (BXC - 1), r = (e >> 4) & (RB - 1), chunk = e >> (4 + LRB);
      const int k = chunk * BXC + l;
      float2 X = make_float2(0.f, 0.f);
      if (k <= h) {
        const float2 zk = Z[(k == h ? 0 : k) * RBP + r];
        const float2 zc = cconj(Z[(k == 0 ? 0 : h - k) * RBP + r]);
        const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
        const float2 O = make_float2(0.5f * (zk.y - zc.y), -0.5f * (zk.x - zc.x));
        const float2 wk = (k == h) ? make_float2(-1.f, 0.f) : tw[k];
        X = cadd(E, cmul(O, wk));
      }
      if (r < nrows) dst[drow[r] + ((int64_t)chunk * rpp) * BXC + l] = X;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ x pass, inverse (c2r) + /n^3: LY -> ROWS or force box
// mode 0: all rows, real rows written to out (ROWS layout).  mode 1: rows enumerate (b, kk, jj) over the
// force box force_f(c, nb-1:nf-nb+1,...) (particle_mesh_threaded.f90:202), b = comp*ntile + tile; only
// the box columns are written, to box + comp*box_comp_stride + tile*fb^3; lo = nb-2 (first box cell).
// Same persistent, pipelined structure as the forward pass; the per-row source/destination tables are
// double-buffered because the prefetch needs the next batch's.
template <int RSET, int RB>
__global__ __launch_bounds__(XTB) void k_fft_x_inv(const float2 *__restrict__ src, float *__restrict__ out, int n, int px, int rows_total,
                                                   Factors fac, const float2 *__restrict__ tw_g, float inv_scale, int mode,
                                                   float *__restrict__ box, int fb, int lo, int ntile, int64_t box_comp_stride, int rpp) {   // rpp: rows per plane of the LY input (mode 0)
  extern __shared__ float2 lds[];
  constexpr int RBP = RB + 1, LRB = __builtin_ctz(RB);
  const int h = n >> 1, nchunk = px / BXC, fbp = (fb + 3) & ~3;
  const int lrows = mode == 1 ? n : rpp;   // line length of the LY bundles this pass reads
  float2 *A = lds, *B = A + (h + 1) * RBP, *tw = B + (h + 1) * RBP;
  __shared__ int64_t src_row[2][RB], dst_off[2][RB];   // element offset of (row, chunk 0, column 0) in LY; chunks are n*16 apart
  for (int i = threadIdx.x; i < n; i += XTB) tw[i] = tw_g[i];
  const int nbatch = (rows_total + RB - 1) / RB;
  const int ncol = (h + BXC) & ~(BXC - 1);   // columns 0..h rounded up to whole chunks
  const int ne = RB * (ncol >> 1);
  auto tables = [&](int w, int buf) {
    const int64_t srow = (int64_t)w * RB + threadIdx.x;
    if ((int)threadIdx.x < RB && srow < rows_total) {
      if (mode == 1) {
        // rows_total is an int: 32-bit unsigned divisions
        const unsigned s32 = (unsigned)srow, t2 = s32 / (unsigned)fb, bb = t2 / (unsigned)fb;
        const int jj = (int)(s32 - t2 * fb), kk = (int)(t2 - bb * fb);
        const int comp = (int)(bb / (unsigned)ntile), tl = (int)(bb - comp * ntile);
        const int64_t b = bb;
        src_row[buf][threadIdx.x] = (((b * n + (kk + lo)) * nchunk) * n + (jj + lo)) * BXC;
        dst_off[buf][threadIdx.x] = comp * box_comp_stride + (((int64_t)tl * fb + kk) * fb + jj) * fbp;
      } else {
        const unsigned bz = (unsigned)srow / (unsigned)rpp;
        src_row[buf][threadIdx.x] = (((int64_t)bz * nchunk) * rpp + ((unsigned)srow - bz * rpp)) * BXC;
        dst_off[buf][threadIdx.x] = srow * (int64_t)(2 * px);
      }
    }
  };
  const fdiv_t dPx = mk_fdiv(px), dNq = mk_fdiv(fbp >> 2 ? fbp >> 2 : 1);
  const float rscale = 1.0f / inv_scale;   // the division by n^3 (fftw2.f90:22) as a multiplication: one ulp, inside the FFT's own rounding
  float4 v[XLU];
  // load e = tid + u*XTB: l4 = e & 7, row = (e >> 3) & (RB-1), chunk = e >> (3 + LRB)
  auto fetch = [&](int w, int buf) {
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - (int64_t)w * RB);
#pragma unroll
    for (int u = 0; u < XLU; u++) {
      const int e = (int)threadIdx.x + u * XTB;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int l4 = e & 7, r = (e >> 3) & (RB - 1), chunk = e >> (3 + LRB);
      if (e < ne && r < nrows) v[u] = reinterpret_cast<const float4 *>(src + src_row[buf][r] + (int64_t)chunk * lrows * BXC)[l4];
    }
  };
  int w = blockIdx.x, buf = 0;
  if (w < nbatch) tables(w, 0);
  __syncthreads();
  if (w < nbatch) fetch(w, 0);
  for (; w < nbatch; w += gridDim.x, buf ^= 1) {
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - (int64_t)w * RB);
    const int wn = w + gridDim.x;
    // the gathered rows (columns 0..h) go to B[k*RBP + r]
#pragma unroll
    for (int u = 0; u < XLU; u++) {
      const int e = (int)threadIdx.x + u * XTB;
      if (e < ne) {
        const int l4 = e & 7, r = (e >> 3) & (RB - 1), chunk = e >> (3 + LRB);
        const int k = chunk * BXC + 2 * l4;
        float2 *pb = B + k * RBP + r;
        if (k <= h) pb[0] = make_float2(v[u].x, v[u].y);
        if (k + 1 <= h) pb[RBP] = make_float2(v[u].z, v[u].w);
      }
    }
    if (wn < nbatch) tables(wn, buf ^ 1);
    __syncthreads();
    if (wn < nbatch) fetch(wn, buf ^ 1);
    // Z'[m] = (X[m] + conj X[h-m]) + i (X[m] - conj X[h-m]) W_n^{-m}; conj() on the way in: the
    // forward machinery then yields conj(IFFT), undone on the way out.
    for (int e = threadIdx.x; e < RB * h; e += XTB) {
      const int m = e >> LRB, r = e & (RB - 1);
      const float2 xk = B[m * RBP + r], xc = cconj(B[(h - m) * RBP + r]);
      const float2 e2 = cadd(xk, xc), d = csub(xk, xc);
      const float2 o = cmul(d, cconj(tw[m]));
      A[m * RBP + r] = make_float2(e2.x - o.y, -(e2.y + o.x));  // conj(e + i o)
    }
    __syncthreads();
    const float2 *Z = fft_lds<false, RSET, RB>(A, B, h, RB, RBP, 1, fac, tw, 2);
    if (mode == 0) {
      for (int e = threadIdx.x; e < nrows * px; e += XTB) {
        const int r = fdiv(e, dPx), m = e - r * px;
        float2 z = make_float2(0.f, 0.f);
        if (m < h) { z = Z[m * RBP + r]; z = make_float2(z.x * rscale, -z.y * rscale); }
        reinterpret_cast<float2 *>(out + dst_off[buf][r])[m] = z;
      }
    } else {
      // 4 consecutive box cells = 2 complex values per lane, one 16-byte store (lo is even, fbp % 4 == 0)
      const int nq = fbp >> 2;
      for (int e = threadIdx.x; e < nrows * nq; e += XTB) {
        const int r = fdiv(e, dNq), q = e - r * nq;
        const int cx = (4 * q + lo) >> 1;
        const float2 z0 = Z[cx * RBP + r], z1 = Z[(cx + 1) * RBP + r];
        float4 o = make_float4(z0.x * rscale, -z0.y * rscale, z1.x * rscale, -z1.y * rscale);
        if (4 * q + 0 >= fb) o.x = 0.f;
        if (4 * q + 1 >= fb) o.y = 0.f;
        if (4 * q + 2 >= fb) o.z = 0.f;
        if (4 * q + 3 >= fb) o.w = 0.f;
        *reinterpret_cast<float4 *>(box + dst_off[buf][r] + 4 * q) = o;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ x pass, forward: two register stages (h = n/2 = R1*R2)
// Thread (q, row) loads the packed-real elements Z[R2*a + q] of its row straight into registers (8-byte loads at
// compile-time offsets from one per-row address), transforms (dft<R1>), twiddles and writes the exchange buffer; thread
// (k1, row) finishes (dft<R2>) and leaves Z^[k1 + R1*k2] in a second LDS array; the split X[k] = E + W^k O, X[h-k] =
// conj(E - W^k O) then runs with lanes along the 16 columns of a chunk and stores both halves to LY.  Row geometry as
// in k_fft_x_inv2; the next batch's rows are in flight during stage 2 and the split.
// CUBES: the rows are not an array of their own -- row (rank r, plane zl, y) of the coarse slab decomposition is read where its cells
// lie, in plane (r % nd^2)*s + zl of the ncn^3 cubes of the ranks layer(r) + (y/ncn)*nd + i, i < nd (pack_slab, fftw3ds.f90:24-52, with
// every logical rank in this process: p3m_group::direct), so the redistribution costs no pass over memory of its own
// U8 (round 6, the NGP density of whole steps: RowDep::rho8): a row is 2*px BYTES, the cells' counts; a count becomes the density the
// reference's deposit leaves -- mass_p added count times (particle_mesh_threaded.f90:148) -- through a table of the first 256 such sums
// (built once per workgroup; beyond 255 the additions are made in place).  A quarter of the bytes of the float rows, and the rows' writer
// (k_row_sort) is rid of the partial sums
template <int R1, int R2, bool CUBES = false, bool U8 = false>
__global__ __launch_bounds__(256) void k_fft_x_fwd2(const float *__restrict__ src, float2 *__restrict__ dst_, int n, int px, int rows_total,
                                                    const float2 *__restrict__ tw_g, int rpp, RowGeom cq, float mass_p = 0.f) {   // rpp: rows per plane of the LY output (n; fewer for a y-split pencil)
  using C = X2Cfg<R1, R2>;
  __shared__ float ctab[U8 ? 256 : 1];
  if constexpr (U8) {
    float r = 0.f;
    for (int i = 0; i < (int)threadIdx.x && i < 255; i++) r = r + mass_p;
    if (threadIdx.x < 256) ctab[threadIdx.x] = r;
  }
  constexpr int h = C::h, Q = C::Q, RB = C::RB, R2P = C::R2P, P = C::P;
  extern __shared__ float2 lds[];
  c32 *Y = reinterpret_cast<c32 *>(lds), *X = Y + RB * P, *tw = X + RB * R1 * R2P;
  __shared__ int64_t drow[2][RB];
  for (int i = threadIdx.x; i < h; i += C::TB) tw[i] = reinterpret_cast<const c32 *>(tw_g)[i];
  c32 *dst = reinterpret_cast<c32 *>(dst_);
  const int nchunk = px / BXC;
  const int lane = threadIdx.x & 63, rw = lane / Q, q = lane - rw * Q;
  const int r = (threadIdx.x >> 6) * C::RPW + rw;
  const bool act = rw < C::RPW, s1 = act && q < R2, s2 = act && q < R1;
  const int nbatch = (rows_total + RB - 1) / RB;
  const int64_t cstride = (int64_t)rpp * BXC;
  c32 twq[R1];   // W_h^{q*k1}
#pragma unroll
  for (int k1 = 0; k1 < R1; k1++) twq[k1] = reinterpret_cast<const c32 *>(tw_g)[s1 ? 2 * q * k1 : 0];
  c32 v[R1];
  unsigned short cr[U8 ? R1 : 1];   // U8: the next batch's counts (two cells per element), turned into v at the top of the trip
  auto count_mass = [&](unsigned cn) { float m = ctab[min(cn, 255u)]; for (unsigned i = 255u; i < cn; i++) m = m + mass_p; return m; };
  auto fetch = [&](int w) {
    const int64_t row = (int64_t)w * RB + r;
    if constexpr (U8) {
#pragma unroll
      for (int a = 0; a < R1; a++) cr[a] = 0;
      if (s1 && row < rows_total) {
        const unsigned short *ps = reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(src) + row * (int64_t)(2 * px)) + q;
#pragma unroll
        for (int a = 0; a < R1; a++) cr[a] = ps[R2 * a];
      }
      return;
    }
#pragma unroll
    for (int a = 0; a < R1; a++) v[a] = (c32){0.f, 0.f};
    if (s1 && row < rows_total) {
      if constexpr (CUBES) {
        const fdiv_t d_rpp{cq.m_rpp, cq.rpp}, d_s{cq.m_s, cq.s}, d_ncn{cq.m_ncn, cq.ncn};
        const unsigned plane = fdiv((int)row, d_rpp), y = (unsigned)row - plane * cq.rpp, rank = fdiv((int)plane, d_s), zl = plane - rank * cq.s;
        const unsigned j = fdiv((int)y, d_ncn), yy = y - j * cq.ncn, nd2 = cq.nd * cq.nd, layer = rank / nd2, qz = rank - layer * nd2;
        const int64_t n3 = (int64_t)cq.ncn * cq.ncn * cq.ncn;
        const float *pc = src + (int64_t)(layer * nd2 + j * cq.nd) * n3 + ((int64_t)(qz * cq.s + zl) * cq.ncn + yy) * cq.ncn;
#pragma unroll
        for (int a = 0; a < R1; a++) {
          const int x = 2 * (R2 * a + q), i = fdiv(x, d_ncn);            // ncn is even: a pair of cells never straddles two cubes
          v[a] = *reinterpret_cast<const c32 *>(pc + i * (n3 - cq.ncn) + x);
        }
      } else {
        const c32 *ps = reinterpret_cast<const c32 *>(src + row * (int64_t)(2 * px)) + q;
#pragma unroll
        for (int a = 0; a < R1; a++) v[a] = ps[R2 * a];
      }
    }
  };
  // split item e = tid + u*TB: l = e & 15, row = (e >> 4) % RB, chunk = (e >> 4) / RB, k = 16*chunk + l <= h/2
  constexpr int NCS = (h / 2) / BXC + 1, NSP = (RB * NCS * BXC + C::TB - 1) / C::TB;
  int w = blockIdx.x, buf = 0;
  if (w < nbatch) fetch(w);
  __syncthreads();
  for (; w < nbatch; w += gridDim.x, buf ^= 1) {
    const int64_t row0 = (int64_t)w * RB;
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
    if ((int)threadIdx.x < nrows) {   // LY offset of (row, chunk 0, column 0)
      const unsigned row = (unsigned)row0 + threadIdx.x, bz = row / (unsigned)rpp;   // bz = b*n + z; rows_total is an int
      drow[buf][threadIdx.x] = (((int64_t)bz * nchunk) * rpp + (row - bz * rpp)) * BXC;
    }
    if (s1) {
      if constexpr (U8) {
#pragma unroll
        for (int a = 0; a < R1; a++) v[a] = (c32){count_mass(cr[a] & 255u), count_mass(cr[a] >> 8)};
      }
      dft<R1>(v);
      c32 *pxw = X + (r * R1) * R2P + q;
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) pxw[k1 * R2P] = k1 ? vmulr(v[k1], twq[k1]) : v[0];
    }
    __syncthreads();
    fetch(w + gridDim.x);
    if (s2) {
      c32 u[R2];
      const c32 *pxr = X + (r * R1 + q) * R2P;
#pragma unroll
      for (int b = 0; b < R2; b++) u[b] = pxr[b];
      dft<R2>(u);
      c32 *py = Y + r * P + q;
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) py[R1 * k2] = u[k2];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NSP; u++) {
      const int e = (int)threadIdx.x + u * C::TB, l = e & (BXC - 1), t = e >> 4, ch = t / RB, rr = t - ch * RB;
      const int k = ch * BXC + l;
      if (ch < NCS && k <= h / 2 && rr < nrows) {
        const c32 zk = Y[rr * P + k], zm = Y[rr * P + (k == 0 ? 0 : h - k)];
        const c32 E = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)}, O = {0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)};
        const c32 wk = tw[k];
        const c32 T = {O.x * wk.x - O.y * wk.y, O.x * wk.y + O.y * wk.x};
        c32 *pd = dst + drow[buf][rr];
        const int km = h - k;   // k = 0: X[h] = E - T (real); k = h/2: the same element twice
        pd[ch * cstride + l] = E + T;
        pd[(km >> 4) * cstride + (km & 15)] = (c32){E.x - T.x, -(E.y - T.y)};
      }
    }
    // pad columns h+1 .. px-1 hold zeros
    for (int e = threadIdx.x; e < nrows * (px - h - 1); e += C::TB) {
      const int rr = e / (px - h - 1), k = h + 1 + (e - rr * (px - h - 1));
      dst[drow[buf][rr] + (k >> 4) * cstride + (k & 15)] = (c32){0.f, 0.f};
    }
  }
}

// ------------------------------------------------------------------ x pass, inverse, force box only: two register stages (h = n/2 = R1*R2)
// Same idea as k_fft_lines2 below, along rows.  The rows of a batch are gathered from LY with 16-byte loads into LDS
// (element m of row r at r*P + m, P = h+1) -- the only place a per-element global address is formed; everything after
// that addresses LDS at a per-thread base plus compile-time offsets.  Thread (q, row) reads X[R2*a + q] and
// X[h - R2*a - q], forms the packed c2r input, transforms (dft<R1>), twiddles (factors held in registers: q is fixed for
// the life of the workgroup) and writes the exchange buffer once; thread (k1, row) finishes (dft<R2>) and stores the box
// columns of real elements 2j, 2j+1, j = k1 + R1*k2, as 8-byte pairs at compile-time offsets.  A wave holds RPW whole
// rows (Q = max(R1,R2) lanes each); exchange element (row, k1, b) sits at (row*R1 + k1)*R2P + b, R2P odd.  The next
// batch's gather is in flight during both stages.
// BOX = false: every row of `rows_total` LY rows (rpp rows per plane) as a real row of 2*px floats at box + row*2*px (the
// coarse mesh, the probes): fb = n, lo = 0, the pad floats n .. 2*px-1 are written as zeros.
template <int R1, int R2, bool BOX = true>
__global__ __launch_bounds__(256) void k_fft_x_inv2(const float2 *__restrict__ src, int n, int px, int rows_total, const float2 *__restrict__ tw_g,
                                                    float inv_scale, float *__restrict__ box, int fb, int lo, int ntile, int64_t box_comp_stride, int rpp) {
  using C = X2Cfg<R1, R2>;
  constexpr int h = C::h, Q = C::Q, RB = C::RB, R2P = C::R2P, P = C::P, NLD = C::NLD;
  extern __shared__ float2 lds[];
  c32 *B = reinterpret_cast<c32 *>(lds), *X = B + RB * P, *tw = X + RB * R1 * R2P;
  // THREE sets of the per-row tables: the set of batch i is read until the last store of trip i (stage 2), and trip i + 1 -- which a
  // wavefront enters without a barrier -- writes the set of batch i + 2: with two sets that was the one a slower wavefront of the
  // workgroup was still reading, and its four rows went to the rows of batch i + 2 (overwritten there later) while their own kept what
  // they held before (round 4: one step in ten of the full-size 8-rank run, when the fine-mesh passes on the other stream skewed
  // the wavefronts of the coarse transform; tests/determinism_check.py)
  __shared__ int64_t src_row[3][RB], dst_off[3][RB];
  for (int i = threadIdx.x; i < h; i += C::TB) tw[i] = reinterpret_cast<const c32 *>(tw_g)[i];
  const int nchunk = px / BXC, fbp = (fb + 3) & ~3;
  const int lane = threadIdx.x & 63, rw = lane / Q, q = lane - rw * Q;
  const int r = (threadIdx.x >> 6) * C::RPW + rw;
  const bool act = rw < C::RPW, s1 = act && q < R2, s2 = act && q < R1;
  const int nbatch = (rows_total + RB - 1) / RB;
  const float rscale = 1.0f / inv_scale;
  const int64_t cstride = (int64_t)(BOX ? n : rpp) * BXC;
  c32 twq[R1];   // W_h^{q*k1}
#pragma unroll
  for (int k1 = 0; k1 < R1; k1++) twq[k1] = reinterpret_cast<const c32 *>(tw_g)[s1 ? 2 * q * k1 : 0];
  auto tables = [&](int w, int buf) {
    const int64_t srow = (int64_t)w * RB + threadIdx.x;
    if ((int)threadIdx.x < RB && srow < rows_total) {
      if (!BOX) {
        const unsigned bz = (unsigned)srow / (unsigned)rpp;
        src_row[buf][threadIdx.x] = (((int64_t)bz * nchunk) * rpp + ((unsigned)srow - bz * rpp)) * BXC;
        dst_off[buf][threadIdx.x] = srow * (int64_t)(2 * px);
        return;
      }
      const unsigned s32 = (unsigned)srow, t2 = s32 / (unsigned)fb, bb = t2 / (unsigned)fb;
      const int jj = (int)(s32 - t2 * fb), kk = (int)(t2 - bb * fb);
      const int comp = (int)(bb / (unsigned)ntile), tl = (int)(bb - comp * ntile);
      src_row[buf][threadIdx.x] = ((((int64_t)bb * n + (kk + lo)) * nchunk) * n + (jj + lo)) * BXC;
      dst_off[buf][threadIdx.x] = comp * box_comp_stride + (((int64_t)tl * fb + kk) * fb + jj) * fbp;
    }
  };
  // gather item e = tid + u*TB: l4 = e & 7, row = (e >> 3) % RB, chunk = (e >> 3) / RB
  int grc[NLD];   // row | chunk << 8 | l4 << 16, or -1
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int e = (int)threadIdx.x + u * C::TB, t = e >> 3, ch = t / RB;
    grc[u] = ch < C::NCH ? ((t - ch * RB) | (ch << 8) | ((e & 7) << 16)) : -1;
  }
  float4 g4[NLD];
  auto fetch = [&](int w, int buf) {
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - (int64_t)w * RB);
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      g4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int rr = grc[u] & 255, ch = (grc[u] >> 8) & 255, l4 = grc[u] >> 16;
      if (grc[u] >= 0 && rr < nrows) g4[u] = reinterpret_cast<const float4 *>(src + src_row[buf][rr] + ch * cstride)[l4];
    }
  };
  int w = blockIdx.x, buf = 0;
  if (w < nbatch) tables(w, 0);
  __syncthreads();
  if (w < nbatch) fetch(w, 0);
  for (; w < nbatch; w += gridDim.x, buf = buf == 2 ? 0 : buf + 1) {
    const int nxt = buf == 2 ? 0 : buf + 1;
    const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - (int64_t)w * RB);
    const bool rowok = r < nrows;
    const int wn = w + gridDim.x;
#pragma unroll
    for (int u = 0; u < NLD; u++)
      if (grc[u] >= 0) {
        const int rr = grc[u] & 255, k = ((grc[u] >> 8) & 255) * BXC + 2 * (grc[u] >> 16);
        c32 *pb = B + rr * P + k;
        if (k <= h) pb[0] = (c32){g4[u].x, g4[u].y};
        if (k + 1 <= h) pb[1] = (c32){g4[u].z, g4[u].w};
      }
    if (wn < nbatch) tables(wn, nxt);
    __syncthreads();
    if (wn < nbatch) fetch(wn, nxt);
    if (s1 && rowok) {
      const c32 *pk = B + r * P + q, *pm = B + r * P + (h - R2 * (R1 - 1)) - q, *pt = tw + q;
      c32 v[R1];
#pragma unroll
      for (int a = 0; a < R1; a++) {
        v[a] = c2r_pre(pk[R2 * a], pm[R2 * (R1 - 1 - a)], pt[R2 * a]);   // X[m], X[h-m], exp(-2 pi i m / n), m = R2*a + q -> conj(e + i o): the forward machinery then yields conj(IFFT)
      }
      dft<R1>(v);
      c32 *pxw = X + (r * R1) * R2P + q;
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) pxw[k1 * R2P] = k1 ? vmulr(v[k1], twq[k1]) : v[0];
    }
    __syncthreads();
    if (s2 && rowok) {
      c32 u[R2];
      const c32 *pxr = X + (r * R1 + q) * R2P;
#pragma unroll
      for (int b = 0; b < R2; b++) u[b] = pxr[b];
      dft<R2>(u);
      const int x0 = 2 * q - lo;   // box column of real element 2j for k2 = 0; lo is even
      float *pd = box + dst_off[buf][r] + x0;
      if (!BOX) {
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) *reinterpret_cast<c32 *>(pd + 2 * R1 * k2) = (c32){u[k2].x * rscale, -u[k2].y * rscale};
      } else if (fb == fbp) {
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++)
          if ((unsigned)(x0 + 2 * R1 * k2) < (unsigned)fb) __builtin_nontemporal_store((c32){u[k2].x * rscale, -u[k2].y * rscale}, reinterpret_cast<c32 *>(pd + 2 * R1 * k2));
      } else {
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) {
          const int x = x0 + 2 * R1 * k2;
          if ((unsigned)x < (unsigned)fbp) {
            float2 o2 = make_float2(u[k2].x * rscale, -u[k2].y * rscale);
            if (x >= fb) o2.x = 0.f;
            if (x + 1 >= fb) o2.y = 0.f;
            *reinterpret_cast<float2 *>(pd + 2 * R1 * k2) = o2;
          }
        }
      }
    }
    if (!BOX) {   // pad floats n .. 2*px-1 of every row
      const int npad = px - h;
      for (int e = threadIdx.x; e < nrows * npad; e += C::TB) {
        const int rr = e / npad;
        *reinterpret_cast<c32 *>(box + dst_off[buf][rr] + 2 * (h + (e - rr * npad))) = (c32){0.f, 0.f};
      }
    }
  }
}

// ------------------------------------------------------------------ x pass, inverse, coarse slab decomposition with every rank in this process
// k_fft_x_inv2<.., false> and the copy of its rows into the owners' force arrays (unpack_slab, fftw3ds.f90:69-99) in one pass: row
// (rank r, plane zl, y) of component c is stored where its cells belong -- plane 1 + (r % nd^2)*s + zl, row 1 + y % ncn of component c
// of the (ncn+2)^3 force arrays of the ranks layer(r) + (y/ncn)*nd + i -- and max |F| over the interior (coarse_max_dt.f90:24-31)
// is formed on the way: a workgroup takes the three components of a batch of rows in three consecutive trips and keeps the
// squares in registers.  ncn % RB == 0: the rows of a batch share rank, plane and y/ncn.  (The stores are 4-byte: the force
// arrays' rows start one cell in.)
template <int R1, int R2>
__global__ __launch_bounds__(256) void k_fft_x_inv2c(const float2 *__restrict__ src, int n, int px, int rows1, const float2 *__restrict__ tw_g, float inv_scale,
                                                     float *__restrict__ fc, RowGeom cq, RankPtrs red) {
  using C = X2Cfg<R1, R2>;
  constexpr int h = C::h, Q = C::Q, RB = C::RB, R2P = C::R2P, P = C::P, NLD = C::NLD;
  extern __shared__ float2 lds[];
  c32 *B = reinterpret_cast<c32 *>(lds), *X = B + RB * P, *tw = X + RB * R1 * R2P;
  __shared__ int64_t src_row[3][RB], dst_off[3][RB];   // three sets: see k_fft_x_inv2
  __shared__ int owner0[3];
  for (int i = threadIdx.x; i < h; i += C::TB) tw[i] = reinterpret_cast<const c32 *>(tw_g)[i];
  const int nchunk = px / BXC, rpp = cq.rpp;
  const int lane = threadIdx.x & 63, rw = lane / Q, q = lane - rw * Q;
  const int r = (threadIdx.x >> 6) * C::RPW + rw;
  const bool act = rw < C::RPW, s1 = act && q < R2, s2 = act && q < R1;
  const int nb1 = (rows1 + RB - 1) / RB;               // row batches of one component
  const float rscale = 1.0f / inv_scale;
  const int64_t cstride = (int64_t)rpp * BXC;
  const int m = cq.ncn + 2; const int64_t fcs = (int64_t)m * m * m;
  const fdiv_t d_rpp{cq.m_rpp, cq.rpp}, d_s{cq.m_s, cq.s}, d_ncn{cq.m_ncn, cq.ncn};
  c32 twq[R1];   // W_h^{q*k1}
#pragma unroll
  for (int k1 = 0; k1 < R1; k1++) twq[k1] = reinterpret_cast<const c32 *>(tw_g)[s1 ? 2 * q * k1 : 0];
  // trip t of this workgroup: row batch blockIdx.x + (t/3)*gridDim.x, component t % 3
  auto tables = [&](int wb, int comp, int buf) {
    const int64_t row1 = (int64_t)wb * RB + threadIdx.x;
    if ((int)threadIdx.x < RB && row1 < rows1) {
      const unsigned srow = (unsigned)(comp * (int64_t)rows1 + row1), bz = srow / (unsigned)rpp;
      src_row[buf][threadIdx.x] = (((int64_t)bz * nchunk) * rpp + (srow - bz * rpp)) * BXC;
      const unsigned plane = fdiv((int)row1, d_rpp), y = (unsigned)row1 - plane * rpp, rank = fdiv((int)plane, d_s), zl = plane - rank * cq.s;
      const unsigned j = fdiv((int)y, d_ncn), yy = y - j * cq.ncn, nd2 = cq.nd * cq.nd, layer = rank / nd2, qz = rank - layer * nd2;
      const int own = layer * nd2 + j * cq.nd;
      dst_off[buf][threadIdx.x] = ((int64_t)own * 3 + comp) * fcs + ((int64_t)(1 + qz * cq.s + zl) * m + (1 + yy)) * m + 1;
      if (threadIdx.x == 0) owner0[buf] = own;
    }
  };
  int grc[NLD];   // row | chunk << 8 | l4 << 16, or -1
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int e = (int)threadIdx.x + u * C::TB, t = e >> 3, ch = t / RB;
    grc[u] = ch < C::NCH ? ((t - ch * RB) | (ch << 8) | ((e & 7) << 16)) : -1;
  }
  float4 g4[NLD];
  auto fetch = [&](int wb, int buf) {
    const int nrows = (int)min((int64_t)RB, (int64_t)rows1 - (int64_t)wb * RB);
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      g4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int rr = grc[u] & 255, ch = (grc[u] >> 8) & 255, l4 = grc[u] >> 16;
      if (grc[u] >= 0 && rr < nrows) g4[u] = reinterpret_cast<const float4 *>(src + src_row[buf][rr] + ch * cstride)[l4];
    }
  };
  float sq[R2][2];
  int wb = blockIdx.x, comp = 0, buf = 0;
  if (wb < nb1) tables(wb, 0, 0);
  __syncthreads();
  if (wb < nb1) fetch(wb, 0);
  while (wb < nb1) {
    const int nxt = buf == 2 ? 0 : buf + 1;
    const int nrows = (int)min((int64_t)RB, (int64_t)rows1 - (int64_t)wb * RB);
    const bool rowok = r < nrows;
    const int ncomp = comp == 2 ? 0 : comp + 1, nwb = comp == 2 ? wb + (int)gridDim.x : wb;
#pragma unroll
    for (int u = 0; u < NLD; u++)
      if (grc[u] >= 0) {
        const int rr = grc[u] & 255, k = ((grc[u] >> 8) & 255) * BXC + 2 * (grc[u] >> 16);
        c32 *pb = B + rr * P + k;
        if (k <= h) pb[0] = (c32){g4[u].x, g4[u].y};
        if (k + 1 <= h) pb[1] = (c32){g4[u].z, g4[u].w};
      }
    if (nwb < nb1) tables(nwb, ncomp, nxt);
    __syncthreads();
    if (nwb < nb1) fetch(nwb, nxt);
    if (s1 && rowok) {
      const c32 *pk = B + r * P + q, *pm = B + r * P + (h - R2 * (R1 - 1)) - q, *pt = tw + q;
      c32 v[R1];
#pragma unroll
      for (int a = 0; a < R1; a++) {
        v[a] = c2r_pre(pk[R2 * a], pm[R2 * (R1 - 1 - a)], pt[R2 * a]);
      }
      dft<R1>(v);
      c32 *pxw = X + (r * R1) * R2P + q;
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) pxw[k1 * R2P] = k1 ? vmulr(v[k1], twq[k1]) : v[0];
    }
    __syncthreads();
    if (s2 && rowok) {
      c32 u[R2];
      const c32 *pxr = X + (r * R1 + q) * R2P;
#pragma unroll
      for (int b = 0; b < R2; b++) u[b] = pxr[b];
      dft<R2>(u);
      float *pd = fc + dst_off[buf][r];
      // 4-byte stores: the force arrays' rows start one cell in.  (Measured alternatives, n = 1024: the pair as one 8-byte store at
      // 4-byte alignment 9.6 ms, the row through LDS to stores of 64 consecutive cells 8.9 ms, this 8.4 ms; at one wavefront per
      // SIMD -- 400 registers -- the pass is bound by its instruction stream, not by how its stores coalesce)
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) {
        const int x = 2 * (q + R1 * k2), i = fdiv(x, d_ncn);          // cell x of the row belongs to owner i = x / ncn (x < nc: h = R1*R2)
        const float a = u[k2].x * rscale, b = -u[k2].y * rscale;
        float *pp = pd + (int64_t)i * (3 * fcs - cq.ncn) + x;
        pp[0] = a; pp[1] = b;
        if (comp == 0) { sq[k2][0] = a * a; sq[k2][1] = b * b; } else { sq[k2][0] += a * a; sq[k2][1] += b * b; }
      }
    }
    if (comp == 2) {   // max |F| of the batch's cells, per owner (uniform branch)
      for (int i = 0; i < cq.nd; i++) {
        float mx = 0.f;
        if (s2 && rowok) {
#pragma unroll
          for (int k2 = 0; k2 < R2; k2++)
            if (fdiv(2 * (q + R1 * k2), d_ncn) == i) mx = fmaxf(mx, fmaxf(sq[k2][0], sq[k2][1]));
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0 && mx > 0.f) p3m_atomic_max_nonneg(red.p[owner0[buf] + i] + p3m_slot() * 16, sqrtf(mx));
      }
    }
    wb = nwb; comp = ncomp; buf = nxt;
  }
}

// ------------------------------------------------------------------ y / z passes on bundles
// One workgroup transforms one bundle: n line elements x 16 columns, contiguous in the source
// (LY for a y pass, LZ for a z pass).  The result goes either to the same bundle (TR = false:
// z forward, y inverse) or, element idx of bundle (o,chunk), to element o of bundle (idx,chunk) of
// the other bundle layout (TR = true: y forward LY->LZ, z inverse LZ->LY).
// NC = 0: plain.  NC = 1: fused k-space multiply (particle_mesh_threaded.f90:183-192; all three components: k_fft_lines3); the kernels
// K_c are stored in LZ like rho-hat; component c goes to dst + c*dst_comp_stride.
// Pruning: bundles o in [olo, olo+ocount) only; line elements [slo, slo+scount) are stored.
// Distributed (slab) transforms use src_planes / dst_planes / dst_line != n: a rank then holds only
// its nc_slab planes, and the transposing store writes straight into the all-to-all send layout.
struct LinesArgs {
  float2 *dst; const float2 *src; const float *kern;
  int64_t kern_comp_stride, dst_comp_stride;
  int64_t kern_batch_stride;   // floats between the kernel tables of consecutive batch entries: 0 where all tiles share kern_f, one slab where every logical rank of a distributed transform has its own ky range
  int n, nchunk, olo, ocount, slo, scount, nbundles;
  int src_planes, dst_planes, dst_line;
  // seg > 0 (register-stage kernels, !TR): the source line arrives in n/seg segments of `seg` elements, one from each peer of an
  // all-to-all -- element idx of bundle (o, chunk) of batch entry b sits at
  //   b*src_planes*nchunk*n*16 + (idx/seg)*(src_planes*nchunk*seg*16) + ((o*nchunk + chunk)*seg + idx%seg)*16
  // (the receive buffer [peer][plane][chunk][seg][16] as it is): the pass reads it in place of a separate permutation kernel
  int seg; unsigned seg_magic;
  // direct_s > 0 (register-stage kernels, TR): every logical rank of the transpose lives in this process and is a batch entry --
  // line element idx of batch entry b belongs to peer idx / direct_s, and is stored straight into THAT peer's receive block
  // (where the all-to-all would have copied it): the normal address plus (peer - b) * direct_delta elements
  int direct_s; unsigned direct_magic; int64_t direct_delta;
  // kmirror (k_fft_lines3r): K(c, n - z) = +-K(c, z) holds EXACTLY in `kern` (odd for component 2, the line's own axis; even for 0 and 1) --
  // the upper half of a line is read from its mirror elements, which the thread that owns those reads at the same time: the table's
  // HBM traffic halves (0.53 of 3.7 GB per pass at n = 560), the values are bit for bit the ones a direct read returns
  int kmirror;
};
// Each workgroup walks a grid-stride list of work items and is software-pipelined: the next
// item's global loads are issued into registers (LUX 16-byte loads per lane) before the butterflies
// of the current one run, so HBM latency hides under the LDS stages.  (Three components: k_fft_lines3.)
template <bool INV, bool TR, int NC, int RSET, int TB, int LUX>
__global__ __launch_bounds__(TB) void k_fft_lines(LinesArgs a, Factors fac, const float2 *__restrict__ tw_g) {
  extern __shared__ float2 lds[];
  const int n = a.n;
  constexpr int T = TB;
  float2 *A = lds, *B = A + n * BXC, *tw = B + n * BXC;
  for (int i = threadIdx.x; i < n; i += T) tw[i] = tw_g[i];
  const int ne = n * (BXC / 2);
  static_assert(NC == 0 || NC == 1, "three components: k_fft_lines3");
  const int nwork = a.nbundles;
  float4 v[LUX]; float2 K[LUX];
  auto decode = [&](int w, int &comp, int &bid) { comp = 0; bid = w; };
  auto locate = [&](int bid, int &o, int &chunk, int64_t &b) {
    chunk = bid % a.nchunk; const int rest = bid / a.nchunk; o = a.olo + rest % a.ocount; b = rest / a.ocount;
  };
  auto fetch = [&](int w) {
    int comp, bid; decode(w, comp, bid);
    bid = min(bid, a.nbundles - 1);
    int o, chunk; int64_t b; locate(bid, o, chunk, b);
    const float4 *src4 = reinterpret_cast<const float4 *>(a.src + bundle_off2(b, a.src_planes, n, a.nchunk, o, chunk));
#pragma unroll
    for (int u = 0; u < LUX; u++) { v[u] = make_float4(0.f, 0.f, 0.f, 0.f); if ((int)threadIdx.x + u * T < ne) v[u] = src4[(int)threadIdx.x + u * T]; }
    if (NC != 0) {
      const float2 *k2 = reinterpret_cast<const float2 *>(a.kern + comp * a.kern_comp_stride + b * a.kern_batch_stride + bundle_off2(0, a.src_planes, n, a.nchunk, o, chunk));
#pragma unroll
      for (int u = 0; u < LUX; u++) { K[u] = make_float2(0.f, 0.f); if ((int)threadIdx.x + u * T < ne) K[u] = k2[(int)threadIdx.x + u * T]; }
    }
  };
  int w = blockIdx.x;
  if (w < nwork) fetch(w);
  for (; w < nwork; w += gridDim.x) {
    int comp, bid; decode(w, comp, bid);
    const bool valid = bid < a.nbundles;   // uniform over the workgroup
    if (valid) {
#pragma unroll
      for (int u = 0; u < LUX; u++) {
        const int e = (int)threadIdx.x + u * T;
        if (e < ne) {
          float4 r = v[u];
          if (NC != 0) r = make_float4(-r.y * K[u].x, -(r.x * K[u].x), -r.w * K[u].y, -(r.z * K[u].y));  // (re,im)*i*K, then conj
          else if (INV) { r.y = -r.y; r.w = -r.w; }
          reinterpret_cast<float4 *>(A)[e] = r;
        }
      }
    }
    __syncthreads();
    { const int wn = w + gridDim.x; if (wn < nwork) fetch(wn); }   // in flight during the butterflies
    if (valid) {
      int o, chunk; int64_t b; locate(bid, o, chunk, b);
      const float2 *Z = fft_lds<false, RSET, BXC>(A, B, n, BXC, BXC, 1, fac, tw, 1);
      float2 *dbase = a.dst + comp * a.dst_comp_stride;
      const int e0 = a.slo * (BXC / 2), e1 = (a.slo + a.scount) * (BXC / 2);
      if (!TR) {
        float4 *dst4 = reinterpret_cast<float4 *>(dbase + bundle_off2(b, a.src_planes, n, a.nchunk, o, chunk));
        for (int e = e0 + threadIdx.x; e < e1; e += T) {
          float4 r = reinterpret_cast<const float4 *>(Z)[e];
          if (INV) { r.y = -r.y; r.w = -r.w; }
          dst4[e] = r;
        }
      } else {
        for (int e = e0 + threadIdx.x; e < e1; e += T) {
          const int idx = e >> 3, l4 = e & 7;
          float4 r = reinterpret_cast<const float4 *>(Z)[e];
          if (INV) { r.y = -r.y; r.w = -r.w; }
          reinterpret_cast<float4 *>(dbase + bundle_off2(b, a.dst_planes, a.dst_line, a.nchunk, idx, chunk) + (int64_t)o * BXC)[l4] = r;
        }
      }
    }
    __syncthreads();
  }
}

// The inverse z pass with the fused multiply, bundle-major: a work item is ONE bundle of rho-hat, loaded once and held
// in registers while the three force components i*K_c*rho-hat are formed, transformed and stored one after the other;
// K_{c+1} (and, behind the last component, the next bundle's rho-hat and K_0) is in flight during the butterflies of c.
// FWD: the bundle arrives BEFORE its forward z transform (the output of the forward y pass): the forward z pass runs
// here first, in LDS, and rho-hat goes from LDS straight into the registers -- it is never written to or read from HBM.
template <int RSET, int TB, int LUX, bool FWD>
__global__ __launch_bounds__(TB) void k_fft_lines3(LinesArgs a, Factors fac, const float2 *__restrict__ tw_g) {
  extern __shared__ float2 lds[];
  const int n = a.n;
  constexpr int T = TB;
  float2 *A = lds, *B = A + n * BXC, *tw = B + n * BXC;
  for (int i = threadIdx.x; i < n; i += T) tw[i] = tw_g[i];
  const int ne = n * (BXC / 2), nwork = a.nbundles;
  float4 v[LUX]; float2 K[LUX];
  auto locate = [&](int bid, int &o, int &chunk, int64_t &b) {
    chunk = bid % a.nchunk; const int rest = bid / a.nchunk; o = a.olo + rest % a.ocount; b = rest / a.ocount;
  };
  auto fetch_rho = [&](int bid) {
    int o, chunk; int64_t b; locate(bid, o, chunk, b);
    const float4 *src4 = reinterpret_cast<const float4 *>(a.src + bundle_off2(b, a.src_planes, n, a.nchunk, o, chunk));
#pragma unroll
    for (int u = 0; u < LUX; u++) { v[u] = make_float4(0.f, 0.f, 0.f, 0.f); if ((int)threadIdx.x + u * T < ne) v[u] = src4[(int)threadIdx.x + u * T]; }
  };
  auto fetch_k = [&](int bid, int comp) {
    int o, chunk; int64_t b; locate(bid, o, chunk, b);
    const float2 *k2 = reinterpret_cast<const float2 *>(a.kern + comp * a.kern_comp_stride + b * a.kern_batch_stride + bundle_off2(0, a.src_planes, n, a.nchunk, o, chunk));
#pragma unroll
    for (int u = 0; u < LUX; u++) { K[u] = make_float2(0.f, 0.f); if ((int)threadIdx.x + u * T < ne) K[u] = k2[(int)threadIdx.x + u * T]; }
  };
  int w = blockIdx.x;
  if (w < nwork) { fetch_rho(w); fetch_k(w, 0); }
  for (; w < nwork; w += gridDim.x) {
    int o, chunk; int64_t b; locate(w, o, chunk, b);
    if (FWD) {
#pragma unroll
      for (int u = 0; u < LUX; u++) { const int e = (int)threadIdx.x + u * T; if (e < ne) reinterpret_cast<float4 *>(A)[e] = v[u]; }
      __syncthreads();
      const float2 *Zf = fft_lds<false, RSET, BXC>(A, B, n, BXC, BXC, 1, fac, tw, 1);
#pragma unroll
      for (int u = 0; u < LUX; u++) { const int e = (int)threadIdx.x + u * T; if (e < ne) v[u] = reinterpret_cast<const float4 *>(Zf)[e]; }
      __syncthreads();
    }
#pragma unroll 1
    for (int comp = 0; comp < 3; comp++) {
#pragma unroll
      for (int u = 0; u < LUX; u++) {
        const int e = (int)threadIdx.x + u * T;
        if (e < ne) {
          const float4 r = v[u];
          reinterpret_cast<float4 *>(A)[e] = make_float4(-r.y * K[u].x, -(r.x * K[u].x), -r.w * K[u].y, -(r.z * K[u].y));  // (re,im)*i*K, then conj
        }
      }
      __syncthreads();
      if (comp < 2) fetch_k(w, comp + 1);
      else { const int wn = w + gridDim.x; if (wn < nwork) { fetch_rho(wn); fetch_k(wn, 0); } }
      const float2 *Z = fft_lds<false, RSET, BXC>(A, B, n, BXC, BXC, 1, fac, tw, 1);
      float2 *dbase = a.dst + comp * a.dst_comp_stride;
      const int e0 = a.slo * (BXC / 2), e1 = (a.slo + a.scount) * (BXC / 2);
      for (int e = e0 + threadIdx.x; e < e1; e += T) {
        const int idx = e >> 3, l4 = e & 7;
        float4 r = reinterpret_cast<const float4 *>(Z)[e];
        r.y = -r.y; r.w = -r.w;
        reinterpret_cast<float4 *>(dbase + bundle_off2(b, a.dst_planes, a.dst_line, a.nchunk, idx, chunk) + (int64_t)o * BXC)[l4] = r;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ y / z passes, two register stages (n = R1*R2)
// For line lengths that split into two radices <= 32 a line never makes more than ONE trip through LDS: thread (b, col)
// loads the R1 elements R2*a + b of column col straight from global memory into registers, transforms them (dft<R1>),
// applies the twiddle W_n^{b*k1} and writes Y_b[k1] to LDS; after the barrier thread (k1, col) reads Y_0..R2-1[k1],
// transforms (dft<R2>) and stores X[k1 + R1*k2] straight to global memory.  LDS traffic per element: 16 bytes instead of
// 16 per Stockham stage plus the staging copy; one exchange buffer of n*16 complex, so two workgroups share a CU.
// LDS element (k1, b, col) sits at ((k1*R2P + b)*16 + col), R2P = R2|1: the four line groups of a wave then fall in
// different bank halves on the stage-2 read, and the stage-1 write is contiguous.
// Stage-1 epilogue of the two-register-stage kernels: element k1 times W^(g*k1) into the exchange buffer.  The twiddle comes
// from the LDS table; read in the statement that uses it, every ds_read is followed by a wait a handful of instructions later
// (the kernels sit at their VGPR limit, so the compiler does not hoist them).  Here the reads run TWD elements ahead.
#ifndef P3M_TWD
#define P3M_TWD 2
#endif
template <int R, int STRIDE> __device__ __forceinline__ void store_twiddled(c32 *px, const c32 (&v)[R], const c32 *tw, int gq) {
  c32 w[P3M_TWD];
#pragma unroll
  for (int d = 0; d < P3M_TWD; d++) if (1 + d < R) w[d] = tw[__mul24(gq, 1 + d)];
  px[0] = v[0];
#pragma unroll
  for (int k1 = 1; k1 < R; k1++) {
    const c32 wk = w[(k1 - 1) % P3M_TWD];
    if (k1 + P3M_TWD < R) w[(k1 - 1) % P3M_TWD] = tw[__mul24(gq, k1 + P3M_TWD)];
    px[k1 * STRIDE] = vmulr(v[k1], wk);
  }
}
template <int R1, int R2> struct L2Cfg {
  static constexpr int n = R1 * R2, R2P = R2 | 1, S1 = R2 * BXC, S2 = R1 * BXC, TB = ((S1 > S2 ? S1 : S2) + 63) / 64 * 64;
  static constexpr size_t lds = sizeof(float2) * ((size_t)R1 * R2P * BXC + n);
  // wavefronts per SIMD the LDS lets in, capped at 4 (128 VGPRs); the long lines (one workgroup per CU) get the whole file
  static constexpr int wgs = (int)((size_t)160 * 1024 / lds), wv = ((TB / 64) * (wgs < 1 ? 1 : wgs) + 3) / 4, WPE = wv > 4 ? 4 : (wv < 1 ? 1 : wv);
};
// NC = 2 (INV, TR): one component, K at a.kern.  NC = 1 (INV, TR): the k-space multiply of particle_mesh_threaded.f90:183-192 on the way in, one force component after the other
// from the same bundle of rho-hat (re-read per component: the workgroup holds K_c and ONE component at a time, so two
// workgroups still share a CU); component c goes to dst + c*dst_comp_stride.
template <int R1, int R2, bool INV, bool TR, int NC = 0>
__global__ __launch_bounds__((L2Cfg<R1, R2>::TB)) __attribute__((amdgpu_waves_per_eu(L2Cfg<R1, R2>::WPE, L2Cfg<R1, R2>::WPE))) void k_fft_lines2(LinesArgs a, const float2 *__restrict__ tw_g) {
  using C = L2Cfg<R1, R2>;
  constexpr int n = C::n, R2P = C::R2P;
  extern __shared__ float2 lds[];
  c32 *X = reinterpret_cast<c32 *>(lds), *tw = X + R1 * R2P * BXC;
  for (int i = threadIdx.x; i < n; i += C::TB) tw[i] = reinterpret_cast<const c32 *>(tw_g)[i];
  const int col = threadIdx.x & (BXC - 1), g = threadIdx.x >> 4;
  const bool s1 = (C::S1 == C::TB) || g < R2, s2 = (C::S2 == C::TB) || g < R1;
  const int nwork = a.nbundles;
  auto locate = [&](int bid, int &o, int &chunk, int64_t &b) {
    chunk = bid % a.nchunk; const int rest = bid / a.nchunk; o = a.olo + rest % a.ocount; b = rest / a.ocount;
  };
  static_assert(NC == 0 || (INV && TR), "the fused multiply belongs to the transposing inverse z pass");
  c32 v[R1];
  auto fetch = [&](int w, int comp, bool on) {   // defines v on every path: a stale v would stay live through the whole loop body
#pragma unroll
    for (int m = 0; m < R1; m++) v[m] = (c32){0.f, 0.f};
    if (!on) return;
    int o, chunk; int64_t b; locate(w, o, chunk, b);
    if (!TR && a.seg > 0) {   // segmented source (see LinesArgs): 32-bit offsets inside one batch entry
      const c32 *base = reinterpret_cast<const c32 *>(a.src) + b * ((int64_t)a.src_planes * a.nchunk * n * BXC) + (unsigned)((o - a.olo) * a.nchunk + chunk) * (unsigned)a.seg * BXC + col;
      const unsigned sst = (unsigned)a.src_planes * a.nchunk * a.seg * BXC;
      fdiv_t ds; ds.m = a.seg_magic; ds.d = a.seg;
#pragma unroll
      for (int m = 0; m < R1; m++) {
        const int idx = R2 * m + g, t = fdiv(idx, ds);
        v[m] = __builtin_nontemporal_load(base + ((unsigned)t * sst + (unsigned)(idx - t * a.seg) * BXC));
      }
      return;
    }
    const c32 *src = reinterpret_cast<const c32 *>(a.src + bundle_off2(b, a.src_planes, n, a.nchunk, o, chunk)) + g * BXC + col;
#pragma unroll
    for (int m = 0; m < R1; m++) v[m] = TR ? src[m * (R2 * BXC)] : __builtin_nontemporal_load(src + m * (R2 * BXC));   // in place, read once and written once: past the caches
    if (NC != 0) {
      const float *k = a.kern + comp * a.kern_comp_stride + b * a.kern_batch_stride + bundle_off2(0, a.src_planes, n, a.nchunk, o, chunk) + g * BXC + col;
      float K[R1];
#pragma unroll
      for (int m = 0; m < R1; m++) K[m] = k[m * (R2 * BXC)];
#pragma unroll
      for (int m = 0; m < R1; m++) v[m] = (c32){-v[m].y * K[m], v[m].x * K[m]};   // i K rho-hat; the conj of the inverse follows below
    }
  };
  int w = blockIdx.x;
  __syncthreads();
  for (; w < nwork; w += gridDim.x)
  for (int comp = 0; comp < (NC == 1 ? 3 : 1); comp++) {
    fetch(w, comp, s1);
    if (s1) {
      if (INV) {
#pragma unroll
        for (int m = 0; m < R1; m++) v[m].y = -v[m].y;
      }
      dft<R1>(v);
      c32 *px = X + g * BXC + col;
      int gq = g; asm volatile("" : "+v"(gq));   // opaque: keeps the R1 twiddle offsets from being hoisted out of the loop into live registers
      store_twiddled<R1, R2P * BXC>(px, v, tw, gq);
    }
    __syncthreads();
    if (s2) {
      c32 u[R2];
      const c32 *px = X + (g * R2P) * BXC + col;
#pragma unroll
      for (int m = 0; m < R2; m++) u[m] = px[m * BXC];
      dft<R2>(u);
      int o, chunk; int64_t b; locate(w, o, chunk, b);
      c32 *d0; int64_t rstride;
      if (!TR) { d0 = reinterpret_cast<c32 *>(a.dst + bundle_off2(b, a.src_planes, n, a.nchunk, o, chunk)) + col; rstride = BXC; }
      else { d0 = reinterpret_cast<c32 *>(a.dst + bundle_off2(b, a.dst_planes, a.dst_line, a.nchunk, 0, chunk)) + o * BXC + col; rstride = (int64_t)a.nchunk * a.dst_line * BXC; }
      d0 += g * rstride;
      if (NC != 0) d0 += comp * a.dst_comp_stride;
      int r0 = g - a.slo; asm volatile("" : "+v"(r0));        // likewise the R2 row-range predicates
      asm volatile("" : "+s"(rstride));                        // and row offsets
      if (TR && a.direct_s > 0) {                              // the transposing store lands in the peers' receive blocks
        fdiv_t dd; dd.m = a.direct_magic; dd.d = a.direct_s;
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) {
          c32 r = u[k2];
          if (INV) r.y = -r.y;
          const int idx = g + R1 * k2, peer = fdiv(idx, dd);
          if ((unsigned)(r0 + R1 * k2) < (unsigned)a.scount) d0[(int64_t)(R1 * k2) * rstride + (int64_t)(peer - (int)b) * a.direct_delta] = r;
        }
      } else
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) {
        c32 r = u[k2];
        if (INV) r.y = -r.y;
        if ((unsigned)(r0 + R1 * k2) < (unsigned)a.scount) { if (TR) d0[(int64_t)(R1 * k2) * rstride] = r; else __builtin_nontemporal_store(r, d0 + (int64_t)(R1 * k2) * rstride); }
      }
    }
    __syncthreads();
  }
}

// The fused z pass (forward z, multiply, three inverse z transforms; see k_fft_lines3) on two register stages.  The forward
// transform runs as n = R1*R2 and leaves rho-hat element k1 + R1*k2 in thread (k1, col); the inverse transforms run as
// n = R2*R1 (the roles of the radices swapped), whose first stage wants exactly those elements in that thread: rho-hat
// stays in registers from the forward transform to the last component.  Two exchange buffers alternate, so that the
// second stage of component c (role A: R2 line groups) and the first stage of component c+1 (role B: R1 line groups)
// overlap across the waves of the workgroup and a bundle costs four barriers.
template <int R1, int R2> struct L3Cfg {
  static constexpr int n = R1 * R2, R2P = R2 | 1, R1P = R1 | 1, SA = R2 * BXC, SB = R1 * BXC, TB = ((SA > SB ? SA : SB) + 63) / 64 * 64;
  static constexpr int bufe = (R1 * R2P > R2 * R1P ? R1 * R2P : R2 * R1P) * BXC;
  static constexpr size_t lds = sizeof(float2) * ((size_t)2 * bufe + n);
};
template <int R1, int R2, bool KM>   // KM: LinesArgs::kmirror (a runtime branch around the loads cost the pass more than the halved table traffic bought)
__global__ __launch_bounds__((L3Cfg<R1, R2>::TB)) void k_fft_lines3r(LinesArgs a, const float2 *__restrict__ tw_g) {
  using C = L3Cfg<R1, R2>;
  constexpr int n = C::n, R2P = C::R2P, R1P = C::R1P;
  extern __shared__ float2 lds[];
  c32 *X0 = reinterpret_cast<c32 *>(lds), *X1 = X0 + C::bufe, *tw = X1 + C::bufe;
  for (int i = threadIdx.x; i < n; i += C::TB) tw[i] = reinterpret_cast<const c32 *>(tw_g)[i];
  const int col = threadIdx.x & (BXC - 1), g = threadIdx.x >> 4;
  const bool rA = (C::SA == C::TB) || g < R2, rB = (C::SB == C::TB) || g < R1;
  const int nwork = a.nbundles;
  auto locate = [&](int bid, int &o, int &chunk, int64_t &b) {
    chunk = bid % a.nchunk; const int rest = bid / a.nchunk; o = a.olo + rest % a.ocount; b = rest / a.ocount;
  };
  c32 v[R1];      // role A: the line elements R2*m + g on the way in
  c32 rh[R2];     // role B: rho-hat elements g + R1*m
  float K[R2];    // role B: K_c at those elements
  auto fetch_rho = [&](int w, bool on) {
#pragma unroll
    for (int m = 0; m < R1; m++) v[m] = (c32){0.f, 0.f};
    if (!on) return;
    int o, chunk; int64_t b; locate(w, o, chunk, b);
    const c32 *src = reinterpret_cast<const c32 *>(a.src + bundle_off2(b, a.src_planes, n, a.nchunk, o, chunk)) + g * BXC + col;
#pragma unroll
    for (int m = 0; m < R1; m++) v[m] = src[m * (R2 * BXC)];
  };
  auto fetch_k = [&](int w, int comp, bool on) {
#pragma unroll
    for (int m = 0; m < R2; m++) K[m] = 0.f;
    if (!on) return;
    int o, chunk; int64_t b; locate(w, o, chunk, b);
    const float *k = a.kern + comp * a.kern_comp_stride + b * a.kern_batch_stride + bundle_off2(0, a.src_planes, n, a.nchunk, o, chunk) + g * BXC + col;
    if constexpr (KM) {   // elements z = g + R1*m > n/2 from z' = n - z = (n - R1*m) - g: a second base, compile-time offsets
      const float *km = k - 2 * g * BXC;   // (the odd component's sign is applied where K is used: a multiply here would wait for the loads)
#pragma unroll
      for (int m = 0; m < R2; m++) K[m] = m >= (R2 + 1) / 2 ? km[(n - m * R1) * BXC] : k[m * (R1 * BXC)];
      return;
    }
#pragma unroll
    for (int m = 0; m < R2; m++) K[m] = k[m * (R1 * BXC)];
  };
  int w = blockIdx.x;
  fetch_rho(w, w < nwork && rA); fetch_k(w, 0, w < nwork && rB);
  __syncthreads();
  for (; w < nwork; w += gridDim.x) {
    int o, chunk; int64_t b; locate(w, o, chunk, b);
    // forward, stage 1 (role A) -> X0 as (k1, b = g)
    if (rA) {
      dft<R1>(v);
      c32 *px = X0 + g * BXC + col;
      int gq = g; asm volatile("" : "+v"(gq));   // opaque: keeps the R1 twiddle offsets from being hoisted out of the loop into live registers
      store_twiddled<R1, R2P * BXC>(px, v, tw, gq);
    }
    __syncthreads();
    // forward, stage 2 (role B): rho-hat g + R1*k2 into registers
    if (rB) {
      const c32 *px = X0 + (g * R2P) * BXC + col;
#pragma unroll
      for (int m = 0; m < R2; m++) rh[m] = px[m * BXC];
      dft<R2>(rh);
    }
    const int64_t rstride = (int64_t)a.nchunk * a.dst_line * BXC;
    c32 *d0 = reinterpret_cast<c32 *>(a.dst + bundle_off2(b, a.dst_planes, a.dst_line, a.nchunk, 0, chunk)) + o * BXC + col + g * rstride;
    const int r0 = g - a.slo;
#pragma unroll 1
    for (int comp = 0; comp < 3; comp++) {
      c32 *Xw = (comp & 1) ? X0 : X1;   // comp 0 -> X1, 1 -> X0, 2 -> X1
      // inverse, stage 1 (role B): elements g + R1*m = R1*m + b', b' = g; n = R2 * R1, first radix R2
      if (rB) {
        c32 t[R2];
#pragma unroll
        for (int m = 0; m < R2; m++) {
          const float km = (KM && m >= (R2 + 1) / 2 && comp == 2) ? -K[m] : K[m];   // K_z is odd along z: the mirrored half carries the other sign
          t[m] = (c32){-rh[m].y * km, -(rh[m].x * km)};   // conj(i K rho-hat)
        }
        dft<R2>(t);
        c32 *px = Xw + g * BXC + col;
        int gq = g; asm volatile("" : "+v"(gq));
        store_twiddled<R2, R1P * BXC>(px, t, tw, gq);
      }
      __syncthreads();
      if (comp < 2) fetch_k(w, comp + 1, rB);
      else { const int wn = w + gridDim.x; fetch_k(wn, 0, wn < nwork && rB); }
      // inverse, stage 2 (role A, k1' = g): elements g + R2*k2'
      if (rA) {
        c32 u[R1];
        const c32 *px = Xw + (g * R1P) * BXC + col;
#pragma unroll
        for (int m = 0; m < R1; m++) u[m] = px[m * BXC];
        dft<R1>(u);
        c32 *dc = d0 + comp * a.dst_comp_stride;
        int rq = r0; asm volatile("" : "+v"(rq));   // likewise the R1 row-range predicates
        int64_t rs = rstride; asm volatile("" : "+s"(rs));   // and the R1 row offsets
#pragma unroll
        for (int k2 = 0; k2 < R1; k2++)
          if ((unsigned)(rq + R2 * k2) < (unsigned)a.scount) dc[(int64_t)(R2 * k2) * rs] = (c32){u[k2].x, -u[k2].y};
      }
    }
    { const int wn = w + gridDim.x; fetch_rho(wn, wn < nwork && rA); }
    // X0 is rewritten by the next bundle's forward stage 1: the component-1 reads of X0 finished before the last barrier
  }
}

// layout converters for the probes / raw kernel upload (not on the hot path)
__global__ __launch_bounds__(256) void k_rows_to_lz(const float2 *__restrict__ rows, float2 *__restrict__ lz, int n, int px) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n * n * px) return;
  const int x = (int)(i % px); const int y = (int)((i / px) % n), z = (int)(i / ((int64_t)px * n));
  lz[bundle_off(0, n, px / BXC, y, x / BXC) + (int64_t)z * BXC + (x % BXC)] = rows[i];
}
__global__ __launch_bounds__(256) void k_lz_to_rows(const float2 *__restrict__ lz, float2 *__restrict__ rows, int n, int px) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n * n * px) return;
  const int x = (int)(i % px); const int y = (int)((i / px) % n), z = (int)(i / ((int64_t)px * n));
  rows[i] = lz[bundle_off(0, n, px / BXC, y, x / BXC) + (int64_t)z * BXC + (x % BXC)];
}

// ================================================================== host side
static bool factorize(int n, int *nfac, int *fac) {
  int m = n, k = 0;
  while (m % 16 == 0) { fac[k++] = 16; m /= 16; }
  while (m % 8 == 0) { fac[k++] = 8; m /= 8; }
  while (m % 4 == 0) { fac[k++] = 4; m /= 4; }
  while (m % 2 == 0) { fac[k++] = 2; m /= 2; }
  static const int odd[] = {3, 5, 7, 11, 13, 17, 19};
  for (int p : odd) while (m % p == 0) { fac[k++] = p; m /= p; if (k >= 12) return false; }
  *nfac = k;
  return m == 1 && k <= 12;
}

int fft_plan_create(FftPlan *pl, int n) {
  if (n < 4 || (n & 3)) { p3m_set_error("fft: n=%d must be a multiple of 4", n); return P3M_EINVAL; }
  if (n > P3M_STOCKHAM_MAX) {   // longer lines run the register-stage kernels only: both the line length and its half need an instance
    bool l2 = false, x2 = false;
#define X(N, A, B) if (n == N) l2 = true;
    P3M_LINES2_SIZES(X)
#undef X
#define X(H, A, B) if (n == 2 * H) x2 = true;
    P3M_X2_SIZES(X)
#undef X
    if (!l2 || !x2) { p3m_set_error("fft: n=%d > %d has no register-stage instance (640, 704, 768, 832, 896, 1024)", n, P3M_STOCKHAM_MAX); return P3M_EINVAL; }
  }
  pl->n = n;
  pl->px = ((n / 2 + 1) + 15) / 16 * 16;
  if (!factorize(n, &pl->nfac_full, pl->fac_full) || !factorize(n / 2, &pl->nfac_half, pl->fac_half)) {
    p3m_set_error("fft: n=%d has a prime factor > 19 (supported radices 2,3,4,5,7,8,11,13,17,19)", n);
    return P3M_EINVAL;
  }
  std::vector<float2> tw(n);
  for (int q = 0; q < n; q++) {
    const double a = -2.0 * M_PI * (double)q / (double)n;
    tw[q] = make_float2((float)cos(a), (float)sin(a));
  }
  HIP_TRY(hipMalloc(&pl->d_tw, sizeof(float2) * n));
  HIP_TRY(hipMemcpy(pl->d_tw, tw.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
  return P3M_OK;
}
void fft_plan_destroy(FftPlan *pl) { if (pl->d_tw) (void)hipFree(pl->d_tw); pl->d_tw = nullptr; }

static Factors mkfac(int nfac, const int *f) {
  Factors F{}; F.nfac = nfac;
  int n = 1; for (int i = 0; i < nfac; i++) n *= f[i];
  int Ns = 1;
  for (int i = 0; i < nfac; i++) { F.f[i] = f[i]; F.mNs[i] = fdiv_magic(Ns); F.mNb[i] = fdiv_magic(n / f[i]); Ns *= f[i]; }
  return F;
}

template <typename K> static int set_lds(K kern, size_t bytes) {
  if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return P3M_OK;
}
// rows per batch: the largest power of two <= 64 whose batch fits the XLU prefetch registers of both x kernels
static int x_rows(int n) {
  const int h = n / 2, ncol = (h + BXC) & ~(BXC - 1);
  int rb = 64;
  while (rb > 8 && (rb * h > 2816 || rb * (ncol / 2) > XLU * XTB || rb * (n / 4) > XLU * XTB)) rb >>= 1;
  return rb;
}
// persistent grid of the x kernels: what is resident at once, never more than there are batches
template <typename K> static int x_grid(K kern, size_t lds, int64_t nbatch) {
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), XTB, lds) != hipSuccess || occ < 1) occ = 1;
  const int64_t g = (int64_t)256 * occ;
  return (int)(g < nbatch ? g : (nbatch < 1 ? 1 : nbatch));
}

static int rset_of(int nfac, const int *fac) {
  int r = 0;
  for (int i = 0; i < nfac; i++) { if (fac[i] >= 17) r = 2; else if (fac[i] >= 11 && r < 1) r = 1; }
  return r;
}
template <int RSET, int RB> static int x_fwd_impl(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int64_t rows, int rpp) {
  const int n = pl.n;
  if (n > 8 * XLU * XTB / 2 || rows > 0x7fffffffLL) { p3m_set_error("fft x pass: n=%d or %lld rows out of range", n, (long long)rows); return P3M_EINVAL; }
  const size_t lds = sizeof(float2) * ((size_t)2 * (n / 2) * (RB + 1) + n);
  P3M_TRY((set_lds(k_fft_x_fwd<RSET, RB>, lds)));
  hipLaunchKernelGGL((k_fft_x_fwd<RSET, RB>), dim3(x_grid(k_fft_x_fwd<RSET, RB>, lds, cdiv(rows, RB))), dim3(XTB), lds, c->stream, src,
                     reinterpret_cast<float2 *>(dst), n, pl.px, (int)rows, mkfac(pl.nfac_half, pl.fac_half), pl.d_tw, rpp);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int RSET> static int x_fwd_rb(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int64_t rows, int rpp) {
  switch (x_rows(pl.n)) {
    case 8: return x_fwd_impl<RSET, 8>(c, pl, src, dst, rows, rpp);
    case 16: return x_fwd_impl<RSET, 16>(c, pl, src, dst, rows, rpp);
    case 32: return x_fwd_impl<RSET, 32>(c, pl, src, dst, rows, rpp);
    default: return x_fwd_impl<RSET, 64>(c, pl, src, dst, rows, rpp);
  }
}
template <int R1, int R2, bool CUBES = false, bool U8 = false> static int x_fwd2_impl(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int64_t rows, int rpp, const RowGeom &cq = RowGeom{}, float mass_p = 0.f) {
  using C = X2Cfg<R1, R2>;
  static_assert(!U8 || C::TB == 256, "the count table is built by 256 threads");
  if (rows > 0x7fffffffLL) { p3m_set_error("fft x pass: %lld rows out of range", (long long)rows); return P3M_EINVAL; }
  P3M_TRY((set_lds(k_fft_x_fwd2<R1, R2, CUBES, U8>, C::lds)));
  static int occ = 0;
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_fft_x_fwd2<R1, R2, CUBES, U8>), C::TB, C::lds));
    if (occ < 1) occ = 1;
  }
  const int64_t nbatch = cdiv(rows, C::RB), g = (int64_t)256 * occ;
  hipLaunchKernelGGL((k_fft_x_fwd2<R1, R2, CUBES, U8>), dim3((unsigned)(g < nbatch ? g : nbatch)), dim3(C::TB), C::lds, c->stream, src, reinterpret_cast<float2 *>(dst), pl.n,
                     pl.px, (int)rows, pl.d_tw, rpp, cq, mass_p);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
// the tile sizes whose forward x pass also exists reading one byte per cell (the NGP density of whole steps): two register stages, 256 threads
bool fft_x_forward_reads_u8(const FftPlan &pl) {
  if (lines2_off(pl.n)) return false;
#define X(H, A, B) if (pl.n == 2 * H) return X2Cfg<A, B>::TB == 256;
  P3M_X2_SIZES(X)
#undef X
  return false;
}
template <int R1, int R2> static int x_fwd2_u8(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int64_t rows, float mass_p) {
  if constexpr (X2Cfg<R1, R2>::TB == 256) return x_fwd2_impl<R1, R2, false, true>(c, pl, src, dst, rows, pl.n, RowGeom{}, mass_p);
  else { p3m_set_error("fft x pass: no byte-reading kernel for n=%d", pl.n); return P3M_EINVAL; }
}
int fft_x_forward_u8(p3m_ctx *c, const FftPlan &pl, const float *src8, float *dst, int batch, float mass_p) {
  if (!lines2_off(pl.n)) {
#define X(H, A, B) if (pl.n == 2 * H) return x_fwd2_u8<A, B>(c, pl, src8, dst, (int64_t)batch * pl.n * pl.n, mass_p);
    P3M_X2_SIZES(X)
#undef X
  }
  p3m_set_error("fft x pass: no byte-reading kernel for n=%d", pl.n); return P3M_EINVAL;
}
// the coarse sizes whose x passes read the ranks' cubes / write the ranks' force arrays themselves (instantiated for these only)
#define P3M_X2_CUBE_SIZES(X) X(32, 8, 4) X(64, 8, 8) X(128, 16, 8) X(256, 16, 16) X(512, 16, 32)
// (round 6: h = 512 as 16 x 32 -- the first stage, which also holds the next batch's loads, on the small radix: the forward pass of the literal
// 1024^3 slab transform 2.4 -> 1.86 ms per launch at 289 registers (361).  The force-writing inverse pass is the other way round -- its
// SECOND stage carries the three components' squares for max |F| -- and stays 32 x 16: 7.4 ms against 7.96)
#define P3M_X2_CUBE_INV_SIZES(X) X(32, 8, 4) X(64, 8, 8) X(128, 16, 8) X(256, 16, 16) X(512, 32, 16)
bool fft_x_has_cubes(const FftPlan &pl, const RowGeom &q) {
  if (lines2_off(pl.n) || q.ncn % 2 || (int64_t)q.nd * q.ncn * q.ncn * q.ncn >= 0x7fffffffLL) return false;
#define X(H, A, B) if (pl.n == 2 * H) return q.ncn % X2Cfg<A, B>::RB == 0;
  P3M_X2_CUBE_SIZES(X)
#undef X
  return false;
}
// forward x pass over the rows (rank, plane, y) of the slab decomposition, read from the ranks' cubes (k_fft_x_fwd2<.., CUBES>)
int fft_x_forward_cubes(p3m_ctx *c, const FftPlan &pl, const float *cubes, float *dst, const RowGeom &q) {
  const int64_t rows = (int64_t)q.nl * q.s * q.rpp;
#define X(H, A, B) if (pl.n == 2 * H) return x_fwd2_impl<A, B, true>(c, pl, cubes, dst, rows, q.rpp, q);
  P3M_X2_CUBE_SIZES(X)
#undef X
  p3m_set_error("fft_x_forward_cubes: n=%d has no cube-reading x pass", pl.n); return P3M_EINVAL;
}
// rpp: rows per plane of the LY output; 0 = pl.n (whole planes).  A pencil decomposition hands in planes of fewer rows.
int fft_x_forward_rows(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int64_t rows, int rpp) {
  if (rpp <= 0) rpp = pl.n;
  if (!lines2_off(pl.n)) {
#define X(H, A, B) if (pl.n == 2 * H) return x_fwd2_impl<A, B>(c, pl, src, dst, rows, rpp);
    P3M_X2_SIZES(X)
#undef X
  }
  switch (rset_of(pl.nfac_half, pl.fac_half)) {
    case 0: return x_fwd_rb<0>(c, pl, src, dst, rows, rpp);
    case 1: return x_fwd_rb<1>(c, pl, src, dst, rows, rpp);
    default: return x_fwd_rb<2>(c, pl, src, dst, rows, rpp);
  }
}
int fft_x_forward(p3m_ctx *c, const FftPlan &pl, const float *src, float *dst, int batch) {
  return fft_x_forward_rows(c, pl, src, dst, (int64_t)batch * pl.n * pl.n, 0);
}
template <int RSET, int RB>
static int x_inv_impl(p3m_ctx *c, const FftPlan &pl, const float *src, float *out, int batch, int mode, float *box, int fb, int lo, int ntile, int64_t bcs, int rpp) {
  const int n = pl.n;
  // mode 0: `batch` counts ROWS when negative (distributed slabs), whole n^2 arrays otherwise
  const int64_t rows = mode == 0 ? (batch < 0 ? -(int64_t)batch : (int64_t)batch * n * n) : (int64_t)batch * fb * fb;
  if (rows > 0x7fffffffLL) { p3m_set_error("fft x pass: %lld rows out of range", (long long)rows); return P3M_EINVAL; }
  const size_t lds = sizeof(float2) * ((size_t)2 * (n / 2 + 1) * (RB + 1) + n);
  const float scale = (float)n * (float)n * (float)n;  // real(nf_tile)**3, fftw2.f90:22
  P3M_TRY((set_lds(k_fft_x_inv<RSET, RB>, lds)));
  hipLaunchKernelGGL((k_fft_x_inv<RSET, RB>), dim3(x_grid(k_fft_x_inv<RSET, RB>, lds, cdiv(rows, RB))), dim3(XTB), lds, c->stream,
                     reinterpret_cast<const float2 *>(src), out, n, pl.px, (int)rows, mkfac(pl.nfac_half, pl.fac_half), pl.d_tw, scale, mode, box, fb, lo,
                     ntile, bcs, rpp);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int RSET>
static int x_inv_rb(p3m_ctx *c, const FftPlan &pl, const float *src, float *out, int batch, int mode, float *box, int fb, int lo, int ntile, int64_t bcs, int rpp) {
  switch (x_rows(pl.n)) {
    case 8: return x_inv_impl<RSET, 8>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
    case 16: return x_inv_impl<RSET, 16>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
    case 32: return x_inv_impl<RSET, 32>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
    default: return x_inv_impl<RSET, 64>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
  }
}
template <int R1, int R2, bool BOX> static int x_inv2_impl(p3m_ctx *c, const FftPlan &pl, const float *src, int64_t rows, float *box, int fb, int lo, int ntile, int64_t bcs, int rpp) {
  using C = X2Cfg<R1, R2>;
  const int n = pl.n;
  if (rows > 0x7fffffffLL) { p3m_set_error("fft x pass: %lld rows out of range", (long long)rows); return P3M_EINVAL; }
  const float scale = (float)n * (float)n * (float)n;
  P3M_TRY((set_lds(k_fft_x_inv2<R1, R2, BOX>, C::lds)));
  static int occ = 0;
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_fft_x_inv2<R1, R2, BOX>), C::TB, C::lds));
    if (occ < 1) occ = 1;
  }
  const int64_t nbatch = cdiv(rows, C::RB);
  const int64_t g = (int64_t)256 * occ;
  hipLaunchKernelGGL((k_fft_x_inv2<R1, R2, BOX>), dim3((unsigned)(g < nbatch ? g : nbatch)), dim3(C::TB), C::lds, c->stream, reinterpret_cast<const float2 *>(src), n,
                     pl.px, (int)rows, pl.d_tw, scale, box, fb, lo, ntile, bcs, rpp);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
// inverse x pass of the three force components over the rows of the slab decomposition (LY, [comp][rank][plane][y]), stored into the
// owners' force arrays with max |F| (k_fft_x_inv2c); only where fft_x_has_cubes
template <int R1, int R2> static int x_inv2c_impl(p3m_ctx *c, const FftPlan &pl, const float *src, float *fc, const RowGeom &q, const RankPtrs &red) {
  using C = X2Cfg<R1, R2>;
  const int n = pl.n; const int64_t rows1 = (int64_t)q.nl * q.s * q.rpp;
  if (3 * rows1 > 0x7fffffffLL) { p3m_set_error("fft x pass: %lld rows out of range", (long long)(3 * rows1)); return P3M_EINVAL; }
  const float scale = (float)n * (float)n * (float)n;
  P3M_TRY((set_lds(k_fft_x_inv2c<R1, R2>, C::lds)));
  static int occ = 0;
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_fft_x_inv2c<R1, R2>), C::TB, C::lds));
    if (occ < 1) occ = 1;
  }
  const int64_t nb1 = cdiv(rows1, C::RB), g = (int64_t)256 * occ;
  hipLaunchKernelGGL((k_fft_x_inv2c<R1, R2>), dim3((unsigned)(g < nb1 ? g : nb1)), dim3(C::TB), C::lds, c->stream, reinterpret_cast<const float2 *>(src), n, pl.px,
                     (int)rows1, pl.d_tw, scale, fc, q, red);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
int fft_x_inverse_cubes(p3m_ctx *c, const FftPlan &pl, const float *src, float *fc, const RowGeom &q, const RankPtrs &red) {
#define X(H, A, B) if (pl.n == 2 * H) return x_inv2c_impl<A, B>(c, pl, src, fc, q, red);
  P3M_X2_CUBE_INV_SIZES(X)
#undef X
  p3m_set_error("fft_x_inverse_cubes: n=%d has no force-writing x pass", pl.n); return P3M_EINVAL;
}
// the force-box inverse x pass of line length n is k_fft_x_inv2 (and can therefore run as k_fft_x_inv2_kick, kick_fused.hip)
bool fft_x2_box_pass(int n, int lo) {
  if ((lo & 1) || lines2_off(n)) return false;
#define X(H, A, B) if (n == 2 * H) return true;
  P3M_X2_SIZES(X)
#undef X
  return false;
}
// src in LY; mode 0 writes real ROWS to out, mode 1 the force box; rpp (mode 0): rows per plane of the LY input, 0 = pl.n
int fft_x_inverse(p3m_ctx *c, const FftPlan &pl, const float *src, float *out, int batch, int mode, float *box, int fb, int lo, int ntile, int64_t bcs, int rpp) {
  if (rpp <= 0) rpp = pl.n;
  if (mode == 1 && (lo & 1) == 0 && !lines2_off(pl.n)) {
#define X(H, A, B) if (pl.n == 2 * H) return x_inv2_impl<A, B, true>(c, pl, src, (int64_t)batch * fb * fb, box, fb, lo, ntile, bcs, rpp);
    P3M_X2_SIZES(X)
#undef X
  }
  if (mode == 0 && !lines2_off(pl.n)) {   // `batch` counts ROWS when negative (distributed slabs), whole n^2 arrays otherwise
    const int64_t rows = batch < 0 ? -(int64_t)batch : (int64_t)batch * pl.n * pl.n;
#define X(H, A, B) if (pl.n == 2 * H) return x_inv2_impl<A, B, false>(c, pl, src, rows, out, pl.n, 0, 1, 0, rpp);
    P3M_X2_SIZES(X)
#undef X
  }
  switch (rset_of(pl.nfac_half, pl.fac_half)) {
    case 0: return x_inv_rb<0>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
    case 1: return x_inv_rb<1>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
    default: return x_inv_rb<2>(c, pl, src, out, batch, mode, box, fb, lo, ntile, bcs, rpp);
  }
}
template <bool INV, bool TR, int NC, int RSET, int TB, int LUX> static int lines_impl(p3m_ctx *c, const FftPlan &pl, LinesArgs a, int batch) {
  const int n = pl.n;
  a.n = n; a.nchunk = pl.px / BXC; a.nbundles = batch * a.ocount * a.nchunk;
  const size_t lds = sizeof(float2) * ((size_t)2 * n * BXC + n);
  P3M_TRY((set_lds(k_fft_lines<INV, TR, NC, RSET, TB, LUX>, lds)));
  // persistent grid: as many workgroups as are resident at once
  const int nwork = a.nbundles;
  // never more than fit at once (a straggler wave of workgroups would double the time)
  static int occ_cache = 0, occ_lds = 0;
  if (occ_cache == 0 || occ_lds != (int)lds) {
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_fft_lines<INV, TR, NC, RSET, TB, LUX>), TB, lds));
    occ_cache = occ < 1 ? 1 : occ; occ_lds = (int)lds;
  }
  hipDeviceProp_t prop; int ncu = 256;
  (void)prop;
  int grid = ncu * occ_cache;
  if (grid > nwork) grid = nwork;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_fft_lines<INV, TR, NC, RSET, TB, LUX>), dim3((unsigned)grid), dim3(TB), lds, c->stream, a, mkfac(pl.nfac_full, pl.fac_full), pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int RSET, int TB, int LUX, bool FWD> static int lines3_impl(p3m_ctx *c, const FftPlan &pl, LinesArgs a, int batch) {
  const int n = pl.n;
  a.n = n; a.nchunk = pl.px / BXC; a.nbundles = batch * a.ocount * a.nchunk;
  const size_t lds = sizeof(float2) * ((size_t)2 * n * BXC + n);
  P3M_TRY((set_lds(k_fft_lines3<RSET, TB, LUX, FWD>, lds)));
  int occ = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_fft_lines3<RSET, TB, LUX, FWD>), TB, lds));
  int grid = 256 * (occ < 1 ? 1 : occ);
  if (grid > a.nbundles) grid = a.nbundles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_fft_lines3<RSET, TB, LUX, FWD>), dim3((unsigned)grid), dim3(TB), lds, c->stream, a, mkfac(pl.nfac_full, pl.fac_full), pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <bool FWD> static int launch_lines3_t(p3m_ctx *c, const FftPlan &pl, const LinesArgs &a, int batch) {
  const int rs = rset_of(pl.nfac_full, pl.fac_full);
#define L3(TBv, LUXv) (rs == 0 ? lines3_impl<0, TBv, LUXv, FWD>(c, pl, a, batch) : rs == 1 ? lines3_impl<1, TBv, LUXv, FWD>(c, pl, a, batch) : lines3_impl<2, TBv, LUXv, FWD>(c, pl, a, batch))
  if (pl.n <= 128) return L3(256, 4);
  if (pl.n <= 192) return L3(256, 6);
  if (pl.n <= 320) return L3(512, 5);
  return L3(1024, 5);
#undef L3
}
static bool lines2_has(int n) {
  if (lines2_off(n)) return false;
#define X(N, A, B) if (n == N) return true;
  P3M_LINES2_SIZES(X)
#undef X
  return false;
}
template <int R1, int R2> static int lines3r_impl(p3m_ctx *c, const FftPlan &pl, LinesArgs a, int batch) {
  using C = L3Cfg<R1, R2>;
  a.n = pl.n; a.nchunk = pl.px / BXC; a.nbundles = batch * a.ocount * a.nchunk;
  auto kern = a.kmirror ? k_fft_lines3r<R1, R2, true> : k_fft_lines3r<R1, R2, false>;
  P3M_TRY((set_lds(kern, C::lds)));
  static int occ2[2] = {0, 0};                        // per variant: the mirrored-table kernel and the plain one differ in registers
  int &occ = occ2[a.kmirror ? 1 : 0];
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), C::TB, C::lds));
    if (occ < 1) occ = 1;
  }
  int grid = 256 * occ;
  if (grid > a.nbundles) grid = a.nbundles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::TB), C::lds, c->stream, a, pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int R1, int R2, bool INV, bool TR, int NC> static int lines2_impl(p3m_ctx *c, const FftPlan &pl, LinesArgs a, int batch);
// P3M_Z_UNFUSED=1: the forward z pass and the multiply + inverse z pass as two launches of the one-buffer kernel even where the
// fused kernel fits (A/B measurements; lines longer than 608 always run this way)
static bool z_unfused() { static const bool on = getenv("P3M_Z_UNFUSED") && getenv("P3M_Z_UNFUSED")[0] == '1'; return on; }
template <bool INV, bool TR, int NC> static int launch_lines(p3m_ctx *c, const FftPlan &pl, const LinesArgs &a, int batch);
static int launch_lines3(p3m_ctx *c, const FftPlan &pl, const LinesArgs &a, int batch, bool fwd = false) {
  if (lines2_has(pl.n)) {
    if (fwd && !z_unfused()) {
#define X(N, A, B) if (pl.n == N) { if constexpr (L3Cfg<A, B>::lds <= 160 * 1024) return lines3r_impl<A, B>(c, pl, a, batch); }   // two exchange buffers must fit the LDS
      P3M_LINES2_SIZES(X)
#undef X
    }
    if (fwd) {   // forward z in place, then the multiply + inverse z from rho-hat
      LinesArgs f = a;
      f.dst = const_cast<float2 *>(a.src); f.kern = nullptr; f.slo = 0; f.scount = pl.n; f.dst_planes = a.src_planes; f.dst_line = pl.n;
      P3M_TRY((launch_lines<false, false, 0>(c, pl, f, batch)));
    }
#define X(N, A, B) if (pl.n == N) return lines2_impl<A, B, true, true, 1>(c, pl, a, batch);
    P3M_LINES2_SIZES(X)
#undef X
  }
  return fwd ? launch_lines3_t<true>(c, pl, a, batch) : launch_lines3_t<false>(c, pl, a, batch);
}
template <int R1, int R2, bool INV, bool TR, int NC> static int lines2_impl(p3m_ctx *c, const FftPlan &pl, LinesArgs a, int batch) {
  using C = L2Cfg<R1, R2>;
  a.n = pl.n; a.nchunk = pl.px / BXC; a.nbundles = batch * a.ocount * a.nchunk;
  P3M_TRY((set_lds(k_fft_lines2<R1, R2, INV, TR, NC>, C::lds)));
  static int occ = 0;
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(k_fft_lines2<R1, R2, INV, TR, NC>), C::TB, C::lds));
    if (occ < 1) occ = 1;
  }
  int grid = 256 * occ;
  if (grid > a.nbundles) grid = a.nbundles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_fft_lines2<R1, R2, INV, TR, NC>), dim3((unsigned)grid), dim3(C::TB), C::lds, c->stream, a, pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <bool INV, bool TR, int NC> static int launch_lines(p3m_ctx *c, const FftPlan &pl, const LinesArgs &a, int batch) {
  const int rs = rset_of(pl.nfac_full, pl.fac_full);
  if constexpr (NC == 1) if (lines2_has(pl.n)) {   // single-component multiply (fft3d_inverse with a kernel)
#define X(N, A, B) if (pl.n == N) return lines2_impl<A, B, INV, TR, 2>(c, pl, a, batch);
    P3M_LINES2_SIZES(X)
#undef X
  }
  if constexpr (NC == 0) if (lines2_has(pl.n)) {
#define X(N, A, B) if (pl.n == N) return lines2_impl<A, B, INV, TR, 0>(c, pl, a, batch);
    P3M_LINES2_SIZES(X)
#undef X
  }
  if (pl.n <= 128) {   // 256 threads hold a bundle in 4 float4 per lane
    if (rs == 0) return lines_impl<INV, TR, NC, 0, 256, 4>(c, pl, a, batch);
    if (rs == 1) return lines_impl<INV, TR, NC, 1, 256, 4>(c, pl, a, batch);
    return lines_impl<INV, TR, NC, 2, 256, 4>(c, pl, a, batch);
  }
  if (pl.n <= 192) {   // ... in 6
    if (rs == 0) return lines_impl<INV, TR, NC, 0, 256, 6>(c, pl, a, batch);
    if (rs == 1) return lines_impl<INV, TR, NC, 1, 256, 6>(c, pl, a, batch);
    return lines_impl<INV, TR, NC, 2, 256, 6>(c, pl, a, batch);
  }
  if (pl.n <= 320) {
    if (rs == 0) return lines_impl<INV, TR, NC, 0, 512, 5>(c, pl, a, batch);
    if (rs == 1) return lines_impl<INV, TR, NC, 1, 512, 5>(c, pl, a, batch);
    return lines_impl<INV, TR, NC, 2, 512, 5>(c, pl, a, batch);
  }
  if (rs == 0) return lines_impl<INV, TR, NC, 0, 1024, 5>(c, pl, a, batch);
  if (rs == 1) return lines_impl<INV, TR, NC, 1, 1024, 5>(c, pl, a, batch);
  return lines_impl<INV, TR, NC, 2, 1024, 5>(c, pl, a, batch);
}
static LinesArgs full_args(const FftPlan &pl, float *dst, const float *src) {
  LinesArgs a{};
  a.dst = reinterpret_cast<float2 *>(dst); a.src = reinterpret_cast<const float2 *>(src); a.kern = nullptr;
  a.olo = 0; a.ocount = pl.n; a.slo = 0; a.scount = pl.n;
  a.src_planes = pl.n; a.dst_planes = pl.n; a.dst_line = pl.n;
  return a;
}

// data: real ROWS in, rho-hat in LZ out (same array); scratch: one more array of the same size
int fft3d_forward(p3m_ctx *c, const FftPlan &pl, float *data, float *scratch, int batch) {
  P3M_TRY(fft_x_forward(c, pl, data, scratch, batch));                                        // ROWS -> LY
  P3M_TRY((launch_lines<false, true, 0>(c, pl, full_args(pl, data, scratch), batch)));        // LY -> LZ
  P3M_TRY((launch_lines<false, false, 0>(c, pl, full_args(pl, data, data), batch)));          // LZ in place
  return P3M_OK;
}

// full-size inverse (coarse mesh, probes): out (real ROWS) <- c2r(hat [* i*kern]) / n^3; tmp: scratch (LY)
int fft3d_inverse(p3m_ctx *c, const FftPlan &pl, const float *hat, float *tmp, float *out, int batch, const float *kern) {
  LinesArgs z = full_args(pl, tmp, hat);
  if (kern) { z.kern = kern; P3M_TRY((launch_lines<true, true, 1>(c, pl, z, batch))); }
  else P3M_TRY((launch_lines<true, true, 0>(c, pl, z, batch)));
  P3M_TRY((launch_lines<true, false, 0>(c, pl, full_args(pl, tmp, tmp), batch)));
  return fft_x_inverse(c, pl, tmp, out, batch, 0, nullptr, 0, 0, 1, 0, 0);
}

// fine mesh: the three force components of `batch` tiles from rho-hat (LZ), pruned to the force box.
// work holds 3*batch arrays ([comp][tile]); box points at tile0 of component 0, bcs = component stride.
// data: real ROWS in; out: after the forward x and y passes only (LZ layout, z still in real space) -- input of the
// fused z pass of fft_inverse3_box_z(..., zfwd = true)
// u8: the rows are bytes, the cells' counts, at the head of `data` (fft_x_forward_u8)
int fft3d_forward_xy(p3m_ctx *c, const FftPlan &pl, float *data, float *scratch, int batch, bool u8, float mass_p) {
  if (u8) P3M_TRY(fft_x_forward_u8(c, pl, data, scratch, batch, mass_p));
  else P3M_TRY(fft_x_forward(c, pl, data, scratch, batch));                                   // ROWS -> LY
  return launch_lines<false, true, 0>(c, pl, full_args(pl, data, scratch), batch);            // LY -> LZ
}
int fft_inverse3_box_z(p3m_ctx *c, const FftPlan &pl, const float *rho_hat, float *work, const float *kern3, int batch, int fb, int lo, bool zfwd) {
  LinesArgs z = full_args(pl, work, rho_hat);
  z.kmirror = (kern3 == c->kern_f && c->kf_zmirror) ? 1 : 0;
  z.kern = kern3; z.kern_comp_stride = (int64_t)pl.n * pl.n * pl.px;      // one float per complex element, LZ order
  z.dst_comp_stride = (int64_t)batch * pl.n * pl.n * pl.px;
  z.slo = lo; z.scount = fb;                                              // only box planes are stored
  return launch_lines3(c, pl, z, batch, zfwd);
}
int fft_inverse3_box_y(p3m_ctx *c, const FftPlan &pl, float *work, int batch, int fb, int lo) {
  LinesArgs y = full_args(pl, work, work);
  y.olo = lo; y.ocount = fb; y.slo = lo; y.scount = fb;                   // only box planes, only box rows
  return launch_lines<true, false, 0>(c, pl, y, 3 * batch);
}
int fft_inverse3_box(p3m_ctx *c, const FftPlan &pl, const float *rho_hat, float *work, const float *kern3, int batch, float *box, int fb, int lo,
                     int64_t bcs, bool zfwd) {
  P3M_TRY(fft_inverse3_box_z(c, pl, rho_hat, work, kern3, batch, fb, lo, zfwd));
  P3M_TRY(fft_inverse3_box_y(c, pl, work, batch, fb, lo));
  return fft_x_inverse(c, pl, work, nullptr, 3 * batch, 1, box, fb, lo, batch, bcs, 0);
}

int fft_rows_to_lz(p3m_ctx *c, const FftPlan &pl, const float *rows, float *lz) {
  const int64_t tot = (int64_t)pl.n * pl.n * pl.px;
  hipLaunchKernelGGL(k_rows_to_lz, dim3(cdiv(tot, 256)), dim3(256), 0, c->stream, reinterpret_cast<const float2 *>(rows), reinterpret_cast<float2 *>(lz), pl.n, pl.px);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
int fft_lz_to_rows(p3m_ctx *c, const FftPlan &pl, const float *lz, float *rows) {
  const int64_t tot = (int64_t)pl.n * pl.n * pl.px;
  hipLaunchKernelGGL(k_lz_to_rows, dim3(cdiv(tot, 256)), dim3(256), 0, c->stream, reinterpret_cast<const float2 *>(lz), reinterpret_cast<float2 *>(rows), pl.n, pl.px);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// benchmark hook: one pass kernel over `batch` tiles (see p3m_hip_time_fft_pass)
int fft_single_pass(p3m_ctx *c, const FftPlan &pl, int which, float *data, float *work, const float *kern, int batch, float *box, int fb, int lo,
                    int64_t bcs) {
  switch (which) {
    case 0: return fft_x_forward(c, pl, data, work, batch);
    case 1: return launch_lines<false, true, 0>(c, pl, full_args(pl, data, work), batch);
    case 2: return launch_lines<false, false, 0>(c, pl, full_args(pl, data, data), batch);
    case 3: return fft_inverse3_box_z(c, pl, data, work, kern, batch, fb, lo, true);   // as the step runs it: forward z pass fused in (data: rho after x,y)
    case 4: return fft_inverse3_box_y(c, pl, work, batch, fb, lo);
    case 5: return fft_x_inverse(c, pl, work, nullptr, 3 * batch, 1, box, fb, lo, batch, bcs, 0);
    case 6: {   // the multiply + inverse z pass on its own (one exchange buffer, rho-hat re-read per component): the second half of the un-fused z pair
      LinesArgs z = full_args(pl, work, data);
      z.kern = kern; z.kern_comp_stride = (int64_t)pl.n * pl.n * pl.px; z.dst_comp_stride = (int64_t)batch * pl.n * pl.n * pl.px;
      z.slo = lo; z.scount = fb;
#define X(N, A, B) if (pl.n == N) return lines2_impl<A, B, true, true, 1>(c, pl, z, batch);
      P3M_LINES2_SIZES(X)
#undef X
      p3m_set_error("fft_single_pass: no register-stage kernel for n=%d", pl.n);
      return P3M_EINVAL;
    }
  }
  p3m_set_error("fft_single_pass: bad selector %d", which);
  return P3M_EINVAL;
}

// ------------------------------------------------------------------ entry points for the distributed (slab) transforms, group.hip
// planes: bundle planes held locally (nc_slab); the transposing passes write the all-to-all send layout
// [line element][chunk][plane][16] directly.
// batch: logical ranks laid out one after the other (src and dst batch strides are one rank's slab)
// direct: the batch entries are ALL the ranks of the transpose (batch * planes == n): store into the peers' receive blocks of `send`
int fft_slab_y_fwd(p3m_ctx *c, const FftPlan &pl, const float *ly, float *send, int planes, int batch, bool direct) {
  LinesArgs a = full_args(pl, send, ly);
  a.ocount = planes; a.src_planes = planes; a.dst_planes = pl.n; a.dst_line = planes;
  if (direct) {
    if (!lines2_has(pl.n) || batch * planes != pl.n) return P3M_EINVAL;
    const int64_t rstride = (int64_t)(pl.px / BXC) * planes * BXC;
    a.direct_s = planes; a.direct_magic = fdiv_magic(planes); a.direct_delta = (int64_t)pl.n * rstride - (int64_t)planes * rstride;
  }
  return launch_lines<false, true, 0>(c, pl, a, batch);
}
// seg > 0: src is the all-to-all receive buffer, n/seg segments per line (LinesArgs::seg); register-stage kernels only (fft_has_segmented)
bool fft_has_segmented(const FftPlan &pl) { return lines2_has(pl.n); }
int fft_slab_z_fwd(p3m_ctx *c, const FftPlan &pl, const float *src, float *lz, int planes, int seg, int batch) {
  LinesArgs a = full_args(pl, lz, src);
  a.ocount = planes; a.src_planes = planes;
  if (seg > 0) { if (!lines2_has(pl.n) || pl.n % seg) return P3M_EINVAL; a.seg = seg; a.seg_magic = fdiv_magic(seg); }
  return launch_lines<false, false, 0>(c, pl, a, batch);
}
int fft_slab_z_inv3(p3m_ctx *c, const FftPlan &pl, const float *lz, float *send3, const float *kern3, int planes, int64_t kern_comp_stride,
                    int64_t send_comp_stride, int batch, int64_t kern_batch_stride, bool direct) {
  LinesArgs a = full_args(pl, send3, lz);
  a.ocount = planes; a.src_planes = planes; a.dst_planes = pl.n; a.dst_line = planes;
  a.kern = kern3; a.kern_comp_stride = kern_comp_stride; a.dst_comp_stride = send_comp_stride; a.kern_batch_stride = kern_batch_stride;
  if (direct) {   // see fft_slab_y_fwd
    if (!lines2_has(pl.n) || batch * planes != pl.n) return P3M_EINVAL;
    const int64_t rstride = (int64_t)(pl.px / BXC) * planes * BXC;
    a.direct_s = planes; a.direct_magic = fdiv_magic(planes); a.direct_delta = (int64_t)pl.n * rstride - (int64_t)planes * rstride;
  }
  return launch_lines3(c, pl, a, batch);
}
int fft_slab_y_inv(p3m_ctx *c, const FftPlan &pl, const float *src, float *ly3, int planes, int batch, int seg) {
  LinesArgs a = full_args(pl, ly3, src);
  a.ocount = planes; a.src_planes = planes;
  if (seg > 0) { if (!lines2_has(pl.n) || pl.n % seg) return P3M_EINVAL; a.seg = seg; a.seg_magic = fdiv_magic(seg); }
  return launch_lines<true, false, 0>(c, pl, a, batch);
}
