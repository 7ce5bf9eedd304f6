// fft.hip -- bespoke batched 3-D real FFT for gfx950 (no rocFFT/hipFFT).
//
// Replaces FFTW 2.1.5's rfftwnd_f77_one_real_to_complex / _complex_to_real as the reference
// calls them (source_threads/fftw2.f90:19-22): in-place r2c, unnormalised, sign -1, half-complex
// on the fastest axis, array (n+2, n, n) per tile; c2r followed by the division by n^3.
//
// Structure: three 1-D passes, each a Stockham autosort FFT done entirely in LDS by one
// workgroup over a bundle of lines (mixed radix 2,4,8 and odd primes up to 19 so that the
// tile sizes nf = pt + 48 = 2^4 * {5,7,11,19,35} are covered):
//   x : packed-real trick, one half-length complex FFT per row + split post/pre-processing
//   y,z: complex FFTs over lines that are strided in memory; a workgroup takes BX adjacent
//        x-columns so that every global access is a BX*8-byte contiguous segment.
// The k-space multiply  F^_c = i K_c rho^  (particle_mesh_threaded.f90:183-192) is fused into
// the load of the first inverse pass; the 1/n^3 and the force-box extraction (:202) into the
// store of the last one.  Bound: HBM (no MFMA: nothing here is a dense contraction).
#include "p3m_internal.h"
#include <math.h>

struct Factors { int nfac; int f[12]; };

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// ---- compile-time trig for the odd-radix butterflies -----------------------------------------
constexpr double ct_sin(double x) {  // |x| <= pi
  double term = x, sum = x;
  for (int i = 1; i < 24; i++) { term *= -x * x / ((2.0 * i) * (2.0 * i + 1.0)); sum += term; }
  return sum;
}
constexpr double ct_cos(double x) {
  double term = 1.0, sum = 1.0;
  for (int i = 1; i < 24; i++) { term *= -x * x / ((2.0 * i - 1.0) * (2.0 * i)); sum += term; }
  return sum;
}
template <int R> struct TrigTab {
  float c[R], s[R];
  constexpr TrigTab() : c(), s() {
    for (int k = 0; k < R; k++) {
      double a = 2.0 * 3.14159265358979323846 * k / R;
      if (a > 3.14159265358979323846) a -= 2.0 * 3.14159265358979323846;
      c[k] = (float)ct_cos(a); s[k] = (float)ct_sin(a);
    }
  }
};

// forward DFT (sign -1) of R values held in registers
template <int R> __device__ __forceinline__ void dft(float2 (&v)[R]);

template <> __device__ __forceinline__ void dft<2>(float2 (&v)[2]) {
  float2 a = v[0], b = v[1]; v[0] = cadd(a, b); v[1] = csub(a, b);
}
template <> __device__ __forceinline__ void dft<4>(float2 (&v)[4]) {
  float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
  float2 jd = make_float2(d.y, -d.x);  // -i*d
  v[0] = cadd(a, c); v[1] = cadd(b, jd); v[2] = csub(a, c); v[3] = csub(b, jd);
}
template <> __device__ __forceinline__ void dft<8>(float2 (&v)[8]) {
  const float r = 0.70710678118654752440f;
  float2 a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { a[k] = cadd(v[k], v[k + 4]); b[k] = csub(v[k], v[k + 4]); }
  b[1] = make_float2(r * (b[1].x + b[1].y), r * (b[1].y - b[1].x));    // * (1-i)/sqrt2
  b[2] = make_float2(b[2].y, -b[2].x);                                 // * -i
  b[3] = make_float2(r * (b[3].y - b[3].x), -r * (b[3].x + b[3].y));   // * (-1-i)/sqrt2
  dft<4>(a); dft<4>(b);
#pragma unroll
  for (int q = 0; q < 4; q++) { v[2 * q] = a[q]; v[2 * q + 1] = b[q]; }
}
template <int R> __device__ __forceinline__ void dft_odd(float2 (&v)[R]) {
  constexpr int H = (R - 1) / 2;
  constexpr TrigTab<R> tab{};
  float2 t[H + 1], u[H + 1];
#pragma unroll
  for (int p = 1; p <= H; p++) { t[p] = cadd(v[p], v[R - p]); u[p] = csub(v[p], v[R - p]); }
  float2 v0 = v[0], s0 = v[0];
#pragma unroll
  for (int p = 1; p <= H; p++) s0 = cadd(s0, t[p]);
  v[0] = s0;
#pragma unroll
  for (int a = 1; a <= H; a++) {
    float2 A = v0, B = make_float2(0.f, 0.f);
#pragma unroll
    for (int p = 1; p <= H; p++) {
      const float cc = tab.c[(a * p) % R], ss = tab.s[(a * p) % R];
      A.x += cc * t[p].x; A.y += cc * t[p].y; B.x += ss * u[p].x; B.y += ss * u[p].y;
    }
    v[a] = make_float2(A.x + B.y, A.y - B.x);      // A - iB
    v[R - a] = make_float2(A.x - B.y, A.y + B.x);  // A + iB
  }
}
template <> __device__ __forceinline__ void dft<3>(float2 (&v)[3]) { dft_odd<3>(v); }
template <> __device__ __forceinline__ void dft<5>(float2 (&v)[5]) { dft_odd<5>(v); }
template <> __device__ __forceinline__ void dft<7>(float2 (&v)[7]) { dft_odd<7>(v); }
template <> __device__ __forceinline__ void dft<11>(float2 (&v)[11]) { dft_odd<11>(v); }
template <> __device__ __forceinline__ void dft<13>(float2 (&v)[13]) { dft_odd<13>(v); }
template <> __device__ __forceinline__ void dft<17>(float2 (&v)[17]) { dft_odd<17>(v); }
template <> __device__ __forceinline__ void dft<19>(float2 (&v)[19]) { dft_odd<19>(v); }

// One radix-R Stockham stage on `nl` lines of length n held in LDS.
// element (idx,line) lives at idx*sI + line*sL.  ROWS: lanes run along the line (x pass);
// otherwise lanes run across lines (strided passes).  tw[q*twm] = exp(-2 pi i q / n).
template <int R, bool ROWS>
__device__ __forceinline__ void fft_stage(const float2 *__restrict__ in, float2 *__restrict__ out, int n, int Ns, int nl,
                                          int sI, int sL, const float2 *__restrict__ tw, int twm) {
  const int nb = n / R, ntask = nb * nl;
  const int tstep = (n / (Ns * R)) * twm;
  for (int task = threadIdx.x; task < ntask; task += blockDim.x) {
    int j, line;
    if (ROWS) { line = task / nb; j = task - line * nb; } else { j = task / nl; line = task - j * nl; }
    const int k = j % Ns;
    const float2 *pin = in + line * sL;
    float2 v[R];
#pragma unroll
    for (int m = 0; m < R; m++) v[m] = pin[(j + m * nb) * sI];
    if (Ns > 1) {
      const int ts = tstep * k;
#pragma unroll
      for (int m = 1; m < R; m++) v[m] = cmul(v[m], tw[m * ts]);
    }
    dft<R>(v);
    const int j0 = (j - k) * R + k;
    float2 *pout = out + line * sL;
#pragma unroll
    for (int m = 0; m < R; m++) pout[(j0 + m * Ns) * sI] = v[m];
  }
}

template <bool ROWS>
__device__ __forceinline__ float2 *fft_lds(float2 *A, float2 *B, int n, int nl, int sI, int sL, const Factors &fac,
                                           const float2 *tw, int twm) {
  int Ns = 1;
  float2 *in = A, *out = B;
  for (int s = 0; s < fac.nfac; s++) {
    const int R = fac.f[s];
    switch (R) {
      case 2: fft_stage<2, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 3: fft_stage<3, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 4: fft_stage<4, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 5: fft_stage<5, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 7: fft_stage<7, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 8: fft_stage<8, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 11: fft_stage<11, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 13: fft_stage<13, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 17: fft_stage<17, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
      case 19: fft_stage<19, ROWS>(in, out, n, Ns, nl, sI, sL, tw, twm); break;
    }
    Ns *= R;
    __syncthreads();
    float2 *t = in; in = out; out = t;
  }
  return in;
}

// ------------------------------------------------------------------ x pass, forward (r2c)
// rows_total real rows of length n (row pitch n+2 floats); RB rows per workgroup.
template <int RB>
__global__ __launch_bounds__(256) void k_fft_x_fwd(float *__restrict__ data, int n, int rows_total, Factors fac,
                                                   const float2 *__restrict__ tw_g) {
  extern __shared__ float2 lds[];
  const int h = n >> 1, LP = h + 1;
  float2 *A = lds, *B = A + RB * LP, *tw = B + RB * LP;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
  for (int e = threadIdx.x; e < nrows * h; e += blockDim.x) {
    const int r = e / h, m = e - r * h;
    A[r * LP + m] = reinterpret_cast<const float2 *>(data + (row0 + r) * (int64_t)(n + 2))[m];
  }
  __syncthreads();
  const float2 *Z = fft_lds<true>(A, B, h, nrows, 1, LP, fac, tw, 2);
  // X[k] = E + W_n^k O,  E = (Z[k]+conj Z[h-k])/2,  O = (Z[k]-conj Z[h-k])/(2i)
  for (int e = threadIdx.x; e < nrows * (h + 1); e += blockDim.x) {
    const int r = e / (h + 1), k = e - r * (h + 1);
    const float2 zk = Z[r * LP + (k == h ? 0 : k)];
    const float2 zc = cconj(Z[r * LP + (k == 0 ? 0 : h - k)]);
    const float2 E = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
    const float2 O = make_float2(0.5f * (zk.y - zc.y), -0.5f * (zk.x - zc.x));
    const float2 w = (k == h) ? make_float2(-1.f, 0.f) : tw[k];
    const float2 X = cadd(E, cmul(O, w));
    reinterpret_cast<float2 *>(data + (row0 + r) * (int64_t)(n + 2))[k] = X;
  }
}

// ------------------------------------------------------------------ x pass, inverse (c2r) + /n^3
// mode 0: in place, all rows.  mode 1: only the rows/columns of the force box
// force_f(c, nb-1:nf-nb+1,...) (particle_mesh_threaded.f90:202) are produced and written to
// box[tile][fb][fb][fb]; lo = nb-2 (0-based first cell of the box).
template <int RB>
__global__ __launch_bounds__(256) void k_fft_x_inv(float *__restrict__ data, int n, int rows_total, Factors fac,
                                                   const float2 *__restrict__ tw_g, float inv_scale, int mode,
                                                   float *__restrict__ box, int fb, int lo) {
  extern __shared__ float2 lds[];
  const int h = n >> 1, LP = h + 1;
  float2 *A = lds, *B = A + RB * LP, *tw = B + RB * LP;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
  __syncthreads();
  // Z'[m] = (X[m] + conj X[h-m]) + i (X[m] - conj X[h-m]) W_n^{-m}; conj() on the way in: the
  // forward machinery then yields conj(IFFT), undone on the way out.
  for (int e = threadIdx.x; e < nrows * h; e += blockDim.x) {
    const int r = e / h, m = e - r * h;
    int64_t srow = row0 + r;
    if (mode == 1) {  // row index enumerates (tile, kk, jj) of the box
      const int jj = (int)(srow % fb); const int64_t t2 = srow / fb; const int kk = (int)(t2 % fb); const int64_t tile = t2 / fb;
      srow = (tile * n + (kk + lo)) * n + (jj + lo);
    }
    const float2 *X = reinterpret_cast<const float2 *>(data + srow * (int64_t)(n + 2));
    const float2 xk = X[m], xc = cconj(X[h - m]);
    const float2 e2 = cadd(xk, xc), d = csub(xk, xc);
    const float2 o = cmul(d, cconj(tw[m]));
    A[r * LP + m] = make_float2(e2.x - o.y, -(e2.y + o.x));  // conj(e + i o)
  }
  __syncthreads();
  const float2 *Z = fft_lds<true>(A, B, h, nrows, 1, LP, fac, tw, 2);
  if (mode == 0) {
    for (int e = threadIdx.x; e < nrows * (h + 1); e += blockDim.x) {
      const int r = e / (h + 1), m = e - r * (h + 1);
      float2 z = make_float2(0.f, 0.f);
      if (m < h) { z = Z[r * LP + m]; z = make_float2(z.x / inv_scale, -z.y / inv_scale); }
      reinterpret_cast<float2 *>(data + (row0 + r) * (int64_t)(n + 2))[m] = z;
    }
  } else {
    for (int e = threadIdx.x; e < nrows * fb; e += blockDim.x) {
      const int r = e / fb, ii = e - r * fb;
      const int x = ii + lo;
      const float2 z = Z[r * LP + (x >> 1)];
      const float val = (x & 1) ? -z.y : z.x;
      box[(row0 + r) * (int64_t)fb + ii] = val / inv_scale;
    }
  }
}

// ------------------------------------------------------------------ y / z passes (strided lines)
// layout [b][z][y][x], x in [0,hx).  axis 1: lines along y (one z per workgroup), axis 2: along z.
// A workgroup owns BX adjacent x columns.  FUSE: read src, multiply by i*K on the fly.
template <int BX, bool INV, bool FUSE>
__global__ __launch_bounds__(256) void k_fft_lines(float2 *__restrict__ dst, const float2 *__restrict__ src,
                                                   const float *__restrict__ kern, int n, int hx, int axis, int nchunk,
                                                   Factors fac, const float2 *__restrict__ tw_g) {
  extern __shared__ float2 lds[];
  float2 *A = lds, *B = A + n * BX, *tw = B + n * BX;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  const int ch = blockIdx.x % nchunk;
  const int64_t rest = blockIdx.x / nchunk;
  const int o = (int)(rest % n);
  const int64_t b = rest / n;
  const int x0 = ch * BX, nl = min(BX, hx - x0);
  const int64_t plane = (int64_t)n * hx;
  int64_t base, stride;
  if (axis == 1) { base = (b * n + o) * plane + x0; stride = hx; }
  else { base = b * n * plane + (int64_t)o * hx + x0; stride = plane; }
  int64_t kbase = 0;
  if (FUSE) kbase = base - b * n * plane;  // kern has no batch dimension
  for (int e = threadIdx.x; e < n * nl; e += blockDim.x) {
    const int idx = e / nl, l = e - idx * nl;
    float2 v = src[base + idx * stride + l];
    if (FUSE) { const float K = kern[kbase + idx * stride + l]; v = make_float2(-v.y * K, v.x * K); }
    if (INV) v.y = -v.y;
    A[idx * nl + l] = v;
  }
  __syncthreads();
  const float2 *Z = fft_lds<false>(A, B, n, nl, nl, 1, fac, tw, 1);
  for (int e = threadIdx.x; e < n * nl; e += blockDim.x) {
    const int idx = e / nl, l = e - idx * nl;
    float2 v = Z[idx * nl + l];
    if (INV) v.y = -v.y;
    dst[base + idx * stride + l] = v;
  }
}

// ================================================================== host side
static bool factorize(int n, int *nfac, int *fac) {
  int m = n, k = 0;
  while (m % 8 == 0) { fac[k++] = 8; m /= 8; }
  while (m % 4 == 0) { fac[k++] = 4; m /= 4; }
  while (m % 2 == 0) { fac[k++] = 2; m /= 2; }
  static const int odd[] = {3, 5, 7, 11, 13, 17, 19};
  for (int p : odd) while (m % p == 0) { fac[k++] = p; m /= p; if (k >= 12) return false; }
  *nfac = k;
  return m == 1 && k <= 12;
}

int fft_plan_create(FftPlan *pl, int n) {
  if (n < 4 || (n & 1)) { p3m_set_error("fft: n=%d must be even and >= 4", n); return P3M_EINVAL; }
  pl->n = n;
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

static Factors mkfac(int nfac, const int *f) { Factors F; F.nfac = nfac; for (int i = 0; i < nfac; i++) F.f[i] = f[i]; return F; }

template <typename K> static int set_lds(K kern, size_t bytes) {
  if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return P3M_OK;
}

template <int RB> static int launch_x_fwd(p3m_ctx *c, const FftPlan &pl, float *data, int64_t rows) {
  const int n = pl.n; const size_t lds = sizeof(float2) * ((size_t)2 * RB * (n / 2 + 1) + n);
  P3M_TRY(set_lds(k_fft_x_fwd<RB>, lds));
  hipLaunchKernelGGL(k_fft_x_fwd<RB>, dim3(cdiv(rows, RB)), dim3(256), lds, c->stream, data, n, (int)rows,
                     mkfac(pl.nfac_half, pl.fac_half), pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int RB> static int launch_x_inv(p3m_ctx *c, const FftPlan &pl, float *data, int64_t rows, int mode, float *box, int fb, int lo) {
  const int n = pl.n; const size_t lds = sizeof(float2) * ((size_t)2 * RB * (n / 2 + 1) + n);
  const float scale = (float)n * (float)n * (float)n;  // real(nf_tile)**3, fftw2.f90:22
  P3M_TRY(set_lds(k_fft_x_inv<RB>, lds));
  hipLaunchKernelGGL(k_fft_x_inv<RB>, dim3(cdiv(rows, RB)), dim3(256), lds, c->stream, data, n, (int)rows,
                     mkfac(pl.nfac_half, pl.fac_half), pl.d_tw, scale, mode, box, fb, lo);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int BX, bool INV, bool FUSE>
static int launch_lines(p3m_ctx *c, const FftPlan &pl, float *dst, const float *src, const float *kern, int axis, int batch) {
  const int n = pl.n, hx = n / 2 + 1, nchunk = cdiv(hx, BX);
  const size_t lds = sizeof(float2) * ((size_t)2 * n * BX + n);
  P3M_TRY(set_lds(k_fft_lines<BX, INV, FUSE>, lds));
  const int64_t blocks = (int64_t)batch * n * nchunk;
  hipLaunchKernelGGL((k_fft_lines<BX, INV, FUSE>), dim3((unsigned)blocks), dim3(256), lds, c->stream,
                     reinterpret_cast<float2 *>(dst), reinterpret_cast<const float2 *>(src), kern, n, hx, axis, nchunk,
                     mkfac(pl.nfac_full, pl.fac_full), pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

int fft_x_forward(p3m_ctx *c, const FftPlan &pl, float *data, int batch) {
  const int64_t rows = (int64_t)batch * pl.n * pl.n;
  return pl.n <= 256 ? launch_x_fwd<8>(c, pl, data, rows) : launch_x_fwd<4>(c, pl, data, rows);
}
int fft_x_inverse(p3m_ctx *c, const FftPlan &pl, float *data, int batch, int mode, float *box, int fb, int lo) {
  const int64_t rows = mode == 0 ? (int64_t)batch * pl.n * pl.n : (int64_t)batch * fb * fb;
  return pl.n <= 256 ? launch_x_inv<8>(c, pl, data, rows, mode, box, fb, lo) : launch_x_inv<4>(c, pl, data, rows, mode, box, fb, lo);
}
template <bool INV, bool FUSE>
static int lines_any(p3m_ctx *c, const FftPlan &pl, float *dst, const float *src, const float *kern, int axis, int batch) {
  return pl.n <= 320 ? launch_lines<16, INV, FUSE>(c, pl, dst, src, kern, axis, batch)
                     : launch_lines<8, INV, FUSE>(c, pl, dst, src, kern, axis, batch);
}

int fft3d_forward(p3m_ctx *c, const FftPlan &pl, float *data, int batch) {
  P3M_TRY(fft_x_forward(c, pl, data, batch));
  P3M_TRY((lines_any<false, false>(c, pl, data, data, nullptr, 1, batch)));
  P3M_TRY((lines_any<false, false>(c, pl, data, data, nullptr, 2, batch)));
  return P3M_OK;
}

// z, y strided passes of the inverse (shared by the in-place and the force-box variants)
int fft3d_inverse_zy(p3m_ctx *c, const FftPlan &pl, float *data, int batch, const float *src, const float *kern) {
  if (kern) P3M_TRY((lines_any<true, true>(c, pl, data, src, kern, 2, batch)));
  else P3M_TRY((lines_any<true, false>(c, pl, data, src ? src : data, nullptr, 2, batch)));
  P3M_TRY((lines_any<true, false>(c, pl, data, data, nullptr, 1, batch)));
  return P3M_OK;
}

int fft3d_inverse(p3m_ctx *c, const FftPlan &pl, float *data, int batch, const float *src, const float *kern) {
  P3M_TRY(fft3d_inverse_zy(c, pl, data, batch, src, kern));
  return fft_x_inverse(c, pl, data, batch, 0, nullptr, 0, 0);
}

// benchmark hook: one pass kernel over `batch` tiles (see p3m_hip_time_fft_pass)
int fft_single_pass(p3m_ctx *c, const FftPlan &pl, int which, float *data, float *work, const float *kern, int batch, float *box, int fb, int lo) {
  switch (which) {
    case 0: return fft_x_forward(c, pl, data, batch);
    case 1: return lines_any<false, false>(c, pl, data, data, nullptr, 1, batch);
    case 2: return lines_any<false, false>(c, pl, data, data, nullptr, 2, batch);
    case 3: return lines_any<true, true>(c, pl, work, data, kern, 2, batch);
    case 4: return lines_any<true, false>(c, pl, work, work, nullptr, 1, batch);
    case 5: return fft_x_inverse(c, pl, work, batch, 1, box, fb, lo);
  }
  p3m_set_error("fft_single_pass: bad selector %d", which);
  return P3M_EINVAL;
}
