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

// ------------------------------------------------------------------ memory layout
// One array is [n][n][px] complex (= [n][n][2*px] real), px = n/2+1 rounded up to 16 so that every
// row starts on a 128-byte line and a bundle of BX=16 columns is exactly one line.  The pad
// columns hold zeros and are transformed along (independent lines).

// ------------------------------------------------------------------ x pass, forward (r2c)
// rows_total real rows of length n (row pitch 2*px floats); RB rows per workgroup.
template <int RB>
__global__ __launch_bounds__(256) void k_fft_x_fwd(float *__restrict__ data, int n, int px, int rows_total, Factors fac,
                                                   const float2 *__restrict__ tw_g) {
  extern __shared__ float2 lds[];
  const int h = n >> 1, LP = h + 2, h2 = h >> 1;
  float2 *A = lds, *B = A + RB * LP, *tw = B + RB * LP;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
  for (int e = threadIdx.x; e < nrows * h2; e += blockDim.x) {
    const int r = e / h2, m = e - r * h2;
    const float4 v = reinterpret_cast<const float4 *>(data + (row0 + r) * (int64_t)(2 * px))[m];
    *reinterpret_cast<float4 *>(&A[r * LP + 2 * m]) = v;
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
    reinterpret_cast<float2 *>(data + (row0 + r) * (int64_t)(2 * px))[k] = X;
  }
}

// ------------------------------------------------------------------ x pass, inverse (c2r) + /n^3
// mode 0: in place, all rows.  mode 1: rows enumerate (b, kk, jj) over the force box
// force_f(c, nb-1:nf-nb+1,...) (particle_mesh_threaded.f90:202); b = comp*ntile + tile; only the box
// columns are written, to box + comp*box_comp_stride + tile*fb^3; lo = nb-2 (first box cell).
template <int RB>
__global__ __launch_bounds__(256) void k_fft_x_inv(float *__restrict__ data, int n, int px, int rows_total, Factors fac,
                                                   const float2 *__restrict__ tw_g, float inv_scale, int mode,
                                                   float *__restrict__ box, int fb, int lo, int ntile, int64_t box_comp_stride) {
  extern __shared__ float2 lds[];
  const int h = n >> 1, LP = h + 2;
  float2 *A = lds, *B = A + RB * LP, *tw = B + RB * LP;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  const int64_t row0 = (int64_t)blockIdx.x * RB;
  const int nrows = (int)min((int64_t)RB, (int64_t)rows_total - row0);
  __syncthreads();
  // Z'[m] = (X[m] + conj X[h-m]) + i (X[m] - conj X[h-m]) W_n^{-m}; conj() on the way in: the
  // forward machinery then yields conj(IFFT), undone on the way out.
  __shared__ int64_t src_row[RB], dst_off[RB];
  if (threadIdx.x < nrows) {
    int64_t srow = row0 + threadIdx.x, dofs = 0;
    if (mode == 1) {
      const int jj = (int)(srow % fb); const int64_t t2 = srow / fb; const int kk = (int)(t2 % fb); const int64_t b = t2 / fb;
      const int comp = (int)(b / ntile), tl = (int)(b % ntile);
      srow = (b * n + (kk + lo)) * n + (jj + lo);
      dofs = comp * box_comp_stride + (((int64_t)tl * fb + kk) * fb + jj) * fb;
    }
    src_row[threadIdx.x] = srow; dst_off[threadIdx.x] = dofs;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nrows * h; e += blockDim.x) {
    const int r = e / h, m = e - r * h;
    const float2 *X = reinterpret_cast<const float2 *>(data + src_row[r] * (int64_t)(2 * px));
    const float2 xk = X[m], xc = cconj(X[h - m]);
    const float2 e2 = cadd(xk, xc), d = csub(xk, xc);
    const float2 o = cmul(d, cconj(tw[m]));
    A[r * LP + m] = make_float2(e2.x - o.y, -(e2.y + o.x));  // conj(e + i o)
  }
  __syncthreads();
  const float2 *Z = fft_lds<true>(A, B, h, nrows, 1, LP, fac, tw, 2);
  if (mode == 0) {
    for (int e = threadIdx.x; e < nrows * px; e += blockDim.x) {
      const int r = e / px, m = e - r * px;
      float2 z = make_float2(0.f, 0.f);
      if (m < h) { z = Z[r * LP + m]; z = make_float2(z.x / inv_scale, -z.y / inv_scale); }
      reinterpret_cast<float2 *>(data + (row0 + r) * (int64_t)(2 * px))[m] = z;
    }
  } else {
    for (int e = threadIdx.x; e < nrows * fb; e += blockDim.x) {
      const int r = e / fb, ii = e - r * fb;
      const int x = ii + lo;
      const float2 z = Z[r * LP + (x >> 1)];
      const float val = (x & 1) ? -z.y : z.x;
      box[dst_off[r] + ii] = val / inv_scale;
    }
  }
}

// ------------------------------------------------------------------ y / z passes (strided lines)
// layout [b][z][y][x], x in [0,px).  axis 1: lines along y (one z per workgroup), axis 2: along z.
// A workgroup owns BX adjacent x columns = BX/2 float4 per row segment.
// NC = 0: plain transform src -> dst.
// NC = 1|3: fused k-space multiply (particle_mesh_threaded.f90:183-192): the bundle of rho-hat is
//   read ONCE into registers; for each component c the bundle times i*K_c is transformed and
//   written to dst + c*dst_comp_stride.
// Pruning: only `ocount` values of the other axis starting at `olo` are processed, and only line
//   elements [slo, slo+scount) are stored (the inverse only needs the force box).
struct LinesArgs {
  float2 *dst; const float2 *src; const float *kern;
  int64_t kern_comp_stride, dst_comp_stride, dst_batch_stride;
  int n, px, axis, nchunk, olo, ocount, slo, scount;
};
template <int BX, bool INV, int NC>
__global__ __launch_bounds__(256) void k_fft_lines(LinesArgs a, Factors fac, const float2 *__restrict__ tw_g) {
  extern __shared__ float2 lds[];
  constexpr int L4 = BX / 2;
  const int n = a.n, px = a.px;
  float2 *A = lds, *B = A + n * BX, *tw = B + n * BX;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  const int ch = blockIdx.x % a.nchunk;
  const int64_t rest = blockIdx.x / a.nchunk;
  const int o = a.olo + (int)(rest % a.ocount);
  const int64_t b = rest / a.ocount;
  const int x0 = ch * BX;
  const int64_t plane = (int64_t)n * px;
  int64_t inb, stride;   // offsets inside one array, in float2 units
  if (a.axis == 1) { inb = (int64_t)o * plane + x0; stride = px; }
  else { inb = (int64_t)o * px + x0; stride = plane; }
  const float4 *src4 = reinterpret_cast<const float4 *>(a.src + b * n * plane + inb);
  const int64_t st4 = stride / 2;
  const int ne = n * L4;
  if (NC == 0) {
    for (int e = threadIdx.x; e < ne; e += 256) {
      const int idx = e / L4, l4 = e - idx * L4;
      float4 v = src4[idx * st4 + l4];
      if (INV) { v.y = -v.y; v.w = -v.w; }
      *reinterpret_cast<float4 *>(&A[idx * BX + 2 * l4]) = v;
    }
    __syncthreads();
    const float2 *Z = fft_lds<false>(A, B, n, BX, BX, 1, fac, tw, 1);
    float4 *dst4 = reinterpret_cast<float4 *>(a.dst + b * a.dst_batch_stride + inb);
    for (int e = threadIdx.x; e < ne; e += 256) {
      const int idx = e / L4, l4 = e - idx * L4;
      if (idx < a.slo || idx >= a.slo + a.scount) continue;
      float4 v = *reinterpret_cast<const float4 *>(&Z[idx * BX + 2 * l4]);
      if (INV) { v.y = -v.y; v.w = -v.w; }
      dst4[idx * st4 + l4] = v;
    }
  } else {
    // the bundle of rho-hat is re-read per component: the 2nd and 3rd reads hit this XCD's L2
#pragma unroll 1
    for (int comp = 0; comp < NC; comp++) {
      const float2 *k2 = reinterpret_cast<const float2 *>(a.kern + comp * a.kern_comp_stride + inb);
      for (int e = threadIdx.x; e < ne; e += 256) {
        const int idx = e / L4, l4 = e - idx * L4;
        const float4 r = src4[idx * st4 + l4];
        const float2 K = k2[idx * st4 + l4];
        // (re,im) * i*K = (-im*K, re*K); then conj for the inverse-by-forward trick
        *reinterpret_cast<float4 *>(&A[idx * BX + 2 * l4]) = make_float4(-r.y * K.x, -(r.x * K.x), -r.w * K.y, -(r.z * K.y));
      }
      __syncthreads();
      const float2 *Z = fft_lds<false>(A, B, n, BX, BX, 1, fac, tw, 1);
      float4 *dst4 = reinterpret_cast<float4 *>(a.dst + comp * a.dst_comp_stride + b * a.dst_batch_stride + inb);
      for (int e = threadIdx.x; e < ne; e += 256) {
        const int idx = e / L4, l4 = e - idx * L4;
        if (idx < a.slo || idx >= a.slo + a.scount) continue;
        float4 v = *reinterpret_cast<const float4 *>(&Z[idx * BX + 2 * l4]);
        v.y = -v.y; v.w = -v.w;
        dst4[idx * st4 + l4] = v;
      }
      __syncthreads();
    }
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
  if (n < 4 || (n & 3)) { p3m_set_error("fft: n=%d must be a multiple of 4", n); return P3M_EINVAL; }
  if (n > 1024) { p3m_set_error("fft: n=%d > 1024 not supported by the LDS line kernels", n); return P3M_EINVAL; }
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

static Factors mkfac(int nfac, const int *f) { Factors F; F.nfac = nfac; for (int i = 0; i < nfac; i++) F.f[i] = f[i]; return F; }

template <typename K> static int set_lds(K kern, size_t bytes) {
  if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return P3M_OK;
}

template <int RB> static int launch_x_fwd(p3m_ctx *c, const FftPlan &pl, float *data, int64_t rows) {
  const int n = pl.n; const size_t lds = sizeof(float2) * ((size_t)2 * RB * (n / 2 + 2) + n);
  P3M_TRY(set_lds(k_fft_x_fwd<RB>, lds));
  hipLaunchKernelGGL(k_fft_x_fwd<RB>, dim3(cdiv(rows, RB)), dim3(256), lds, c->stream, data, n, pl.px, (int)rows,
                     mkfac(pl.nfac_half, pl.fac_half), pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int RB>
static int launch_x_inv(p3m_ctx *c, const FftPlan &pl, float *data, int64_t rows, int mode, float *box, int fb, int lo, int ntile, int64_t bcs) {
  const int n = pl.n; const size_t lds = sizeof(float2) * ((size_t)2 * RB * (n / 2 + 2) + n);
  const float scale = (float)n * (float)n * (float)n;  // real(nf_tile)**3, fftw2.f90:22
  P3M_TRY(set_lds(k_fft_x_inv<RB>, lds));
  hipLaunchKernelGGL(k_fft_x_inv<RB>, dim3(cdiv(rows, RB)), dim3(256), lds, c->stream, data, n, pl.px, (int)rows,
                     mkfac(pl.nfac_half, pl.fac_half), pl.d_tw, scale, mode, box, fb, lo, ntile, bcs);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <int BX, bool INV, int NC> static int launch_lines(p3m_ctx *c, const FftPlan &pl, LinesArgs a, int batch) {
  const int n = pl.n;
  a.n = n; a.px = pl.px; a.nchunk = pl.px / BX;
  const size_t lds = sizeof(float2) * ((size_t)2 * n * BX + n);
  P3M_TRY(set_lds(k_fft_lines<BX, INV, NC>, lds));
  const int64_t blocks = (int64_t)batch * a.ocount * a.nchunk;
  hipLaunchKernelGGL((k_fft_lines<BX, INV, NC>), dim3((unsigned)blocks), dim3(256), lds, c->stream, a, mkfac(pl.nfac_full, pl.fac_full), pl.d_tw);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
template <bool INV, int NC> static int lines_any(p3m_ctx *c, const FftPlan &pl, const LinesArgs &a, int batch) {
  return pl.n <= 320 ? launch_lines<16, INV, NC>(c, pl, a, batch) : launch_lines<8, INV, NC>(c, pl, a, batch);
}
static LinesArgs full_args(const FftPlan &pl, float *dst, const float *src, int axis) {
  LinesArgs a{};
  a.dst = reinterpret_cast<float2 *>(dst); a.src = reinterpret_cast<const float2 *>(src); a.kern = nullptr;
  a.dst_batch_stride = (int64_t)pl.n * pl.n * pl.px; a.axis = axis; a.olo = 0; a.ocount = pl.n; a.slo = 0; a.scount = pl.n;
  return a;
}

int fft_x_forward(p3m_ctx *c, const FftPlan &pl, float *data, int batch) {
  const int64_t rows = (int64_t)batch * pl.n * pl.n;
  return pl.n <= 256 ? launch_x_fwd<8>(c, pl, data, rows) : launch_x_fwd<4>(c, pl, data, rows);
}
int fft_x_inverse(p3m_ctx *c, const FftPlan &pl, float *data, int batch, int mode, float *box, int fb, int lo, int ntile, int64_t bcs) {
  const int64_t rows = mode == 0 ? (int64_t)batch * pl.n * pl.n : (int64_t)batch * fb * fb;
  return pl.n <= 256 ? launch_x_inv<8>(c, pl, data, rows, mode, box, fb, lo, ntile, bcs)
                     : launch_x_inv<4>(c, pl, data, rows, mode, box, fb, lo, ntile, bcs);
}

int fft3d_forward(p3m_ctx *c, const FftPlan &pl, float *data, int batch) {
  P3M_TRY(fft_x_forward(c, pl, data, batch));
  P3M_TRY((lines_any<false, 0>(c, pl, full_args(pl, data, data, 1), batch)));
  P3M_TRY((lines_any<false, 0>(c, pl, full_args(pl, data, data, 2), batch)));
  return P3M_OK;
}

// full-size inverse (coarse mesh, probes): data <- c2r(src [* i*kern]) / n^3
int fft3d_inverse(p3m_ctx *c, const FftPlan &pl, float *data, int batch, const float *src, const float *kern) {
  LinesArgs z = full_args(pl, data, src ? src : data, 2);
  if (kern) { z.kern = kern; P3M_TRY((lines_any<true, 1>(c, pl, z, batch))); }
  else P3M_TRY((lines_any<true, 0>(c, pl, z, batch)));
  P3M_TRY((lines_any<true, 0>(c, pl, full_args(pl, data, data, 1), batch)));
  return fft_x_inverse(c, pl, data, batch, 0, nullptr, 0, 0, 1, 0);
}

// fine mesh: the three force components of `batch` tiles from rho-hat, pruned to the force box.
// work holds 3*batch arrays ([comp][tile]); box points at tile0 of component 0, bcs = component stride.
int fft_inverse3_box_z(p3m_ctx *c, const FftPlan &pl, const float *rho_hat, float *work, const float *kern3, int batch, int fb, int lo) {
  LinesArgs z = full_args(pl, work, rho_hat, 2);
  z.kern = kern3; z.kern_comp_stride = (int64_t)pl.n * pl.n * pl.px;      // one float per complex element, same pitch
  z.dst_comp_stride = (int64_t)batch * pl.n * pl.n * pl.px;
  z.slo = lo; z.scount = fb;                                              // only box planes are stored
  return lines_any<true, 3>(c, pl, z, batch);
}
int fft_inverse3_box_y(p3m_ctx *c, const FftPlan &pl, float *work, int batch, int fb, int lo) {
  LinesArgs y = full_args(pl, work, work, 1);
  y.olo = lo; y.ocount = fb; y.slo = lo; y.scount = fb;                   // only box planes, only box rows
  return lines_any<true, 0>(c, pl, y, 3 * batch);
}
int fft_inverse3_box(p3m_ctx *c, const FftPlan &pl, const float *rho_hat, float *work, const float *kern3, int batch, float *box, int fb, int lo,
                     int64_t bcs) {
  P3M_TRY(fft_inverse3_box_z(c, pl, rho_hat, work, kern3, batch, fb, lo));
  P3M_TRY(fft_inverse3_box_y(c, pl, work, batch, fb, lo));
  return fft_x_inverse(c, pl, work, 3 * batch, 1, box, fb, lo, batch, bcs);
}

// benchmark hook: one pass kernel over `batch` tiles (see p3m_hip_time_fft_pass)
int fft_single_pass(p3m_ctx *c, const FftPlan &pl, int which, float *data, float *work, const float *kern, int batch, float *box, int fb, int lo,
                    int64_t bcs) {
  switch (which) {
    case 0: return fft_x_forward(c, pl, data, batch);
    case 1: return lines_any<false, 0>(c, pl, full_args(pl, data, data, 1), batch);
    case 2: return lines_any<false, 0>(c, pl, full_args(pl, data, data, 2), batch);
    case 3: return fft_inverse3_box_z(c, pl, data, work, kern, batch, fb, lo);
    case 4: return fft_inverse3_box_y(c, pl, work, batch, fb, lo);
    case 5: return fft_x_inverse(c, pl, work, 3 * batch, 1, box, fb, lo, batch, bcs);
  }
  p3m_set_error("fft_single_pass: bad selector %d", which);
  return P3M_EINVAL;
}
