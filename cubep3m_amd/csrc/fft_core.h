// fft_core.h -- in-LDS mixed-radix Stockham machinery shared by the FFT kernels (see fft.hip).
#pragma once
// mNs[s] / mNb[s]: division magics (see fdiv) of stage s's Ns = f[0]*...*f[s-1] and nb = n/f[s], filled by the host
struct Factors { int nfac; int f[12]; unsigned mNs[12], mNb[12]; };

// Exact x / d for x*d < 2^32 as one mulhi (the integer divisions of the index arithmetic otherwise cost more
// issue slots than the butterflies): m = floor((2^32-1)/d) + 1; d == 1 wraps to m == 0 and is special-cased.
struct fdiv_t { unsigned m; int d; };
__host__ __device__ __forceinline__ unsigned fdiv_magic(int d) { return 0xFFFFFFFFu / (unsigned)d + 1u; }
__device__ __forceinline__ fdiv_t mk_fdiv(int d) { fdiv_t f; f.m = fdiv_magic(d); f.d = d; return f; }
__device__ __forceinline__ int fdiv(int x, fdiv_t f) { return f.d == 1 ? x : (int)__umulhi((unsigned)x, f.m); }

// geometry of the coarse slab decomposition's x-line rows (group.hip: row_geom): nl local ranks of s planes each, rpp rows per plane,
// rows of rp floats (nc real cells + pad); cubes of ncn^3 cells, nd per dimension
#define P3M_MAX_LOCAL 64        // local ranks one batched launch can address (per-rank pointers in kernel arguments)
struct RowGeom { int nl, s, nc, ncn, nd, rpp, rp; unsigned m_rpp, m_s, m_ncn; };
struct RankPtrs { float *p[P3M_MAX_LOCAL]; };

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

// The butterflies work on a native 2-vector (re, im) so that every complex add is ONE v_pk_add_f32 and every
// complex product TWO packed instructions (swizzles and sign flips ride on op_sel / neg modifiers); written with
// HIP's float2 struct the compiler pairs up unrelated scalars and pays for it in v_mov shuffles.
typedef float c32 __attribute__((ext_vector_type(2), may_alias));
__device__ __forceinline__ c32 vmul(c32 a, c32 b) { return a.yx * (c32){-b.y, b.y} + a * b.xx; }
// a * (c - i s) with compile-time c, s: four scalar VOP2 operations carrying the constants as literals (the packed form
// needs them in registers, 4 VGPRs per twiddle, hoisted out of every loop)
__device__ __forceinline__ c32 vmulk(c32 a, float c, float s) { return (c32){a.x * c + a.y * s, a.y * c - a.x * s}; }
__device__ __forceinline__ c32 vmi(c32 a) { return (c32){a.y, -a.x}; }   // * -i
// The same with the swizzles and sign flips spelled out as op_sel / neg modifiers of ONE or TWO packed instructions: from the vector
// expressions above the compiler builds the swapped / negated operand first (v_xor + v_mov per product or +-i sum).
//   vmulr: a * b, b in registers (twiddles from LDS or memory; compile-time constants stay with vmul / vmulk)
__device__ __forceinline__ c32 vmulr(c32 a, c32 b) {
  c32 t;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));                                           // a * b.xx
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(a), "v"(b));       // + a.yx * (-b.y, b.y)
  return t;
}
//   a - i b and a + i b
__device__ __forceinline__ c32 vsubi(c32 a, c32 b) { c32 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // (a.x + b.y, a.y - b.x)
__device__ __forceinline__ c32 vaddi(c32 a, c32 b) { c32 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }   // (a.x - b.y, a.y + b.x)
// The Hermitian unscramble of the real transforms, one packed instruction per line with the swizzles and signs as op_sel / neg modifiers
// (written on .x / .y the compiler forms sum and difference vectors and then moves their halves about: 12 instructions against 5):
//   x[m], x[h-m], t = exp(-2 pi i m / n)  ->  conj(e + i o),  e = (x[m] + conj(x[h-m])) , o = (x[m] - conj(x[h-m])) conj(t)
__device__ __forceinline__ c32 c2r_pre(c32 xk, c32 xm, c32 t) {
  c32 e, d, o, v;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(e) : "v"(xk), "v"(xm));                                     // (xk.x + xm.x, xk.y - xm.y)
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(xk), "v"(xm));                                     // (xk.x - xm.x, xk.y + xm.y)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(o) : "v"(d), "v"(t));                                    // d * t.xx
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "+v"(o) : "v"(d), "v"(t));   // + d.yx * (t.y, -t.y)
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(v) : "v"(e), "v"(o));   // (e.x - o.y, -e.y - o.x)
  return v;
}

// forward DFT (sign -1) of R values held in registers
template <int R> __device__ __forceinline__ void dft(c32 (&v)[R]);

template <> __device__ __forceinline__ void dft<2>(c32 (&v)[2]) {
  const c32 a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b;
}
template <> __device__ __forceinline__ void dft<4>(c32 (&v)[4]) {
  const c32 a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], d = v[1] - v[3];
  v[0] = a + c; v[1] = vsubi(b, d); v[2] = a - c; v[3] = vaddi(b, d);
}
template <> __device__ __forceinline__ void dft<8>(c32 (&v)[8]) {
  const float r = 0.70710678118654752440f;
  c32 a[4], b[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { a[k] = v[k] + v[k + 4]; b[k] = v[k] - v[k + 4]; }
  b[1] = r * (b[1] + vmi(b[1]));    // * (1-i)/sqrt2
  b[2] = vmi(b[2]);                 // * -i
  b[3] = r * (vmi(b[3]) - b[3]);    // * (-1-i)/sqrt2
  dft<4>(a); dft<4>(b);
#pragma unroll
  for (int q = 0; q < 4; q++) { v[2 * q] = a[q]; v[2 * q + 1] = b[q]; }
}
// radix 16 = 4 x 4: Y_{n1} = DFT4 over n2 of v[n1+4*n2]; twiddle W16^{n1*k2}; X[4*k1+k2] = DFT4 over n1
template <> __device__ __forceinline__ void dft<16>(c32 (&v)[16]) {
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r = 0.70710678118654752440f;
  c32 y[4][4];
#pragma unroll
  for (int n1 = 0; n1 < 4; n1++) {
    c32 t[4] = {v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]};
    dft<4>(t);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) y[n1][k2] = t[k2];
  }
  // W16^m = exp(-2 pi i m/16): m = n1*k2
  const c32 w1 = {c1, -s1}, w3 = {s1, -c1}, w9 = {-c1, s1};
  y[1][1] = vmul(y[1][1], w1); y[1][2] = r * (y[1][2] + vmi(y[1][2])); y[1][3] = vmul(y[1][3], w3);
  y[2][1] = r * (y[2][1] + vmi(y[2][1])); y[2][2] = vmi(y[2][2]); y[2][3] = r * (vmi(y[2][3]) - y[2][3]);
  y[3][1] = vmul(y[3][1], w3); y[3][2] = r * (vmi(y[3][2]) - y[3][2]); y[3][3] = vmul(y[3][3], w9);
#pragma unroll
  for (int k2 = 0; k2 < 4; k2++) {
    c32 t[4] = {y[0][k2], y[1][k2], y[2][k2], y[3][k2]};
    dft<4>(t);
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) v[4 * k1 + k2] = t[k1];
  }
}
template <int R> __device__ __forceinline__ void dft_odd(c32 (&v)[R]) {
  constexpr int H = (R - 1) / 2;
  constexpr TrigTab<R> tab{};
  c32 t[H + 1], u[H + 1];
#pragma unroll
  for (int p = 1; p <= H; p++) { t[p] = v[p] + v[R - p]; u[p] = v[p] - v[R - p]; }
  const c32 v0 = v[0];
  c32 s0 = v[0];
#pragma unroll
  for (int p = 1; p <= H; p++) s0 = s0 + t[p];
  v[0] = s0;
#pragma unroll
  for (int a = 1; a <= H; a++) {
    c32 A = v0, B = {0.f, 0.f};
#pragma unroll
    for (int p = 1; p <= H; p++) {
      const float cc = tab.c[(a * p) % R], ss = tab.s[(a * p) % R];
      A = A + cc * t[p]; B = B + ss * u[p];
    }
    v[a] = vsubi(A, B);       // A - iB
    v[R - a] = vaddi(A, B);   // A + iB
  }
}
template <> __device__ __forceinline__ void dft<3>(c32 (&v)[3]) { dft_odd<3>(v); }
template <> __device__ __forceinline__ void dft<5>(c32 (&v)[5]) { dft_odd<5>(v); }
template <> __device__ __forceinline__ void dft<7>(c32 (&v)[7]) { dft_odd<7>(v); }
template <> __device__ __forceinline__ void dft<11>(c32 (&v)[11]) { dft_odd<11>(v); }
template <> __device__ __forceinline__ void dft<13>(c32 (&v)[13]) { dft_odd<13>(v); }
template <> __device__ __forceinline__ void dft<17>(c32 (&v)[17]) { dft_odd<17>(v); }
template <> __device__ __forceinline__ void dft<19>(c32 (&v)[19]) { dft_odd<19>(v); }

// composite radix R = P*Q in registers: input j = Q*p + q, output k = kp + P*kq (one Cooley-Tukey step, all indices and
// the inner twiddles W_R^{q*kp} compile-time)
template <int P, int Q> __device__ __forceinline__ void dft_pq(c32 (&v)[P * Q]) {
  constexpr int R = P * Q;
  constexpr TrigTab<R> tab{};
  c32 y[Q][P];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    c32 t[P];
#pragma unroll
    for (int p = 0; p < P; p++) t[p] = v[Q * p + q];
    dft<P>(t);
#pragma unroll
    for (int kp = 0; kp < P; kp++) {
      const int m = (q * kp) % R;
      y[q][kp] = (m == 0) ? t[kp] : vmulk(t[kp], tab.c[m], tab.s[m]);
    }
  }
#pragma unroll
  for (int kp = 0; kp < P; kp++) {
    c32 u[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) u[q] = y[q][kp];
    dft<Q>(u);
#pragma unroll
    for (int kq = 0; kq < Q; kq++) v[kp + P * kq] = u[kq];
  }
}
template <> __device__ __forceinline__ void dft<6>(c32 (&v)[6]) { dft_pq<2, 3>(v); }
template <> __device__ __forceinline__ void dft<10>(c32 (&v)[10]) { dft_pq<2, 5>(v); }
template <> __device__ __forceinline__ void dft<12>(c32 (&v)[12]) { dft_pq<4, 3>(v); }
template <> __device__ __forceinline__ void dft<22>(c32 (&v)[22]) { dft_pq<2, 11>(v); }
template <> __device__ __forceinline__ void dft<24>(c32 (&v)[24]) { dft_pq<8, 3>(v); }
template <> __device__ __forceinline__ void dft<26>(c32 (&v)[26]) { dft_pq<2, 13>(v); }
template <> __device__ __forceinline__ void dft<14>(c32 (&v)[14]) { dft_pq<2, 7>(v); }
template <> __device__ __forceinline__ void dft<20>(c32 (&v)[20]) { dft_pq<4, 5>(v); }
template <> __device__ __forceinline__ void dft<28>(c32 (&v)[28]) { dft_pq<4, 7>(v); }
template <> __device__ __forceinline__ void dft<32>(c32 (&v)[32]) { dft_pq<4, 8>(v); }

// One radix-R Stockham stage on `nl` lines of length n held in LDS.
// element (idx,line) lives at idx*sI + line*sL.  ROWS: lanes run along the line (x pass);
// otherwise lanes run across lines (strided passes).  tw[q*twm] = exp(-2 pi i q / n).
// The position k = j mod Ns inside the sub-transform needs no division in the first stage (Ns = 1,
// k = 0) nor in the last (Ns = n/R > j, k = j); NL > 0 makes the line count a compile-time constant.
template <int R, bool ROWS, int NL>
__device__ __forceinline__ void fft_stage(const float2 *__restrict__ in, float2 *__restrict__ out, int n, int Ns, int nl_rt,
                                          int sI, int sL, const float2 *__restrict__ tw, int twm, fdiv_t dNs, fdiv_t dNb, fdiv_t dNl) {
  const int nl = NL > 0 ? NL : nl_rt;
  const int nb = n / R, ntask = nb * nl;
  const int tstep = (n / (Ns * R)) * twm;
  const bool last = (Ns == nb);
  // addresses are base + m*step with uniform steps (scalar multiplies); per-lane products go through the
  // full-rate 24-bit multiplier (v_mul_lo_u32 is quarter rate and the indices are tiny)
  const int step_in = nb * sI, step_out = Ns * sI;
  for (int task = threadIdx.x; task < ntask; task += blockDim.x) {
    int j, line;
    if (ROWS) { line = fdiv(task, dNb); j = task - __mul24(line, nb); }
    else if (NL > 0) { j = task / NL; line = task - j * NL; }
    else { j = fdiv(task, dNl); line = task - __mul24(j, nl); }
    const int k = (Ns == 1) ? 0 : (last ? j : j - __mul24(fdiv(j, dNs), Ns));
    const c32 *pin = reinterpret_cast<const c32 *>(in) + __mul24(line, sL) + __mul24(j, sI);
    c32 v[R];
#pragma unroll
    for (int m = 0; m < R; m++) v[m] = pin[m * step_in];
    if (Ns > 1) {
      const c32 *twv = reinterpret_cast<const c32 *>(tw);
      const int ts = __mul24(tstep, k);
#pragma unroll
      for (int m = 1; m < R; m++) v[m] = vmulr(v[m], twv[__mul24(m, ts)]);   // 24-bit multiply: full rate
    }
    dft<R>(v);
    const int j0 = __mul24(j - k, R) + k;
    c32 *pout = reinterpret_cast<c32 *>(out) + __mul24(line, sL) + __mul24(j0, sI);
#pragma unroll
    for (int m = 0; m < R; m++) pout[m * step_out] = v[m];
  }
}

// RSET selects which odd radices are compiled in (their butterflies set the kernel's VGPR count):
// 0: 2,3,4,5,7,8   1: + 11,13   2: + 17,19
template <bool ROWS, int RSET, int NL = 0>
__device__ __forceinline__ float2 *fft_lds(float2 *A, float2 *B, int n, int nl, int sI, int sL, const Factors &fac,
                                           const float2 *tw, int twm) {
  int Ns = 1;
  float2 *in = A, *out = B;
#ifdef P3M_ABLATE_NOFFT   // timing-only build: global<->LDS traffic without the butterflies (outputs are wrong)
  return in;
#endif
  const fdiv_t dNl = mk_fdiv(NL > 0 ? NL : nl);
  for (int s = 0; s < fac.nfac; s++) {
    const int R = fac.f[s];
    fdiv_t dNs, dNb; dNs.m = fac.mNs[s]; dNs.d = Ns; dNb.m = fac.mNb[s]; dNb.d = n / R;
    switch (R) {
      case 2: fft_stage<2, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      case 3: fft_stage<3, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      case 4: fft_stage<4, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      case 5: fft_stage<5, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      case 7: fft_stage<7, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      case 8: fft_stage<8, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      case 16: fft_stage<16, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl); break;
      default:
        if constexpr (RSET >= 1) {
          if (R == 11) fft_stage<11, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl);
          else if (R == 13) fft_stage<13, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl);
        }
        if constexpr (RSET >= 2) {
          if (R == 17) fft_stage<17, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl);
          else if (R == 19) fft_stage<19, ROWS, NL>(in, out, n, Ns, nl, sI, sL, tw, twm, dNs, dNb, dNl);
        }
        break;
    }
    Ns *= R;
    __syncthreads();
    float2 *t = in; in = out; out = t;
  }
  return in;
}

