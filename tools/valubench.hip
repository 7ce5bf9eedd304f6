// Vector-issue micro-benchmark (gfx950): how many cycles one SIMD spends per wave-instruction of the kinds the short-range (PP)
// kernels are made of -- plain and packed fp32 arithmetic, the reciprocal square root, compares / selects, and a broadcast
// ds_read_b128 -- with 1, 2 and 4 wavefronts per SIMD.  Independent instructions in an unrolled loop, timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/valubench.hip -o tools/valubench.bin && tools/valubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
template <int MODE> __global__ __launch_bounds__(1024) void k(float *__restrict__ out, long long *__restrict__ cyc, int iters, float seed) {
  __shared__ float4 lds[1024];
  lds[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + 3);
  __syncthreads();
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
  const float ms = 1.0001f, cs = 1e-3f;
  unsigned ldsaddr = (unsigned)(size_t)lds + (threadIdx.x & 0) * 16;   // the same address in every lane: a broadcast
  f32x4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  unsigned i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7, i8 = 3, i9 = 1000;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {   // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms), "v"(cs));)
    } else if (MODE == 1) {   // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m), "v"(c));)
    } else if (MODE == 2) {   // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));)
    } else if (MODE == 3) {   // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));)
    } else if (MODE == 4) {   // v_rsq_f32
      REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                        "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 5) {   // v_cmp_lt_f32 (VOP3, SGPR pair) + v_cndmask_b32
      REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32 %1, %1, %9, s[20:21]\n v_cmp_lt_f32 s[22:23], %2, %8\n v_cndmask_b32 %3, %3, %9, s[22:23]\n"
                        "v_cmp_lt_f32 s[24:25], %4, %8\n v_cndmask_b32 %5, %5, %9, s[24:25]\n v_cmp_lt_f32 s[26:27], %6, %8\n v_cndmask_b32 %7, %7, %9, s[26:27]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms), "v"(cs)
                        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (MODE == 6) {   // a broadcast ds_read_b128 beside 8 v_pk_fma_f32 (the shape of a swept pair evaluation)
      REP8(asm volatile("ds_read_b128 %8, %12\n ds_read_b128 %9, %12 offset:16\n"
                        "v_pk_fma_f32 %0, %0, %10, %11\n v_pk_fma_f32 %1, %1, %10, %11\n v_pk_fma_f32 %2, %2, %10, %11\n v_pk_fma_f32 %3, %3, %10, %11\n"
                        "v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "=&v"(q0), "=&v"(q1) : "v"(m), "v"(c), "v"(ldsaddr));)
    } else if (MODE == 7) {   // broadcast ds_read_b128 alone, 8 in flight
      REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                        "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(ldsaddr));)
    } else if (MODE == 9) {   // v_cmp_lt_f32_e32 (VCC) + v_cndmask_b32_e32
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                        "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms), "v"(cs) : "vcc");)
    } else if (MODE == 10) {   // v_mul_f32
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms));)
    } else if (MODE == 11) {   // v_fma_f32 with the clamp modifier (an arithmetic 0/1 mask in one instruction)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9 clamp\n v_fma_f32 %1, %1, %8, %9 clamp\n v_fma_f32 %2, %2, %8, %9 clamp\n v_fma_f32 %3, %3, %8, %9 clamp\n"
                        "v_fma_f32 %4, %4, %8, %9 clamp\n v_fma_f32 %5, %5, %8, %9 clamp\n v_fma_f32 %6, %6, %8, %9 clamp\n v_fma_f32 %7, %7, %8, %9 clamp\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms), "v"(cs));)
    } else if (MODE == 12) {   // v_sub_u32 + v_cmp_lt_u32 (SGPR pair): the window test of a swept partner
      REP8(asm volatile("v_sub_u32 %0, %0, %8\n v_cmp_lt_u32 s[20:21], %0, %9\n v_sub_u32 %1, %1, %8\n v_cmp_lt_u32 s[22:23], %1, %9\n"
                        "v_sub_u32 %2, %2, %8\n v_cmp_lt_u32 s[24:25], %2, %9\n v_sub_u32 %3, %3, %8\n v_cmp_lt_u32 s[26:27], %3, %9\n"
                        : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i8), "v"(i9)
                        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (MODE == 13) {   // v_cmp_lt_f32 alone (SGPR pair)
      REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                        "v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms)
                        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (MODE == 14) {   // v_cndmask_b32 alone (one SGPR pair)
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                        "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms));)
    } else if (MODE == 15) {   // v_readlane_b32
      REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                        "v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :
                        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (MODE == 16) {   // v_fma_f32 with one SGPR operand (a partner held in scalar registers)
      REP8(asm volatile("v_fma_f32 %0, %0, s20, %9\n v_fma_f32 %1, %1, s21, %9\n v_fma_f32 %2, %2, s22, %9\n v_fma_f32 %3, %3, s23, %9\n"
                        "v_fma_f32 %4, %4, s24, %9\n v_fma_f32 %5, %5, s25, %9\n v_fma_f32 %6, %6, s26, %9\n v_fma_f32 %7, %7, s27, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ms), "v"(cs));)
    } else if (MODE == 8) {   // v_mov_b32
      REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                        "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(i0 + i1 + i2 + i3) + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + q0.x + q1.y + q2.z + q3.w;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int MODE> static void run(const char *name, float *o, long long *cy) {
  for (int wps = 1; wps <= 4; wps *= 2) {                // wavefronts per SIMD: workgroups of 256 * wps threads, one per CU
    const int nt = 256 * wps, nb = 256, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(nt), 0, 0, o, cy, iters, 1.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(nb * nt / 64);
    hipMemcpy(h.data(), cy, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (long long v : h) mean += (double)v; mean /= h.size();
    const double ninstr = 64.0 * iters;                  // per wavefront
    // s_memtime ticks at a fixed 100 MHz on this part; the wall time gives shader cycles at the boost clock instead
    printf("%-34s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz), counter ticks/instr %.3f\n", name, wps, ms,
           ms * 1e6 / (ninstr * wps), ms * 1e6 / (ninstr * wps) * 2.4, mean / ninstr);
  }
}
int main() {
  float *o; long long *cy; hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&cy, 256 * 16 * 8);
  run<0>("v_fma_f32", o, cy);
  run<1>("v_pk_fma_f32", o, cy);
  run<2>("v_pk_mul_f32", o, cy);
  run<3>("v_pk_add_f32", o, cy);
  run<4>("v_rsq_f32", o, cy);
  run<5>("v_cmp_lt_f32 + v_cndmask_b32", o, cy);
  run<8>("v_mov_b32", o, cy);
  run<10>("v_mul_f32", o, cy);
  run<11>("v_fma_f32 clamp", o, cy);
  run<16>("v_fma_f32 with an SGPR operand", o, cy);
  run<9>("v_cmp_lt_f32 vcc + v_cndmask vcc", o, cy);
  run<13>("v_cmp_lt_f32 (sgpr pair)", o, cy);
  run<14>("v_cndmask_b32 (sgpr pair)", o, cy);
  run<12>("v_sub_u32 + v_cmp_lt_u32", o, cy);
  run<15>("v_readlane_b32", o, cy);
  run<6>("2 ds_read_b128 bcast + 8 v_pk_fma (per 8)", o, cy);
  run<7>("ds_read_b128 broadcast", o, cy);
  return 0;
}
