// l2bench: does a producer -> consumer hand-off through the XCD's own L2 (inside one persistent launch) save the HBM round
// trip that two launches pay?  Model of the y-pass -> x-pass hand-off of the fine-mesh inverse transform:
//   item  = one z-plane of one force component, [NCH chunks][NR rows][128 B]           (1.29 MB at NCH = 18, NR = 560)
//   A task (one per chunk): streams the chunk's NR*128 B from `in` and stores it to the XCD's scratch slot  (the y lines)
//   B task (NR/RB per item): gathers RB rows (NCH segments of 128 B each) from the slot and streams them to `out` (the x rows)
// Variants: (0) one launch, tasks drawn per XCD from a queue, slots in a 2-deep ring per XCD (2.6 MB of the 4 MB L2);
// (1) two launches through a full-size intermediate array; (2) one launch that reads `in` with B's gather directly (floor).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/l2bench.bin tools/l2bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NCH = 18, NR = 560, SEG = 8;          // 8 f4 = 128 B per (chunk,row)
constexpr int ITEM = NCH * NR * SEG;                // f4 per item
constexpr int RB = 16, NBT = NR / RB;               // rows per B task, B tasks per item
constexpr int TB = 256;

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ f4 load_sc1(const f4 *p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void load_sc1_n(const f4 *(&p)[N], f4 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(p[i]) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A: chunk c of item -> dst (same layout)
__device__ __forceinline__ void task_a(const f4 *__restrict__ in_item, f4 *__restrict__ dst_item, int c) {
  const f4 *s = in_item + (size_t)c * NR * SEG; f4 *d = dst_item + (size_t)c * NR * SEG;
  for (int i = threadIdx.x; i < NR * SEG; i += 4 * TB) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = i + u * TB; v[u] = k < NR * SEG ? __builtin_nontemporal_load(s + k) : f4{0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int k = i + u * TB; if (k < NR * SEG) d[k] = v[u]; }
  }
}
// B: rows r0..r0+RB-1 of src_item -> out rows [row][NCH*SEG]
template <bool SC1> __device__ __forceinline__ void task_b(const f4 *__restrict__ src_item, f4 *__restrict__ out_item, int r0) {
  constexpr int PER = RB * NCH * SEG;   // 2304 f4
  for (int i = threadIdx.x; i < PER; i += 3 * TB) {
    const f4 *p[3]; f4 v[3]; int o[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int k = min(i + u * TB, PER - 1), seg = k % SEG, c = (k / SEG) % NCH, r = k / (SEG * NCH);
      p[u] = src_item + ((size_t)c * NR + (r0 + r)) * SEG + seg; o[u] = (r0 + r) * (NCH * SEG) + c * SEG + seg;
    }
    if (SC1) load_sc1_n<3>(p, v);
    else {
#pragma unroll
      for (int u = 0; u < 3; u++) v[u] = __builtin_nontemporal_load(p[u]);
    }
#pragma unroll
    for (int u = 0; u < 3; u++) if (i + u * TB < PER) __builtin_nontemporal_store(v[u], out_item + o[u]);
  }
}

struct Ctl { int head; int pad[31]; int done_a[480]; int done_b[480]; };   // per XCD

__global__ __launch_bounds__(TB) void k_fused(const f4 *__restrict__ in, f4 *__restrict__ out, f4 *__restrict__ scratch, Ctl *ctl, int nitem, int ring, int stat) {
  __shared__ int s_t;
  const int x = xcc_id();
  Ctl *C = ctl + x;
  const int nloc = (nitem - x + 7) / 8;   // items x, x+8, ...
  if (nloc <= 0) return;
  f4 *slots = scratch + (size_t)x * ring * ITEM;
  const int g0 = NCH, gm = NCH + NBT, total = g0 + (nloc - 1) * gm + NBT;
  // ONE thread-0 region per iteration, between two barriers (signal the finished task, draw the next, wait for what it needs):
  // with thread-0 regions at both ends of the body the compiler rotates them across the back edge into a loop that lane 0
  // leaves alone, and the rest of wave 0 runs ahead through the barriers
  int ptype = -1, pj = 0;
  __shared__ int s_w;
  if (threadIdx.x == 0) s_w = stat ? atomicAdd(&C->pad[1], 1) : 0;   // static: my index among the workgroups of this XCD
  __syncthreads();
  const int nw = gridDim.x / 8;
  int mine = __builtin_amdgcn_readfirstlane(s_w);
  for (;;) {
    __syncthreads();                                     // every wave ran s_waitcnt vmcnt(0) after its part of the last task
    if (threadIdx.x == 0) {
      if (ptype >= 0) __hip_atomic_fetch_add(ptype == 0 ? &C->done_a[pj] : &C->done_b[pj], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int t;
      if (stat) { t = mine; mine += nw; } else t = atomicAdd(&C->head, 1);
      if (t < total) {
        int type, j;
        if (t < g0) { type = 0; j = 0; }
        else {
          const int q = (t - g0) / gm, r = (t - g0) % gm;
          if (q < nloc - 1) { if (r < NCH) { type = 0; j = q + 1; } else { type = 1; j = q; } }
          else { type = 1; j = nloc - 1; }
        }
        int spins = 0;   // bounded: a protocol error must not hang the device
        if (type == 0 && j >= ring) while (__hip_atomic_load(&C->done_b[j - ring], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NBT && ++spins < (1 << 12)) __builtin_amdgcn_s_sleep(1);
        if (type == 1) while (__hip_atomic_load(&C->done_a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NCH && ++spins < (1 << 12)) __builtin_amdgcn_s_sleep(1);
        if (spins >= (1 << 12)) atomicAdd(&C->pad[0], 1);
      }
      s_t = t;
    }
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(s_t);
    if (t >= total) break;
    int type, j, sub;
    if (t < g0) { type = 0; j = 0; sub = t; }
    else {
      const int q = (t - g0) / gm, r = (t - g0) % gm;
      if (q < nloc - 1) { if (r < NCH) { type = 0; j = q + 1; sub = r; } else { type = 1; j = q; sub = r - NCH; } }
      else { type = 1; j = nloc - 1; sub = r; }
    }
    const int item = x + 8 * j;
    f4 *slot = slots + (size_t)(j % ring) * ITEM;
    if (type == 0) task_a(in + (size_t)item * ITEM, slot, sub);
    else task_b<true>(slot, out + (size_t)item * ITEM, sub * RB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ptype = type; pj = j;
    if (stat && threadIdx.x != 0) mine += nw;
  }
}
__global__ __launch_bounds__(TB) void k_a_all(const f4 *__restrict__ in, f4 *__restrict__ mid, int nitem) {
  for (int t = blockIdx.x; t < nitem * NCH; t += gridDim.x) task_a(in + (size_t)(t / NCH) * ITEM, mid + (size_t)(t / NCH) * ITEM, t % NCH);
}
__global__ __launch_bounds__(TB) void k_b_all(const f4 *__restrict__ mid, f4 *__restrict__ out, int nitem) {
  for (int t = blockIdx.x; t < nitem * NBT; t += gridDim.x) task_b<false>(mid + (size_t)(t / NBT) * ITEM, out + (size_t)(t / NBT) * ITEM, (t % NBT) * RB);
}
__global__ void k_census(int *xcc_of_block) { if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = xcc_id(); }

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int nitem = argc > 1 ? atoi(argv[1]) : 546;   // 546 items of 1.29 MB = 705 MB
  const int wgs = argc > 2 ? atoi(argv[2]) : 512;
  const int ring = argc > 3 ? atoi(argv[3]) : 2;
  const int stat = argc > 4 ? atoi(argv[4]) : 0;
  const size_t n = (size_t)nitem * ITEM;
  f4 *in, *out, *mid, *scratch; Ctl *ctl; int *cen;
  CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&mid, n * 16));
  CK(hipMalloc(&scratch, (size_t)8 * ring * ITEM * 16)); CK(hipMalloc(&ctl, 8 * sizeof(Ctl))); CK(hipMalloc(&cen, 4096 * 4));
  std::vector<float> h(n * 4);
  for (size_t i = 0; i < n * 4; i++) h[i] = (float)(i % 1000003) * 0.5f;
  CK(hipMemcpy(in, h.data(), n * 16, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_census, dim3(1024), dim3(64), 0, 0, cen);
  std::vector<int> hc(1024); CK(hipMemcpy(hc.data(), cen, 4096, hipMemcpyDeviceToHost));
  int bad = 0; for (int b = 8; b < 1024; b++) if (hc[b] != hc[b % 8]) bad++;
  printf("census: xcc of blocks 0..7 = %d %d %d %d %d %d %d %d; blocks b with xcc(b) != xcc(b%%8): %d of 1016\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7], bad);
  auto check = [&](const char *name) {
    std::vector<float> o(n * 4); CK(hipMemcpy(o.data(), out, n * 16, hipMemcpyDeviceToHost));
    size_t err = 0;
    for (int it = 0; it < nitem; it++) for (int r = 0; r < NR; r++) for (int c = 0; c < NCH; c++) for (int e = 0; e < SEG * 4; e++) {
      const size_t src = ((size_t)it * ITEM + ((size_t)c * NR + r) * SEG) * 4 + e, dst = ((size_t)it * ITEM + (size_t)r * NCH * SEG + c * SEG) * 4 + e;
      if (o[dst] != h[src]) err++;
    }
    printf("%-28s check: %zu wrong floats of %zu\n", name, err, n * 4);
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  const int reps = 20;
  // (1) two launches
  for (int w = 0; w < 2; w++) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) { hipLaunchKernelGGL(k_a_all, dim3(2048), dim3(TB), 0, 0, in, mid, nitem); hipLaunchKernelGGL(k_b_all, dim3(2048), dim3(TB), 0, 0, mid, out, nitem); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("two launches through HBM    : %.3f ms  (%.2f GB in + out, %.2f TB/s on in+out)\n", ms / reps, 2 * n * 16 / 1e9, 2 * n * 16 / (ms / reps) / 1e9);
  CK(hipMemset(out, 0, n * 16));
  hipLaunchKernelGGL(k_a_all, dim3(2048), dim3(TB), 0, 0, in, mid, nitem); hipLaunchKernelGGL(k_b_all, dim3(2048), dim3(TB), 0, 0, mid, out, nitem);
  check("two launches");
  // (2) floor: B straight from `in`
  for (int w = 0; w < 2; w++) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_b_all, dim3(2048), dim3(TB), 0, 0, in, out, nitem);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("one gather pass (floor)     : %.3f ms  (%.2f TB/s)\n", ms / reps, 2 * n * 16 / (ms / reps) / 1e9);
  // (0) fused through the XCD's L2
  CK(hipMemset(out, 0, n * 16));
  for (int w = 0; w < 2; w++) {
    float tot = 0;
    for (int i = 0; i < reps; i++) {
      CK(hipMemsetAsync(ctl, 0, 8 * sizeof(Ctl)));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_fused, dim3(wgs), dim3(TB), 0, 0, in, out, scratch, ctl, nitem, ring, stat);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
    }
    ms = tot;
  }
  CK(hipGetLastError());
  { std::vector<Ctl> hctl(8); CK(hipMemcpy(hctl.data(), ctl, 8 * sizeof(Ctl), hipMemcpyDeviceToHost));
    for (int x = 0; x < 8; x++) printf("  xcd %d: tasks drawn %d, timeouts %d, done_a[0] %d done_b[0] %d\n", x, hctl[x].head, hctl[x].pad[0], hctl[x].done_a[0], hctl[x].done_b[0]); }
  printf("fused, %4d WGs, ring %d %s: %.3f ms  (%.2f TB/s on in+out)\n", wgs, ring, stat ? "static " : "dynamic", ms / reps, 2 * n * 16 / (ms / reps) / 1e9);
  check("fused through L2");
  return 0;
}
