// mallbench.hip -- measurement tool (not product): does a producer->consumer hand-off between two kernels through a
// small scratch ring stay in the 256 MiB Infinity Cache, i.e. cost no HBM time?  Models the FFT pass pairs
// (x_fwd -> y_fwd per z-slab, y_inv -> x_inv per z-slab): total = 705 MB in, 705 MB out.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mallbench.bin tools/mallbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_copy_nt(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f4 v = __builtin_nontemporal_load(s + i);
    __builtin_nontemporal_store(v, d + i);
  }
}
// nt load from HBM, plain store to scratch
__global__ __launch_bounds__(256) void k_copy_in(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d[i] = __builtin_nontemporal_load(s + i);
}
// plain load from scratch, nt store to HBM
__global__ __launch_bounds__(256) void k_copy_out(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(s[i], d + i);
}

int main(int argc, char **argv) {
  const size_t total = (size_t)705 << 20;   // bytes per array
  f4 *src, *dst, *mid, *ring;
  CK(hipMalloc(&src, total)); CK(hipMalloc(&dst, total)); CK(hipMalloc(&mid, total)); CK(hipMalloc(&ring, (size_t)512 << 20));
  CK(hipMemset(src, 1, total)); CK(hipMemset(dst, 0, total)); CK(hipMemset(mid, 0, total)); CK(hipMemset(ring, 0, (size_t)512 << 20));
  hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
  hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
  const int grid = 256 * 8, reps = 10;
  auto timeit = [&](const char *name, auto &&body) {
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < reps + 2; r++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, st[0]));
      body();
      CK(hipEventRecord(ej, st[1])); CK(hipStreamWaitEvent(st[0], ej, 0));
      CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-58s avg %.3f ms  best %.3f ms  (%.2f TB/s of the 1.41 GB that must cross HBM)\n", name, sum / reps, best, 2.0 * total / (sum / reps * 1e-3) / 1e12);
  };
  timeit("one launch: copy 705 MB -> 705 MB", [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st[0], src, dst, total / 16); });
  timeit("one launch: nt copy", [&] { hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, st[0], src, dst, total / 16); });
  timeit("two launches through a FULL intermediate array", [&] {
    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st[0], src, mid, total / 16);
    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st[0], mid, dst, total / 16);
  });
  const int slabs_mb[] = {5, 10, 20, 40, 80, 160};
  for (int mb : slabs_mb) {
    const size_t sb = (size_t)mb << 20, ns = (total + sb - 1) / sb;
    char name[128];
    for (int variant = 0; variant < 4; variant++) {
      // 0: one stream, ring of 2 slots; 1: two streams alternating, ring of 2; 2: one stream, in place in a full intermediate; 3: one stream ring, nt on the HBM side
      snprintf(name, sizeof name, "slabs of %3d MB, %s", mb, variant == 0 ? "1 stream, ring(2)" : variant == 1 ? "2 streams, ring(2)" : variant == 2 ? "1 stream, full-size intermediate" : "1 stream, ring(2), nt on the HBM side");
      timeit(name, [&] {
        if (variant == 1) { CK(hipEventRecord(ej, st[0])); CK(hipStreamWaitEvent(st[1], ej, 0)); }
        for (size_t s = 0; s < ns; s++) {
          const size_t off = s * sb, len = (off + sb <= total ? sb : total - off);
          hipStream_t q = variant == 1 ? st[s & 1] : st[0];
          f4 *m = variant == 2 ? (f4 *)((char *)mid + off) : (f4 *)((char *)ring + (s & 1) * sb);
          if (variant == 3) {
            hipLaunchKernelGGL(k_copy_in, dim3(grid), dim3(256), 0, q, (const f4 *)((const char *)src + off), m, len / 16);
            hipLaunchKernelGGL(k_copy_out, dim3(grid), dim3(256), 0, q, (const f4 *)m, (f4 *)((char *)dst + off), len / 16);
          } else {
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, q, (const f4 *)((const char *)src + off), m, len / 16);
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, q, (const f4 *)m, (f4 *)((char *)dst + off), len / 16);
          }
        }
      });
    }
  }
  // the pure on-die rate: copy inside a 40 MB + 40 MB working set, repeated
  {
    const size_t sb = (size_t)40 << 20;
    timeit("18 x (40 MB ring slot 0 -> slot 1): on-die only", [&] {
      for (int s = 0; s < 18; s++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, st[0], (const f4 *)ring, (f4 *)((char *)ring + sb), sb / 16);
    });
  }
  return 0;
}
