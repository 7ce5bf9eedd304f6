// LDS scatter-add micro-benchmark (gfx950): ds_add_f32 against a plain read-add-write on conflict-free indices, one wavefront per
// workgroup like k_fine_deposit.  hipcc --offload-arch=gfx950 -O3 tools/ldsbench.hip -o tools/ldsbench.bin && tools/ldsbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE> __global__ __launch_bounds__(64) void k(const int *__restrict__ idx, float *__restrict__ out, int n, int iters) {
  __shared__ float row[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) row[i] = 0.f;
  __syncthreads();
  const int *p = idx + (size_t)blockIdx.x * n;
  for (int it = 0; it < iters; it++)
    for (int i = threadIdx.x; i < n; i += 64) {
      const int c = p[i]; const float w = 1.0f + 1e-3f * i;
      if (MODE == 0) { atomicAdd(&row[c], w); atomicAdd(&row[c + 1], 0.5f * w); }
      else if (MODE == 1) { float v = row[c]; row[c] = v + w; float u = row[c + 1]; row[c + 1] = u + 0.5f * w; }
      else if (MODE == 2) { atomicAdd(reinterpret_cast<int *>(&row[c]), 1); atomicAdd(reinterpret_cast<int *>(&row[c + 1]), 2); }            // ds_add_u32
      else { const int a = atomicAdd(reinterpret_cast<int *>(&row[c]), 1); atomicAdd(reinterpret_cast<int *>(&row[c + 1]), a & 1); }   // ds_add_rtn_u32
    }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 1024; i += 64) s += row[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  const int nb = 256 * 64, n = 256, iters = 64;   // 256 records per block: lanes of one instruction hit cells 8 apart (the reference's density)
  std::vector<int> h((size_t)nb * n);
  for (int b = 0; b < nb; b++) for (int i = 0; i < n; i++) h[(size_t)b * n + i] = ((i % 64) * 8 + (i / 64) * 2 + (b % 5)) % 1000;
  int *d; float *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, nb * 64 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = {"ds_add_f32    ", "read-add-write", "ds_add_u32    ", "ds_add_rtn_u32"};
  for (int mode = 0; mode < 4; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, d, o, n, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, d, o, n, iters);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(64), 0, 0, d, o, n, iters);
      else hipLaunchKernelGGL(k<3>, dim3(nb), dim3(64), 0, 0, d, o, n, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double ops = (double)nb * n * iters * 2;
      if (rep) printf("%s: %.3f ms, %.2f G lane-updates/s, %.2f clocks per wave-instruction-pair per CU-resident wave set\n", names[mode], ms,
                      ops / ms / 1e6, ms * 1e-3 * 2.1e9 * 256 / ((double)nb * (n / 64) * iters));
    }
  }
  return 0;
}
