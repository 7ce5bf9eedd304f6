// scan.hip -- in-place exclusive prefix sum of int32 (cell offsets of the counting sort that
// replaces link_list.f90's hoc/ll chains, and stream compaction for delete_particles.f90).
// Three-kernel scheme (block sums -> recursive scan of the sums -> apply); HBM-bound:
// 2 reads + 1 write of the array.
#include "p3m_internal.h"
#include <algorithm>

#define SCAN_T 256
#define SCAN_I 16
#define SCAN_CH (SCAN_T * SCAN_I)

__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o, 64); if ((threadIdx.x & 63) >= o) v += t; }
  return v;
}
// returns the exclusive prefix of `v` within the block and the block total in *total
__device__ __forceinline__ int block_excl_scan(int v, int *total) {
  __shared__ int wsum[SCAN_T / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_T / 64; i++) { if (i < w) off += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const int *__restrict__ d, int64_t n, int *__restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_CH + (int64_t)threadIdx.x * SCAN_I;
  int s = 0;
  if (base + SCAN_I <= n) {
    const int4 *p = reinterpret_cast<const int4 *>(d + base);
#pragma unroll
    for (int i = 0; i < SCAN_I / 4; i++) { int4 q = p[i]; s += q.x + q.y + q.z + q.w; }
  } else {
    for (int i = 0; i < SCAN_I; i++) if (base + i < n) s += d[base + i];
  }
  int tot; (void)block_excl_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_apply(int *__restrict__ d, int64_t n, const int *__restrict__ sums, int *__restrict__ total_out) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_CH + (int64_t)threadIdx.x * SCAN_I;
  int v[SCAN_I];
  const bool full = base + SCAN_I <= n;
  if (full) {
    const int4 *p = reinterpret_cast<const int4 *>(d + base);
#pragma unroll
    for (int i = 0; i < SCAN_I / 4; i++) { int4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) v[i] = (base + i < n) ? d[base + i] : 0;
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_I; i++) s += v[i];
  int tot;
  int run = block_excl_scan(s, &tot) + (sums ? sums[blockIdx.x] : 0);
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = tot + (sums ? sums[blockIdx.x] : 0);
#pragma unroll
  for (int i = 0; i < SCAN_I; i++) { int t = v[i]; v[i] = run; run += t; }
  if (full) {
    int4 *p = reinterpret_cast<int4 *>(d + base);
#pragma unroll
    for (int i = 0; i < SCAN_I / 4; i++) p[i] = make_int4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) if (base + i < n) d[base + i] = v[i];
  }
}

// ------------------------------------------------------------------ one-launch scan (decoupled look-back)
// The three-kernel scheme costs three launches per scan and a step runs two scans per rank (x-row counts, survivor counts): 48 launches of
// ~5 us per step on 8 ranks.  Here every block takes a ticket (its chunk: tickets are handed out in the order blocks START, so a block
// only ever waits for blocks that are already running), publishes its aggregate, walks back over its predecessors' words until it meets an
// inclusive prefix, and publishes its own.  A word = value | flag << 32 | epoch << 34; the epoch (one per launch, a kernel argument)
// makes every earlier launch's words read as "nothing yet", so nothing is cleared between scans; the holder of the last ticket resets
// the ticket counter (every ticket of this launch has been taken by then).
#define SCAN_AGG 1ull
#define SCAN_PRE 2ull
__global__ __launch_bounds__(SCAN_T) void k_scan_lookback(int *__restrict__ d, int64_t n, unsigned long long *__restrict__ state, int *__restrict__ ticket,
                                                          unsigned epoch, int *__restrict__ total_out) {
  __shared__ int bid_s, prefix_s;
  if (threadIdx.x == 0) { bid_s = atomicAdd(ticket, 1); if (bid_s == (int)gridDim.x - 1) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __syncthreads();
  const int bid = bid_s;
  const int64_t base = (int64_t)bid * SCAN_CH + (int64_t)threadIdx.x * SCAN_I;
  int v[SCAN_I];
  const bool full = base + SCAN_I <= n;
  if (full) {
    const int4 *p = reinterpret_cast<const int4 *>(d + base);
#pragma unroll
    for (int i = 0; i < SCAN_I / 4; i++) { int4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) v[i] = (base + i < n) ? d[base + i] : 0;
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_I; i++) s += v[i];
  int tot;
  int run = block_excl_scan(s, &tot);
  if (threadIdx.x == 0) {
    const unsigned long long ep = (unsigned long long)epoch << 34;
    int prefix = 0;
    if (bid > 0) {
      __hip_atomic_store(state + bid, ep | (SCAN_AGG << 32) | (unsigned)tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      for (int j = bid - 1; j >= 0;) {
        const unsigned long long w = __hip_atomic_load(state + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((w >> 34) != epoch) { __builtin_amdgcn_s_sleep(1); continue; }   // nothing from block j yet
        prefix += (int)(unsigned)w;
        if ((w >> 32) & SCAN_PRE) break;
        j--;
      }
    }
    __hip_atomic_store(state + bid, ep | (SCAN_PRE << 32) | (unsigned)(prefix + tot), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    prefix_s = prefix;
    if (total_out && bid == (int)gridDim.x - 1) *total_out = prefix + tot;
  }
  __syncthreads();
  run += prefix_s;
#pragma unroll
  for (int i = 0; i < SCAN_I; i++) { int t = v[i]; v[i] = run; run += t; }
  if (full) {
    int4 *p = reinterpret_cast<int4 *>(d + base);
#pragma unroll
    for (int i = 0; i < SCAN_I / 4; i++) p[i] = make_int4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_I; i++) if (base + i < n) d[base + i] = v[i];
  }
}

static int scan_rec(p3m_ctx *c, int *d, int64_t n, int *tmp, int *total_out) {
  const int64_t nb = (n + SCAN_CH - 1) / SCAN_CH;
  if (nb == 1) {
    hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(SCAN_T), 0, c->stream, d, n, (const int *)nullptr, total_out);
    HIP_TRY(hipGetLastError());
    return P3M_OK;
  }
  hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(SCAN_T), 0, c->stream, (const int *)d, n, tmp);
  HIP_TRY(hipGetLastError());
  const int64_t nb_al = (nb + 3) & ~(int64_t)3;  // keep the next level 16-byte aligned
  P3M_TRY(scan_rec(c, tmp, nb, tmp + nb_al, nullptr));
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_T), 0, c->stream, d, n, (const int *)tmp, total_out);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// data must be 16-byte aligned and hold n+1 ints: data[n] receives the total.
static size_t scan_tmp_ints(int64_t n) {
  size_t need = 0;
  for (int64_t m = (n + SCAN_CH - 1) / SCAN_CH; m > 1; m = (m + SCAN_CH - 1) / SCAN_CH) need += (size_t)((m + 3) & ~(int64_t)3);
  return need + 8;
}
// sized once for the longest scan a context runs (the record flags): a hipFree / hipMalloc pair in the middle of a step
// is a device-wide synchronisation (28 ms on the default workload, every time the record count set a new maximum)
int scan_reserve(p3m_ctx *c, int64_t n_max) {
  const size_t nst = (size_t)((n_max + SCAN_CH - 1) / SCAN_CH) + 2;   // look-back words (k_scan_lookback) and the ticket counter, zero once
  if (nst > c->scan_state_n) {
    if (c->scan_state) (void)hipFree(c->scan_state);
    HIP_TRY(hipMalloc(&c->scan_state, (nst + 2) * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->scan_state, 0, (nst + 2) * sizeof(unsigned long long)));
    c->scan_state_n = nst; c->scan_epoch = 0;
  }
  const size_t need = scan_tmp_ints(n_max);
  if (need > c->scan_tmp_n) {
    if (c->scan_tmp) (void)hipFree(c->scan_tmp);
    HIP_TRY(hipMalloc(&c->scan_tmp, need * sizeof(int)));
    c->scan_tmp_n = need;
  }
  return P3M_OK;
}
int exclusive_scan_i32(p3m_ctx *c, int *data, int64_t n) {
  const int64_t nb = (n + SCAN_CH - 1) / SCAN_CH;
  if (nb > 1 && (size_t)nb + 2 <= c->scan_state_n && c->scan_epoch < 0x3ffffff0u) {   // one launch (k_scan_lookback); the ticket lives behind the words
    c->scan_epoch++;
    hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)nb), dim3(SCAN_T), 0, c->stream, data, n, c->scan_state, reinterpret_cast<int *>(c->scan_state + c->scan_state_n), c->scan_epoch, data + n);
    HIP_TRY(hipGetLastError());
    return P3M_OK;
  }
  const size_t need = scan_tmp_ints(n);
  if (need > c->scan_tmp_n) {
    if (c->scan_tmp) (void)hipFree(c->scan_tmp);
    HIP_TRY(hipMalloc(&c->scan_tmp, need * sizeof(int)));
    c->scan_tmp_n = need;
  }
  return scan_rec(c, data, n, c->scan_tmp, data + n);
}

// ------------------------------------------------------------------ zero_add / zero_flush (p3m_internal.h)
__global__ __launch_bounds__(256) void k_zero_many(ZeroList z) {
  const int b = blockIdx.y;
  unsigned *p = reinterpret_cast<unsigned *>(z.p[b]);
  const unsigned long long nw = z.n[b] >> 2;   // 4-byte words
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4 *q = reinterpret_cast<uint4 *>(p);
    const unsigned long long n4 = nw >> 2;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < (nw & 3)) p[4 * n4 + threadIdx.x] = 0u;
  } else
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nw; i += (unsigned long long)gridDim.x * 256) p[i] = 0u;
}
int zero_flush(p3m_ctx *c) {
  if (c->zl.cnt == 0) return P3M_OK;
  unsigned long long big = 0;
  for (int i = 0; i < c->zl.cnt; i++) big = std::max(big, c->zl.n[i]);
  const unsigned gx = (unsigned)std::min<unsigned long long>(128, (big / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(k_zero_many, dim3(gx, c->zl.cnt), dim3(256), 0, c->stream, c->zl);
  c->zl.cnt = 0;
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
int zero_add(p3m_ctx *c, void *p, size_t bytes) {
  if (bytes == 0) return P3M_OK;
  if (c->zl.cnt == P3M_ZERO_MAX) P3M_TRY(zero_flush(c));
  c->zl.p[c->zl.cnt] = p; c->zl.n[c->zl.cnt] = (bytes + 3) & ~(size_t)3; c->zl.cnt++;
  return P3M_OK;
}
