// pp.hip -- short-range particle-particle forces on the cell-sorted records.
//   pp_intra    : -DPPINT, particle_mesh_threaded.f90:274-285 (bucketing) + :324-361 (pairs)
//   pp_extended : -DPP_EXT, particle_mesh_threaded.f90:378-624
// The sorted order (records of one fine cell contiguous, cells of one x-row contiguous) replaces
// llf / hoc_fine / ll_fine: a cell's partners are index ranges.  FP32-ALU bound (about 20 flop per
// pair incl. the reciprocal square root); reported as pairs/s, not against the HBM roofline.
#include "p3m_internal.h"
#include "fft_core.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <type_traits>
#include <vector>

struct PPGeo { int T, nb, pt, E, Nn, ms, ppr; float rsoft, pp_bias, ncut; };

__device__ __forceinline__ float3 pair_force(const float4 &a, const float4 &b, float mass_p, float rsoft, float pp_bias) {
  // :336-344 : sep = x1-x2 ; rmag ; if (rmag>rsoft) force_pp = mass_p*(sep/(rmag*pp_bias)**3)
  const float sx = a.x - b.x, sy = a.y - b.y, sz = a.z - b.z;
  const float rmag = sqrtf(sx * sx + sy * sy + sz * sz);
  if (!(rmag > rsoft)) return make_float3(0.f, 0.f, 0.f);
  const float rb = rmag * pp_bias, rb3 = rb * rb * rb;
  return make_float3(mass_p * (sx / rb3), mass_p * (sy / rb3), mass_p * (sz / rb3));
}

// the maximum / minimum of an int over the wavefront's active lanes, in every lane: by DPP (p3m_internal.h) and one v_readlane of lane 63,
// which has to be active -- as __shfl_xor butterflies (ds_bpermute: an LDS round trip per step) the six-deep chains were 192 LDS operations
// of the heavy pass's front end.  A lane without a source keeps the neutral element.
template <bool MAX> __device__ __forceinline__ int wave_red_i(int v) {
  constexpr int ID = MAX ? (int)0x80000000 : 0x7fffffff;
  auto step = [&](auto CTRL, auto RM) {
    const int t = __builtin_amdgcn_update_dpp(ID, v, decltype(CTRL)::value, decltype(RM)::value, 0xf, false);
    v = MAX ? max(v, t) : min(v, t);
  };
  step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{}); step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}); step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) { return wave_red_i<true>(v); }
__device__ __forceinline__ int wave_min_i(int v) { return wave_red_i<false>(v); }
// ------------------------------------------------------------------ intra-cell PP
// One thread per physical record.  The reference buckets the chain of hoc coarse cell
// floor(x/mesh_scale) by sub-cell mod(i1-1,mesh_scale), i1 = floor(x + offset_tile) + 1
// (:248-249,:276-278), and sums all pairs of one bucket.  For every coarse cell in which no
// coordinate rounds across a cell face under the tile offset (cflag == 0, the overwhelmingly
// common case) a bucket IS a sorted fine cell: partners are the index range of the own cell.
// Flagged coarse cells take the slow path: scan the whole coarse cell (ms*ms x-rows) and compare
// the reference's bucket of every candidate.
__device__ __forceinline__ void ref_bucket(const float4 &p, const PPGeo &G, int cc[3], int sub[3]) {
  const int nct = G.pt / G.ms; const float xs[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int d = 0; d < 3; d++) {
    cc[d] = (int)floorf(xs[d] / (float)G.ms);                     // hoc coarse cell, 0-based (link_list.f90:19-21)
    const int t = cc[d] / nct;
    const float xl = xs[d] + ((float)G.nb - (float)(t * G.pt));   // :248
    sub[d] = ((int)floorf(xl)) % G.ms;                            // :277, (i1-1) mod mesh_scale
  }
}

// Dense cells: when a wavefront's records sit in cells of more than PP_INTRA_DENSE records, the wavefront stages the union of
// its lanes' cell ranges (contiguous in the sorted order) in LDS, PP_INTRA_CH partners at a time by coalesced loads, and every
// lane walks the records of ITS cell there (lanes of one cell read one address: a broadcast) -- instead of every lane streaming
// its whole cell from global memory.  The wavefront runs as many trips as its fullest cell has records; round 5's form (one
// v_readlane broadcast per partner of the UNION, each lane keeping those of its own cell) ran as many as the union has, three to
// five times more for blobs that spread over several cells.  Same partner order per lane (ascending sorted index) as the per-lane loop.
#ifndef PP_INTRA_DENSE
#define PP_INTRA_DENSE 6     // (3, 6: 0.323 ms per clustered 560 tile; 12: 0.333; 24: 0.375)
#endif
#define PP_INTRA_CH 256
__global__ __launch_bounds__(256) void k_pp_intra(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs,
                                                  const unsigned char *__restrict__ cflag, int n, PPGeo G, float mass_p, float a_mid, float dt,
                                                  float *__restrict__ fmax_out, float r2_soft, const unsigned char *__restrict__ done) {
  // done (or null): a byte per sorted record, set by k_pp_light for the records whose bucket pairs it has summed on its way (pp_extended:
  // the fused form); a wavefront of such records -- the rule at the background's density -- leaves before it reads anything else
  __shared__ float4 ldsp[4][PP_INTRA_CH];   // the dense path's partners, a chunk per wavefront
  const int s = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool skip = done != nullptr && s < n && done[s] != 0;
  if (done != nullptr && __all(skip || s >= n)) return;
  float mag = 0.f;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  bool phys = false, slow = false, closed = true;
  int cc[3] = {0, 0, 0}, sub[3] = {0, 0, 0}, q0 = 0, q1 = 0, flag = 0;
  // The records of a fine cell are neighbours in the sorted order: a record's cell mates are found by comparing cell indices along
  // the wavefront (plus the record before and the one after it) instead of two look-ups in cell_end per record, which at the mean
  // density cost four times the bytes of the records themselves.  Only a run that crosses the wavefront's ends reads cell_end
  // The kernel is a chain of round trips (at the reference's density it moves a third of the bytes its time would pay for): the requests
  // that do not depend on each other go out together -- the compiler, left alone, waits for each load inside the conditional block that
  // issues it (the empty asm statements name the registers that have to be there, i.e. where the wait may stand)
  int cell = -1, ncell = -1;                          // (E^3 < 2^31: pp_intra checks)
  {
    const int t = lane == 0 ? s - 1 : s + 1;           // the neighbours beyond the wavefront's ends
    const bool nb = (lane == 0 || lane == 63) && t >= 0 && t < n;
    // both through a buffer descriptor of the wavefront's window [s0 - 1, s0 + 65) of the records: a lane with nothing to read passes an
    // offset outside it and gets zeros -- no branch around the load, whose merge of the loaded registers is where the compiler waits
    const int s0 = __builtin_amdgcn_readfirstlane(s - lane), wb = max(s0 - 1, 0), we = min(n, s0 + 65);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(spos + wb), 0, (we - wb) * 16, 0x00020000);
    const float4 o = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, nb ? (t - wb) * 16 : 0x40000000, 0, 0));
    p = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (s - wb) * 16, 0, 0));   // (s >= n: outside, zeros)
    if (nb) ncell = (((int)floorf(o.z) + G.nb) * G.E + ((int)floorf(o.y) + G.nb)) * G.E + ((int)floorf(o.x) + G.nb);
  }
  if (s < n) {
    const float fNn = (float)G.Nn;
    phys = !skip && p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn;
    cell = (((int)floorf(p.z) + G.nb) * G.E + ((int)floorf(p.y) + G.nb)) * G.E + ((int)floorf(p.x) + G.nb);
    if (phys) {
      // the hoc coarse cell alone decides the path (ref_bucket's three float and six integer divisions were a third of this
      // kernel's instructions; x / mesh_scale is x * (1 / mesh_scale) bit for bit when mesh_scale is a power of two)
      const float fms = (float)G.ms, ims = 1.0f / fms;
      const bool pow2 = (G.ms & (G.ms - 1)) == 0;
      cc[0] = (int)floorf(pow2 ? p.x * ims : p.x / fms); cc[1] = (int)floorf(pow2 ? p.y * ims : p.y / fms); cc[2] = (int)floorf(pow2 ? p.z * ims : p.z / fms);
      const int Ec = G.E / G.ms, cb = G.nb / G.ms;
      // requested here, looked at after the bucket pairs of the sorted cell are summed (a flagged coarse cell -- rare -- discards them):
      // waiting for the byte before the partner loop was a fifth of this kernel's time at the reference's density
      flag = cflag[((cc[2] + cb) * Ec + (cc[1] + cb)) * Ec + (cc[0] + cb)];
    }
  }
  {
    const int before = __builtin_amdgcn_update_dpp(-2, cell, 0x138, 0xf, 0xf, false);   // wave_shr:1 (lane 0 keeps -2: it heads a run anyway)
    const unsigned long long heads = __ballot(lane == 0 || cell != before);            // bit l: a run of equal cells starts at lane l
    const int same = cell == ncell ? 1 : 0;
    const bool open_l = __builtin_amdgcn_readlane(same, 0) != 0, open_r = __builtin_amdgcn_readlane(same, 63) != 0;
    const int start = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
    const unsigned long long after = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
    const int end = after ? __ffsll((long long)after) - 1 : 64;
    if (phys) {
      if ((start == 0 && open_l) || (end == 64 && open_r)) { q0 = cs[cell]; q1 = cs[cell + 1]; closed = false; }
      else { const int s0 = s - lane; q0 = s0 + start; q1 = s0 + end; }
    }
  }
  float ax = 0.f, ay = 0.f, az = 0.f;
  const bool fast = phys;
  // the velocity of a record with a cell mate (6 % at the reference's density) is requested here, with the flag and the cell ranges,
  // not after the sums: one round trip less in the chain.  (vv is only read where vhave is set)
  float4 vv; const bool vhave = phys && (!closed || q1 - q0 >= 2);
  if (vhave) vv = vel[__float_as_int(p.w)];
  float4 *const L = ldsp[threadIdx.x >> 6];
  L[lane] = p;   // the wavefront's own records: the partners of every run that does not cross its ends
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // (LDS operations of a wavefront complete in order)
  const int maxc = (int)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_max_nonneg_to_last((float)(fast ? q1 - q0 : 0))), 63));   // (a count: exact as a float)
  if (maxc > PP_INTRA_DENSE) {
    asm volatile("" : "+v"(flag));   // a long walk: the records of flagged coarse cells stay out of it
    const bool fastd = fast && flag == 0;
    const int Q0 = wave_min_i(fastd ? q0 : 0x7fffffff), Q1 = wave_max_i(fastd ? q1 : 0);
    const float ibias = 1.0f / G.pp_bias;
    for (int base = Q0; base < Q1; base += PP_INTRA_CH) {
      const int m = min(PP_INTRA_CH, Q1 - base);
      for (int i = lane; i < m; i += 64) L[i] = spos[base + i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // (LDS operations of a wavefront complete in order)
      const int lo = fastd ? max(q0, base) : 0, hi = fastd ? min(q1, base + m) : 0;
      // four partners per trip: their reads and reciprocal square roots are independent (a blob's wavefront is alone on its SIMD and
      // paid the LDS round trip and the dependent chain of every partner in full: 235 clocks per partner), the sums stay in order.
      // A rejected partner adds (x * 0): the sums it leaves are the same up to the sign of a zero
      int q = lo;
      for (; q + 4 <= hi; q += 4) {
        float4 o[4]; float k[4];
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = L[q - base + u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          o[u].x = p.x - o[u].x; o[u].y = p.y - o[u].y; o[u].z = p.z - o[u].z;      // :336
          const float r2 = o[u].x * o[u].x + o[u].y * o[u].y + o[u].z * o[u].z;
          const float ib = __builtin_amdgcn_rsqf(r2) * ibias;
          k[u] = (q + u != s && r2 >= r2_soft) ? ib * ib * ib : 0.f;                // :340 rmag > rsoft, decided exactly on r^2
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { ax -= mass_p * (o[u].x * k[u]); ay -= mass_p * (o[u].y * k[u]); az -= mass_p * (o[u].z * k[u]); }   // :344-347
      }
      for (; q < hi; q++) {
        const float4 o = L[q - base];
        const float sx = p.x - o.x, sy = p.y - o.y, sz = p.z - o.z;
        const float r2 = sx * sx + sy * sy + sz * sz;
        if (q != s && r2 >= r2_soft) {
          const float ib = __builtin_amdgcn_rsqf(r2) * ibias, irb3 = ib * ib * ib;
          ax -= mass_p * (sx * irb3); ay -= mass_p * (sy * irb3); az -= mass_p * (sz * irb3);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // the chunk is read before the next one overwrites it
    }
  } else if (fast) {
    // (the reciprocal square root of the dense path above: the wavefront runs this loop as often as its fullest cell has records, and the
    // square root and three divisions of pair_force were three quarters of the kernel's instructions)
    // A run inside the wavefront -- all but the two at its ends -- reads its partners from the records staged above: the loop was a chain
    // of global round trips, as many as the wavefront's fullest cell has records (0.083 of the kernel's 0.22 ms per 560 tile)
    const float ibias = 1.0f / G.pp_bias;
    const int s0 = s - lane;
    auto pair = [&](int q, const float4 &o) {
      const float sx = p.x - o.x, sy = p.y - o.y, sz = p.z - o.z;                 // :336
      const float r2 = sx * sx + sy * sy + sz * sz;
      if (q != s && r2 >= r2_soft) {                                              // :340 rmag > rsoft, decided exactly on r^2
        const float ib = __builtin_amdgcn_rsqf(r2) * ibias, irb3 = ib * ib * ib;
        ax -= mass_p * (sx * irb3); ay -= mass_p * (sy * irb3); az -= mass_p * (sz * irb3);   // :344-347
      }
    };
    // (two loops: one loop over a pointer selected per lane reads both memories through flat loads)
    if (closed) for (int q = q0; q < q1; q++) pair(q, L[q - s0]);
    else for (int q = q0; q < q1; q++) pair(q, spos[q]);
  }
  // Records of flagged coarse cells (some record's reference bucket differs from its sorted cell): partners are the records
  // of the whole coarse cell with the same reference bucket.  Wavefront-cooperative: one flagged coarse cell at a time, its
  // ms*ms x-rows 64 candidates at a time, each candidate's bucket computed once by the lane that loaded it and broadcast
  // with its position (a thousand-particle cell made every lane stream and re-bucket the whole coarse cell on its own).
  asm volatile("" : "+v"(flag));   // (the look at the flag stands here, not where the byte was requested)
  slow = phys && flag != 0;
  if (slow) { ax = 0.f; ay = 0.f; az = 0.f; ref_bucket(p, G, cc, sub); }   // the sub-cell is only compared on the slow path
  {
    const int nct = G.pt / G.ms;
    const int mykey = (sub[2] * G.ms + sub[1]) * G.ms + sub[0];
    const float ibias = 1.0f / G.pp_bias;
    bool todo = phys && slow;
    unsigned long long pending;
    while ((pending = __ballot(todo)) != 0ull) {
      const int lead = __ffsll((long long)pending) - 1;
      const int c0 = __builtin_amdgcn_readlane(cc[0], lead), c1 = __builtin_amdgcn_readlane(cc[1], lead), c2 = __builtin_amdgcn_readlane(cc[2], lead);
      const bool mine = todo && cc[0] == c0 && cc[1] == c1 && cc[2] == c2;
      const int x0 = c0 * G.ms + G.nb;
      for (int dz = 0; dz < G.ms; dz++)
        for (int dy = 0; dy < G.ms; dy++) {
          const int64_t rb = ((int64_t)(c2 * G.ms + G.nb + dz) * G.E + (c1 * G.ms + G.nb + dy)) * G.E;
          const int r0 = cs[rb + x0], r1 = cs[rb + x0 + G.ms];
          for (int base = r0; base < r1; base += 64) {
            const int m = min(64, r1 - base);
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            int okey = -1;
            if (lane < m) {
              o = spos[base + lane];
              int oc[3], os[3];
              ref_bucket(o, G, oc, os);
              if (oc[0] == c0 && oc[1] == c1 && oc[2] == c2) okey = (os[2] * G.ms + os[1]) * G.ms + os[0];
            }
            for (int jj = 0; jj < m; jj++) {
              const int pk = __builtin_amdgcn_readlane(okey, jj);
              const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), jj));
              const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), jj));
              const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), jj));
              if (mine && pk == mykey && base + jj != s) {
                const float sx = p.x - px, sy = p.y - py, sz = p.z - pz;
                const float r2 = sx * sx + sy * sy + sz * sz;
                if (r2 >= r2_soft) {
                  const float ib = __builtin_amdgcn_rsqf(r2) * ibias, irb3 = ib * ib * ib;
                  ax -= mass_p * (sx * irb3); ay -= mass_p * (sy * irb3); az -= mass_p * (sz * irb3);
                }
              }
            }
          }
        }
      todo = todo && !mine;
    }
    (void)nct;
  }
  // A record alone in its bucket (15 of 16 at the reference's density) has three zero sums: its kick adds zero and its velocity -- a 16-byte
  // gather and a 16-byte scatter through the arrival index, twice the bytes of the record itself -- is left where it is
  if (phys && (ax != 0.f || ay != 0.f || az != 0.f)) {
    const int vi = __float_as_int(p.w); float4 v;                    // the velocity stays in arrival order (p3m_internal.h)
    if (vhave) v = vv; else v = vel[vi];
    v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;  // :349-350
    vel[vi] = v;
    mag = sqrtf(ax * ax + ay * ay + az * az);                       // :356
  }
  mag = wave_max_nonneg_to_last(mag);
  if ((threadIdx.x & 63) == 63 && mag > 0.f) p3m_atomic_max_nonneg(fmax_out + p3m_slot() * 16, mag);
}

static float first_r2_with_root_above(float t);
int pp_intra(p3m_ctx *c, float a_mid, float dt, float mass_p) {
  P3M_TRY(particles_full_cells(c));
  const Geometry &g = c->g;
  if ((int64_t)g.E * g.E * g.E >= (1ll << 31)) { p3m_set_error("pp_intra: the extended mesh does not fit a 32-bit cell index"); return P3M_EINVAL; }
  const unsigned char *done = c->pp_intra_fused ? c->pp_intra_done : nullptr;   // this step's pp_extended summed the bucket pairs of the records flagged there
  c->pp_intra_fused = false;
  if (c->np_all == 0) return P3M_OK;
  PPGeo G{g.T, g.nb, g.pt, g.E, g.Nn, g.ms, g.pp_range, c->p.rsoft, c->p.pp_bias, (float)g.ncut};
  hipLaunchKernelGGL(k_pp_intra, dim3(cdiv(c->np_all, 256)), dim3(256), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end,
                     (const unsigned char *)c->cflag, c->np_all, G, mass_p, a_mid, dt, c->d_red + 1 * P3M_RED_SPAN, first_r2_with_root_above(G.rsoft), done);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// ------------------------------------------------------------------ extended PP
// One 64-lane workgroup per (tile, z, y) row of the tile's region extended by pp_range cells
// (:397-402); one lane per record of the row.  Partner cells: Chebyshev distance 1..pp_range,
// clipped to the extended region exactly as the reference's half-shell sweep is (:496-523, incl. its
// omission of the pairs inside the top pp_range planes), so the
// per-tile maxval(|pp_ext_force_accum|) (:617) is reproduced including the partial sums of records
// in the rim.  Only records whose cell is in the physical tile are kicked (:576-590).
__global__ __launch_bounds__(64) void k_pp_ext(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, PPGeo G,
                                               float mass_p, float a_mid, float dt, float *__restrict__ tile_max) {
  const int e = G.pt + 2 * G.ppr;
  const int ry = blockIdx.x % e, rz = (blockIdx.x / e) % e, tile = blockIdx.x / (e * e);
  const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
  // extended region in extended-cell coordinates: [lo_d, lo_d + e)
  const int lox = tx * G.pt + G.nb - G.ppr, loy = ty * G.pt + G.nb - G.ppr, loz = tz * G.pt + G.nb - G.ppr;
  const int cy = loy + ry, cz = loz + rz;
  const int64_t rowb = ((int64_t)cz * G.E + cy) * G.E;
  const int p0 = cs[rowb + lox], p1 = cs[rowb + lox + e];
  float mymax = 0.f;
  const float tmax = G.ncut + sqrtf(3.0f);
  for (int s = p0 + threadIdx.x; s < p1; s += 64) {
    const float4 p = spos[s];
    const int cx = (int)floorf(p.x) + G.nb;                                // :412 (floor(xv)+1, global)
    float ax = 0.f, ay = 0.f, az = 0.f;
    int z0 = max(cz - G.ppr, loz), z1 = min(cz + G.ppr, loz + e - 1);
    // the reference's half-shell sweep starts only from the planes k <= pt + pp_range (:496 "we never loop towards
    // smaller z"): a pair whose two cells both lie in the top pp_range planes of the region is never formed
    if (cz - loz >= G.pt + G.ppr) z1 = min(z1, loz + G.pt + G.ppr - 1);
    const int y0 = max(cy - G.ppr, loy), y1 = min(cy + G.ppr, loy + e - 1);
    const int x0 = max(cx - G.ppr, lox), x1 = min(cx + G.ppr, lox + e - 1);
    for (int zz = z0; zz <= z1; zz++)
      for (int yy = y0; yy <= y1; yy++) {
        const int64_t rb = ((int64_t)zz * G.E + yy) * G.E;
        const bool own = (zz == cz && yy == cy);
        const int q0 = cs[rb + x0], q1 = cs[rb + x1 + 1];
        const int s0 = own ? cs[rb + cx] : 0, s1 = own ? cs[rb + cx + 1] : 0;  // own cell is excluded (:515-516)
        for (int q = q0; q < q1; q++) {
          if (own && q >= s0 && q < s1) { q = s1 - 1; continue; }
          const float4 o = spos[q];
          const float sx = p.x - o.x, sy = p.y - o.y, sz = p.z - o.z;            // :551
          const float rmag = sqrtf(sx * sx + sy * sy + sz * sz);
          if (rmag > G.rsoft) {                                                   // :558
            const float rb1 = rmag * G.pp_bias, rb3 = rb1 * rb1 * rb1;
            float fx = mass_p * (sx / rb3), fy = mass_p * (sy / rb3), fz = mass_p * (sz / rb3);
            if (!(rmag > tmax)) {                                                 // :559-564
              const float qq = rb1 / G.ncut;
              const float taper = 1.f - (7.0f / 4.0f) * (qq * qq * qq) + (3.0f / 4.0f) * (qq * qq * qq * qq * qq);
              fx *= taper; fy *= taper; fz *= taper;
            }
            ax -= fx; ay -= fy; az -= fz;                                         // :571
          }
        }
      }
    const bool phys = (cx >= lox + G.ppr && cx < lox + G.ppr + G.pt && ry >= G.ppr && ry < G.ppr + G.pt && rz >= G.ppr && rz < G.ppr + G.pt);
    if (phys) {                                                                   // :576-582
      const int vi = __float_as_int(spos[s].w); float4 v = vel[vi];   // the velocity stays in arrival order (p3m_internal.h)
      v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;
      vel[vi] = v;
    }
    mymax = fmaxf(mymax, sqrtf(ax * ax + ay * ay + az * az));                     // :617
  }
  for (int o = 32; o > 0; o >>= 1) mymax = fmaxf(mymax, __shfl_down(mymax, o, 64));
  if (threadIdx.x == 0 && mymax > 0.f) p3m_atomic_max_nonneg(tile_max + tile, mymax);
}


// task record: {patch, sub-task, gz << 20 | gy << 10 | xb, tz << 20 | ty << 10 | tx}: the patch's and the tile's coordinates are decoded here, once
// per task by one lane of a small kernel, instead of by four run-time divisions at the head of every task of the pair kernels
__global__ __launch_bounds__(256) void k_pp_fill2(const int *__restrict__ plan, int ngroups, int4 *__restrict__ task4, int cap, int npz, int npy, int npx, int T) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const int xb = g % npx, gy = (g / npx) % npy, gz = (g / (npx * npy)) % npz, tile = g / (npx * npy * npz);
  const int tz = tile / (T * T), ty = (tile / T) % T, tx = tile % T;
  const int k0 = plan[g], k1 = min(plan[g + 1], cap);
  for (int k = k0; k < k1; k++) task4[k] = make_int4(g, k - k0, (gz << 20) | (gy << 10) | xb, (tz << 20) | (ty << 10) | tx);
}
// Constants of one pair evaluation.  The force of a partner at separation s, r = |s| (:551-571):
//   -mass_p s / (r pp_bias)^3 * taper(q),  q = r pp_bias / ncut,  taper = 1 - 7/4 q^3 + 3/4 q^5 while r <= ncut + sqrt(3), else 1
// is formed as  -s * [K taper(q)] * (1/r)^3  with K = mass_p / pp_bias^3 folded into the taper's coefficients
struct PPForce { float c1, K, K34, K74, r2_soft, r2_taper, big_s, nrp_big, nbig_t, r2t_big; };
// The two cuts as arithmetic 0/1 factors (the swept evaluation, pp_sweep_row): with rp the float below r2_soft,
//   r2 >= r2_soft  <=>  r2 - rp > 0  <=>  clamp((r2 - rp) * 2^k) = 1   once 2^k * ulp(rp) >= 1   (else the difference is <= 0: factor 0)
// and r2 < r2_taper <=> clamp((r2_taper - r2) * 2^m) = 1 likewise.  Scaling by a power of two is exact and the fused multiply-add
// rounds once, so sign and zero of the differences are exact: the factors decide exactly as the comparisons do
static PPForce pp_force_constants(float mass_p, float pp_bias, float ncut, float r2_soft, float r2_taper) {
  const float ib = 1.0f / pp_bias, K = mass_p * (ib * ib * ib);
  const float rp = nextafterf(r2_soft, 0.0f);
  const float big_s = ldexpf(1.0f, std::min(120, 1 - ilogbf(r2_soft - rp)));
  const float rt_below = nextafterf(r2_taper, 0.0f);
  const float big_t = ldexpf(1.0f, std::min(100, 1 - ilogbf(r2_taper - rt_below)));
  return PPForce{pp_bias * (1.0f / ncut), K, 0.75f * K, -1.75f * K, r2_soft, r2_taper, big_s, -rp * big_s, -big_t, r2_taper * big_t};
}
// One partner of the extended sweep.  The hard cut (:558) is decided on r^2 computed in the reference's order (unfused); the force
// itself is formed from the hardware reciprocal square root with fused multiply-adds and the taper in Horner form, 25 instructions
// (the kick differs from the reference's association in the last bits: ~4e-7 relative per pair against the 1e-5 bar; k_pp_ext
// above keeps the reference's sqrt / division arithmetic, P3M_PP_EXT_REF=1), without branches: a lane outside the cut adds zero
__device__ __forceinline__ void pp_ext_eval(const float4 &p, float ox, float oy, float oz, const PPForce &F, float &ax, float &ay, float &az) {
  const float sx = p.x - ox, sy = p.y - oy, sz = p.z - oz;               // :551
  const float r2 = sx * sx + sy * sy + sz * sz;
  const float ir = __builtin_amdgcn_rsqf(r2), qq = (r2 * ir) * F.c1;
  const float q2 = qq * qq, q3 = q2 * qq;
  float tp = __builtin_fmaf(q3, __builtin_fmaf(q2, F.K34, F.K74), F.K);    // K (1 - 7/4 q^3 + 3/4 q^5) (:559-564)
  tp = r2 < F.r2_taper ? tp : F.K;
  float f = tp * ((ir * ir) * ir);
  f = r2 >= F.r2_soft ? f : 0.0f;                                        // :558, decided exactly on r^2 (first_r2_with_root_above)
  ax = __builtin_fmaf(-sx, f, ax); ay = __builtin_fmaf(-sy, f, ay); az = __builtin_fmaf(-sz, f, az);   // :571
}
// Two partners at once in the halves of packed registers (v_pk_add / v_pk_mul / v_pk_fma_f32: one issue slot for both).  A half
// that is not a partner (okA / okB false) adds zero -- by selection, not by a zero factor: a masked half may be the home record
// itself (r = 0).  Every half goes through the operations of pp_ext_eval; a home record's sum is formed as two partial sums,
// added at the end
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool TAPER_ALL>   // no pair within reach is beyond the taper's range: no switch (pp_extended)
__device__ __forceinline__ void pp_ext_eval2(const float4 &p, const float4 &A, const float4 &B, bool okA, bool okB, const PPForce &F,
                                             f32x2 &ax, f32x2 &ay, f32x2 &az) {
  const f32x2 ox = {A.x, B.x}, oy = {A.y, B.y}, oz = {A.z, B.z};
  const f32x2 sx = p.x - ox, sy = p.y - oy, sz = p.z - oz;               // :551
  const f32x2 r2 = sx * sx + sy * sy + sz * sz;
  const f32x2 ir = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
  const f32x2 qq = (r2 * ir) * F.c1;
  const f32x2 q2 = qq * qq, q3 = q2 * qq;
  const f32x2 k34 = {F.K34, F.K34}, k74 = {F.K74, F.K74}, kk = {F.K, F.K};
  f32x2 tp = __builtin_elementwise_fma(q3, __builtin_elementwise_fma(q2, k34, k74), kk);   // :559-564
  if (!TAPER_ALL) { tp.x = r2.x < F.r2_taper ? tp.x : F.K; tp.y = r2.y < F.r2_taper ? tp.y : F.K; }
  f32x2 f = tp * ((ir * ir) * ir);
  f.x = (okA && r2.x >= F.r2_soft) ? f.x : 0.0f; f.y = (okB && r2.y >= F.r2_soft) ? f.y : 0.0f;   // :558
  ax = __builtin_elementwise_fma(-sx, f, ax); ay = __builtin_elementwise_fma(-sy, f, ay); az = __builtin_elementwise_fma(-sz, f, az);   // :571
}


// The swept evaluation: every lane holds one home record (hx, hy, hz) and the wavefront walks the staged positions [ua, ub) of one
// partner row together, one LDS broadcast per partner.  A lane's window of the row and its own cell are intervals of positions:
// with d = v - (centre of the window) and k4 = 4 * (half width) + 1 the factor clamp(k4 - 4 |d|) is 1 inside and 0 outside (d and
// the half width are multiples of 1/2: exact), one fused multiply-add with the clamp modifier and one add per partner; the two
// cuts on r^2 are factors of the same kind (pp_force_constants).  All operands sit in vector registers; no compare, no select,
// no scalar-register operand (4.3 cycles of a SIMD's issue each against 2.6 for plain vector arithmetic: tools/valubench.hip):
// 26-27 vector instructions per partner.  OWN: some lane's own cell lies in this stretch (its partners are masked: :515-516)
struct PPSweepK { float c1, K, K34, K74, big_s, nrp_big, nbig_t, r2t_big, tiny; };
template <bool OWN, bool TAPER_ALL>
__device__ __forceinline__ void pp_sweep_row(const float4 *pr, int ua, int ub, float hx, float hy, float hz, float d, float k4, float dO, float k4o,
                                             const PPSweepK &S, float &ax, float &ay, float &az) {
  auto step = [&](const float4 &o) {
    float m;
    asm("v_fma_f32 %0, |%1|, -4.0, %2 clamp" : "=v"(m) : "v"(d), "v"(k4));          // 1 inside this lane's window of the row
    d += 1.0f;
    if (OWN) {
      float mo;
      asm("v_fma_f32 %0, |%1|, 4.0, %2 clamp" : "=v"(mo) : "v"(dO), "v"(k4o));      // 0 inside this lane's own cell (:515-516)
      dO += 1.0f;
      m *= mo;
    }
    const float sx = hx - o.x, sy = hy - o.y, sz = hz - o.z;               // :551
    const float r2 = sx * sx + sy * sy + sz * sz;                           // the reference's order, unfused: the cut is decided on it
    // r = 0 (the home record itself, or two records on one point): the reciprocal square root is taken of max(r^2, 1e-12) -- far
    // below r_soft^2 -- so that ir^3 stays near 1e18 and the cut factors below multiply a FINITE value by zero.  In BOTH templates: a
    // heavy rim record's own row can be swept for another lane of its group without that lane's window test ever holding the own
    // cell (the non-OWN template then met r = 0: inf * 0 = NaN in the partial sum, dropped silently from the tile maximum)
    float r2c;
    asm("v_max_f32 %0, %1, %2" : "=v"(r2c) : "v"(r2), "v"(S.tiny));
    const float ir = __builtin_amdgcn_rsqf(r2c), qq = (r2c * ir) * S.c1;
    const float q2 = qq * qq;
    float q3 = q2 * qq;
    if (!TAPER_ALL) {
      float mt;
      asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(mt) : "v"(r2), "v"(S.nbig_t), "v"(S.r2t_big));   // 1 while r2 < r2_taper (:559)
      q3 *= mt;
    }
    const float tp = __builtin_fmaf(q3, __builtin_fmaf(q2, S.K34, S.K74), S.K);   // K (1 - 7/4 q^3 + 3/4 q^5) (:559-564)
    float ms;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(ms) : "v"(r2), "v"(S.big_s), "v"(S.nrp_big));        // 1 when r2 >= r2_soft (:558)
    const float f = (tp * ((ir * ir) * ir)) * (ms * m);
    ax = __builtin_fmaf(-sx, f, ax); ay = __builtin_fmaf(-sy, f, ay); az = __builtin_fmaf(-sz, f, az);   // :571
  };
  int v = ua;
  for (; v + 1 < ub; v += 2) {                       // two partners per trip: both broadcasts in flight before the first is used
    const float4 o0 = pr[v], o1 = pr[v + 1];
    step(o0); step(o1);
  }
  if (v < ub) step(pr[v]);
}

// ------------------------------------------------------------------ extended PP: the partner region of a patch through LDS
// A task is up to PP3_NT home records of a 3-D PATCH (PP3_HZ planes x PP3_HY rows x xbw cells, 7/8 of PP3_NT home records at the
// mean density) worked by PP3_NT / 64 wavefronts, and everything its homes can reach -- the (HZ+2r) x (HY+2r) partner rows clipped
// to the patch's x range +-r -- is brought into LDS ONCE per task by row-wise coalesced loads: the cell offsets of every partner
// row (one load instruction per row, kept as 16-bit offsets from the row's first record) and the partner records themselves (a
// flat copy over the concatenated row segments).  (Gathering windows and partners straight from global memory, round 2's
// kernel, was bound by cache-line traffic: 50 cell offsets and ~15 partner records per home record, 44 cache lines per load
// instruction through an L2 that hit 44 % of the time.)  A home lane then finds its (2r+1)^2 row windows in the LDS offsets and
//   * lists the partners' LDS positions and sums over the list (the rule: ~15 partners), or
//   * walks its windows itself (few partners spread over more than 256 positions), or
//   * is HEAVY (more than PP3_LCAP partners: a blob's members and neighbours) and left to the second launch, which sweeps the
//     staged region wavefront-cooperatively (pp_sweep_row).
// Regions with more records than the staging area holds are worked off in batches of PP3_PCAP records of the concatenated row
// segments; a row segment of more than 65 534 records sends the task down a plain per-lane path over global memory.  The partner
// order per home record (rows in z, y order, ascending sorted index) is that of k_pp_ext for listed and walked records; swept
// records add four partial sums (the wavefronts' shares of that order).
#define PP3_HZ 8        // patch: 8 planes x 8 rows (4 x 16 measured 4.5 % slower: 160 partner rows per task against 144)
#define PP3_HY 8
#define PP3_NT 256
// A patch of more than PP3_DENSE home records (a blob) is cut into tasks of PP3_NTD home records instead of PP3_NT: its tasks are
// thousands of times longer than those of the background, and the kernel ends when the last of them does -- 2700 tasks of 256
// records on 1280 resident workgroups left the chip 22 % idle (48 blobs of 13 000).  The wavefronts without home records share
// the sweep of the others' (see "wpg" in k_pp_ext3)
#define PP3_DENSE 1024
#define PP3_NTD 128
__host__ __device__ __forceinline__ int pp3_task_homes(int count) { return count > PP3_DENSE ? PP3_NTD : PP3_NT; }
#ifndef PP3_PCAP
#define PP3_PCAP 704
#endif
#define PP3_LCAP 32      // list entries per lane
#define PP3_CROWDED (PP3_PCAP - 24)   // staged records of a region from which the window lists are built with care (the background's: 648 +- 25)
#ifndef PP3_WPE
#define PP3_WPE 5        // wavefronts per SIMD the register allocation aims at (96 VGPRs): five workgroups of 32 KB LDS per CU at the reference density
#endif
#define PP3_LSTR 36      // bytes per lane of the list (entries + room for the stores past the capacity; 9 words: conflict-free across lanes)
#define PP3_NSEG 8      // task counters (each hands out a contiguous eighth of the tasks)
// one wavefront per patch, one lane per home row (PP3_HZ * PP3_HY = 64): a thread per patch walked its 64 rows alone, 130 us per tile
__global__ __launch_bounds__(256) void k_pp_plan3(const int *__restrict__ cs, PPGeo G, int npy, int npx, int xbw, int ngroups, int *__restrict__ plan) {
  static_assert(PP3_HZ * PP3_HY == 64, "one lane per home row");
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
  if (g >= ngroups) return;
  const int e = G.pt + 2 * G.ppr, npz = (e + PP3_HZ - 1) / PP3_HZ;
  const int xb = g % npx, gy = (g / npx) % npy, gz = (g / (npx * npy)) % npz, tile = g / (npx * npy * npz);
  const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
  const int lox = tx * G.pt + G.nb - G.ppr, loy = ty * G.pt + G.nb - G.ppr, loz = tz * G.pt + G.nb - G.ppr;
  const int hx0 = lox + xb * xbw, hx1 = min(hx0 + xbw, lox + e);
  int count = 0;
  const int rz = gz * PP3_HZ + j / PP3_HY, ry = gy * PP3_HY + j % PP3_HY;
  if (rz < e && ry < e) { const int64_t rb = ((int64_t)(loz + rz) * G.E + (loy + ry)) * G.E; count = cs[rb + hx1] - cs[rb + hx0]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o, 64);
  if (j == 0) { const int nt = pp3_task_homes(count); plan[g] = (count + nt - 1) / nt; }
}
// Two launches of one kernel.  PASS 0 (the light pass) works every task of the plan: its lanes list or walk their partners; the
// records that are HEAVY (more than PP3_LCAP partners: members and neighbours of a blob) are left out and the task is entered, with
// a bit per heavy lane, in the heavy-task list.  PASS 1 (the heavy pass) works that list: it stages the same region again and
// SWEEPS it for the heavy records.  Two launches because the two halves want different registers (the sweep's live state spilled
// the list path's: 2.19 -> 2.52 ms per tile at uniform density with both in one kernel) and because a launch of heavy tasks only
// balances itself (the heavy tasks of one kernel ended long after its light ones: a quarter of the chip's time idle)
#define PP3_HREC 16      // ints per heavy-task record: patch, sub-task, the heavy lanes' box of partner rows and cells (3 words), 8 words of lane bits
template <int PPR, bool TAPER_ALL, int PASS>   // PPR > 0: pp_range known at compile time (the reference's default 2); 0: any.  TAPER_ALL: no pair within reach is beyond the taper's range
__global__ __launch_bounds__(PP3_NT) __attribute__((amdgpu_waves_per_eu(PASS == 0 ? PP3_WPE : PP3_WPE - 1, PASS == 0 ? PP3_WPE : PP3_WPE - 1))) void k_pp_ext3(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, PPGeo G, PPForce F,
                                                    float a_mid, float dt, float *__restrict__ tile_max,
                                                    const int4 *__restrict__ task4, const int *__restrict__ ntask_ptr, int npy, int npx, int xbw, int ntask_cap, int *__restrict__ counter,
                                                    int Wp, int NRmax, int fat_limit, int *__restrict__ htask, int *__restrict__ hcount) {
  // fat_limit: 65534 (see "fat" below); task4: the task records (k_pp_fill2; pass 0 works the *ntask_ptr first of them: the whole plan, or the list of
  // tasks the lean light pass k_pp_light left to this kernel); Wp: entries per row of the offset table (xbw + 2r + 1 rounded
  // up to even); NRmax: partner rows; htask / hcount: the heavy-task list (PP3_HREC ints per task, written by pass 0, read by pass 1) and its length
  extern __shared__ int sm[];
  constexpr int NH = PP3_HZ * PP3_HY, NW = PP3_NT / 64;
  // LDS: prec | offs | list | rowg | cum | rstart | roff | misc.  The unrolled window walk below reads offs and cum at rows up to
  // 2 NRY + 2 outside the region for lanes whose window is clipped away (their counts are forced to zero): with this order such
  // reads land in prec / list resp. rowg / rstart (5.4 KB and 168 B at most), inside the allocation
  float4 *prec = reinterpret_cast<float4 *>(sm);                              // [PP3_PCAP]  staged partner records (16-byte aligned: first)
  unsigned short *offs = reinterpret_cast<unsigned short *>(prec + PP3_PCAP); // [NRmax][Wp]  records of partner row r before cell X0 + i
  // partner lists: one byte per entry, the partner's position in the staged batch minus the position of the lane's first window
  // (a lane's windows span ~200 positions at the reference's density with 8 x 8-row patches; a lane whose span exceeds 255 is
  // walked, not listed): 9 KB per workgroup instead of 16 KB of 16-bit entries, which is what lets a fifth workgroup stay on a CU
  unsigned char *list = reinterpret_cast<unsigned char *>(offs + (size_t)NRmax * Wp);   // [NW][64][PP3_LSTR]
  int *rowg = reinterpret_cast<int *>(list + NW * 64 * PP3_LSTR);             // [NRmax]      sorted index of the row segment's first record
  int *cum = rowg + NRmax;                          // [NRmax + 1]  records of the rows before r in the concatenated partner sequence
  int *rstart = cum + NRmax + 1, *roff = rstart + NH;   // home rows: first record, exclusive prefix of the home counts ([NH + 1])
  int *misc = roff + NH + 1;                        // [48]: [0] task state, [1] fat flag, [4] the task has heavy lanes, [5..7] the drawn task, [8..15] heavy lanes (bits) per wavefront, [16..19] light lanes per wavefront, [20..22] pass 1: the task's box, [24..47] the wavefronts' boxes
  // pass 1 has no lists: the area holds the task's heavy records, their home rows and the wavefronts' partial sums
  float4 *hrec = reinterpret_cast<float4 *>(list);                            // [PP3_NT]
  float *part = reinterpret_cast<float *>(hrec + PP3_NT);                     // [NW][64][3]
  unsigned char *hj = reinterpret_cast<unsigned char *>(part + NW * 64 * 3);  // [PP3_NT]
  static_assert(PP3_NT * 16 + (PP3_NT / 64) * 64 * 12 + PP3_NT <= (PP3_NT / 64) * 64 * PP3_LSTR, "the heavy pass's tables fit the list area");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned char *mylist = list + (wv * 64 + lane) * PP3_LSTR;                 // this lane's entries
  const int ppr = G.ppr, e = G.pt + 2 * ppr, E = G.E, npz = (e + PP3_HZ - 1) / PP3_HZ;
  const int ntask = PASS == 0 ? min(*ntask_ptr, ntask_cap) : min(*hcount, ntask_cap);
  if ((int)blockIdx.x >= ntask) return;            // no task (no blob anywhere, nothing left by the lean pass: the rule at the background's density), or fewer tasks than
                                                   // workgroups: the launch costs its dispatch only (a workgroup without a task probed eight dry counters between barriers)
  const int per = (ntask + PP3_NSEG - 1) / PP3_NSEG;
  int seg = blockIdx.x % PP3_NSEG;
  // Thread 0 runs a two-stage pipeline of task draws, one stage per task worked: the counter atomic of the task after the next
  // (stage A) and the {group, sub-task} look-up of the next one (stage B) are in flight while the workgroup works; each result is
  // first touched a whole task after its request, so the workgroup never waits for a draw (drawn in one go -- counter, task ->
  // group, group -> first task -- wavefront 0 stood still for three round trips at the head of every task and the others
  // waited for it at the first barrier).  A segment that has run dry costs one idle trip through the loop.
  int a_tf = 0, a_seg = 0, tried = 0;
  bool a_live = false, b_live = false;
  int2 b_val = make_int2(0, 0);
  int b_t = 0, b_box0 = 0, b_box1 = 0, b_box2 = 0;
  auto advance = [&]() {   // thread 0 only
    b_live = false;
    if (a_live) {
      const int sbeg = min(a_seg * per, ntask), send = min(sbeg + per, ntask), t = sbeg + a_tf;
      if (t < send) {
        if (PASS == 0) { const int4 t4 = task4[t]; b_val = make_int2(t4.x, t4.y); }
        else { const int4 r4 = *reinterpret_cast<const int4 *>(htask + (size_t)PP3_HREC * t); b_val = make_int2(r4.x, r4.y); b_box0 = r4.z; b_box1 = r4.w; b_box2 = htask[(size_t)PP3_HREC * t + 4]; }
        b_t = t; b_live = true; tried = 0;
      }
      else { tried++; seg = (seg + 1) % PP3_NSEG; }
    }
    a_live = tried < PP3_NSEG;
    if (a_live) { a_seg = seg; a_tf = atomicAdd(counter + 32 * seg, 1); }
  };
  if (tid == 0) { a_live = true; a_seg = seg; a_tf = atomicAdd(counter + 32 * seg, 1); advance(); }
  if (tid < 2 * NW) misc[8 + tid] = 0;
  if (tid == 0) misc[4] = 0;
  int prev_g = 0, prev_sub = 0, cur_g = 0, cur_sub = 0;   // thread 0: the task just worked, the task published
  for (;;) {
    __syncthreads();                                  // the previous task's readers of the LDS tables (and of misc) are done
    if (tid == 0) { misc[0] = b_live ? 1 : (a_live ? 0 : -1); misc[1] = 0; misc[5] = b_t; misc[6] = b_val.x; misc[7] = b_val.y; cur_g = b_val.x; cur_sub = b_val.y; if (PASS == 1) { misc[20] = b_box0; misc[21] = b_box1; misc[22] = b_box2; } advance(); }
    __syncthreads();
    const int state = misc[0];
    if (PASS == 0 && tid == 0) {
      // (the flag is read in the shadow of the state's read; the waves write it, the bits and the boxes again only behind the two
      // barriers of the table set-up below, so the record is made here, off the path the other wavefronts wait on)
      if (misc[4]) {
        // the task just worked has heavy lanes: a bit per lane (left in misc[8..15] by the wavefronts' first lanes) and the box of
        // partner rows and cells those lanes reach (misc[24 + 6 w ..]) into the heavy-task list: here, behind a barrier the loop has
        // anyway, instead of one more barrier per task
        const int slot = atomicAdd(hcount, 1);
        int bz0 = 0x7fff, bz1 = 0, by0 = 0x7fff, by1 = 0, bx0 = 0x7fff, bx1 = 0;
#pragma unroll
        for (int w = 0; w < NW; w++)
          if (misc[8 + 2 * w] | misc[9 + 2 * w]) {
            const int *bw = misc + 24 + 6 * w;
            bz0 = min(bz0, bw[0]); bz1 = max(bz1, bw[1]); by0 = min(by0, bw[2]); by1 = max(by1, bw[3]); bx0 = min(bx0, bw[4]); bx1 = max(bx1, bw[5]);
          }
        if (slot < ntask_cap) {
          int *rec = htask + (size_t)PP3_HREC * slot;
          rec[0] = prev_g; rec[1] = prev_sub; rec[2] = bz0 | (bz1 << 16); rec[3] = by0 | (by1 << 16); rec[4] = bx0 | (bx1 << 16);
#pragma unroll
          for (int k = 0; k < 2 * NW; k++) rec[5 + k] = misc[8 + k];
        }
#pragma unroll
        for (int k = 0; k < 2 * NW; k++) misc[8 + k] = 0;
        misc[4] = 0;
      }
      prev_g = cur_g; prev_sub = cur_sub;
    }
    if (state < 0) break;                             // every segment has run dry
    if (state == 0) continue;                         // a dry segment: the next draw is on its way
    const int g = misc[6], sub = misc[7] & 0xffffff, half = misc[7] >> 24;   // half (pass 1 only): a task of k_pp_light's second launch -- the lower (1) / upper (2) half of the patch along x
    const int xb = g % npx, gy = (g / npx) % npy, gz = (g / (npx * npy)) % npz, tile = g / (npx * npy * npz);
    const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
    const int lox = tx * G.pt + G.nb - ppr, loy = ty * G.pt + G.nb - ppr, loz = tz * G.pt + G.nb - ppr;
    const int px0 = lox + xb * xbw, px1 = min(px0 + xbw, lox + e), wl = (xbw + 1) >> 1;
    const int hx0 = half == 2 ? min(px0 + wl, px1) : px0, hx1 = half == 1 ? min(px0 + wl, px1) : px1;   // home cells [hx0, hx1)
    const int hz0 = loz + gz * PP3_HZ, hy0 = loy + gy * PP3_HY;
    // the partner region: rows [Z0, Z1] x [Y0, Y1], cells [X0, X1]
    // (pass 1: only what the task's heavy lanes reach -- the box pass 0 recorded: a blob's neighbourhood is a third of the patch's)
    const int Z0 = PASS == 0 ? max(hz0 - ppr, loz) : (misc[20] & 0xffff), Z1 = PASS == 0 ? min(hz0 + PP3_HZ - 1 + ppr, loz + e - 1) : (misc[20] >> 16);
    const int Y0 = PASS == 0 ? max(hy0 - ppr, loy) : (misc[21] & 0xffff), Y1 = PASS == 0 ? min(hy0 + PP3_HY - 1 + ppr, loy + e - 1) : (misc[21] >> 16);
    const int X0 = PASS == 0 ? max(hx0 - ppr, lox) : (misc[22] & 0xffff), X1 = PASS == 0 ? min(hx1 - 1 + ppr, lox + e - 1) : (misc[22] >> 16);
    const int NRY = Y1 - Y0 + 1, NR = (Z1 - Z0 + 1) * NRY, W = X1 - X0 + 2;
    if (PASS == 1 && tid < 2 * NW) misc[8 + tid] = htask[(size_t)PP3_HREC * misc[5] + 5 + tid];   // the task's heavy lanes (read behind the next barrier)
    // home rows of the patch (wavefront 0), cell offsets of every partner row (one coalesced load per row, all wavefronts)
    int hcnt = 0, hst = 0;
    if (tid < NH) {
      const int rz = gz * PP3_HZ + tid / PP3_HY, ry = gy * PP3_HY + tid % PP3_HY;
      if (rz < e && ry < e) { const int64_t rb = ((int64_t)(loz + rz) * E + (loy + ry)) * E; hst = cs[rb + hx0]; hcnt = cs[rb + hx1] - hst; }
    }
    constexpr int RCH = 40;                           // (HZ + 4) (HY + 4) / NW rows: one trip at the default reach
    {
      // this wavefront's rows wv, wv + NW, ...: the row index and its (z, y) are walked in scalar registers (as quotients of
      // a per-lane row number they cost 30 vector instructions per row, half of the kernel's instructions)
      const int wvu = __builtin_amdgcn_readfirstlane(wv);
      int yy = Y0 + wvu, zz = Z0;
      while (yy > Y1) { yy -= NRY; zz++; }
      for (int rc = wvu; rc < NR; rc += NW * RCH) {   // RCH rows at a time: their loads are in flight together
        int o[RCH];
#pragma unroll
        for (int u = 0; u < RCH; u++) {
          o[u] = 0;
          if (rc + u * NW < NR) {
            if (lane < W) o[u] = cs[((int64_t)zz * E + yy) * E + X0 + lane];
            yy += NW;
            while (yy > Y1) { yy -= NRY; zz++; }
          }
        }
#pragma unroll
        for (int u = 0; u < RCH; u++) {
          const int r = rc + u * NW;
          if (r >= NR) break;
          const int first = __builtin_amdgcn_readfirstlane(o[u]), d = o[u] - first;
          if (lane < W) offs[r * Wp + lane] = (unsigned short)min(d, 65535);
          if (lane == W - 1) { if (d > fat_limit) misc[1] = 1; }
          if (lane == 0) rowg[r] = first;
        }
      }
    }
    if (wv == 0) {
      const int hinc = wave_scan_incl_i(hcnt);
      const int total = __builtin_amdgcn_readlane(hinc, NH - 1);
      if (lane < NH) { rstart[lane] = hst; roff[lane] = hinc - hcnt; }
      if (lane == 0) roff[NH] = total;
    }
    __syncthreads();
    const int total = roff[NH];
    const bool fat = misc[1] != 0;                    // uniform
    // concatenated partner sequence: cum[r] = records of rows < r
    if (wv == 0) {
      int carry = 0;
      for (int r0 = 0; r0 < NR; r0 += 64) {
        const int r = r0 + lane;
        const int cnt = r < NR ? (int)offs[r * Wp + W - 1] : 0;
        const int inc = wave_scan_incl_i(cnt);
        if (r < NR) cum[r] = carry + inc - cnt;
        carry += __builtin_amdgcn_readlane(inc, 63);
      }
      if (lane == 0) cum[NR] = carry;
    }
    // this thread's home record
    const int nth = pp3_task_homes(total);           // home records per task of this patch (uniform)
    const int h = sub * nth + tid;
    const bool valid = tid < nth && h < total;
    int j = 0;
    { int lo = 0, hi = NH; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= h) lo = mid; else hi = mid; } j = lo; }
    const int cz = hz0 + (valid ? j / PP3_HY : 0), cy = hy0 + (valid ? j % PP3_HY : 0);
    __syncthreads();
    const int Ptot = cum[NR];                         // uniform
    // flat copy of the partner records [b0, b1) of the concatenated sequence: v -> (row, index) by bisection over cum; the loads of
    // a thread's records are in flight together
    auto stage = [&](int b0, int b1) {
      constexpr int NV = (PP3_PCAP + PP3_NT - 1) / PP3_NT;
      float4 q[NV];
#pragma unroll
      for (int u = 0; u < NV; u++) {
        const int v = b0 + tid + u * PP3_NT;
        q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < b1) {
          int lo = 0, hi = NR;
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cum[mid] <= v) lo = mid; else hi = mid; }
          q[u] = spos[rowg[lo] + (v - cum[lo])];
        }
      }
#pragma unroll
      for (int u = 0; u < NV; u++) { const int v = b0 + tid + u * PP3_NT; if (v < b1) prec[v - b0] = q[u]; }
    };
    // a record's windows: rows [z0, z1] x [y0, y1], cells [x0, x1] -- Chebyshev distance <= pp_range clipped to the tile's extended
    // region; the reference's half-shell sweep starts only from the planes k <= pt + pp_range (:496): see k_pp_ext
    auto reach = [&](int ccx, int ccy, int ccz, int &z0, int &z1, int &y0, int &y1, int &x0, int &x1) {
      z0 = max(ccz - ppr, loz); z1 = min(ccz + ppr, loz + e - 1);
      if (ccz - loz >= G.pt + ppr) z1 = min(z1, loz + G.pt + ppr - 1);
      y0 = max(ccy - ppr, loy); y1 = min(ccy + ppr, loy + e - 1);
      x0 = max(ccx - ppr, lox); x1 = min(ccx + ppr, lox + e - 1);
    };
    auto is_phys = [&](int ccx, int ccy, int ccz) {                                // :576-582: only records of the physical tile are kicked
      return ccx >= lox + ppr && ccx < lox + ppr + G.pt && ccy - loy >= ppr && ccy - loy < ppr + G.pt && ccz - loz >= ppr && ccz - loz < ppr + G.pt;
    };
    float mag = 0.f;
    if constexpr (PASS == 0) {
    // The whole region in one batch (the rule away from blobs): the home records are among the staged ones -- they are read from
    // LDS, not in a round trip of their own before the staging (a task is a chain of dependent round trips: cell offsets, records,
    // velocities; the kernel's time is set by how many of them a task waits for, not by its instructions)
    const bool single = PPR > 0 && !fat && Ptot <= PP3_PCAP;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (single) {
      stage(0, Ptot);
      __syncthreads();
      if (valid) { const int r = (cz - Z0) * NRY + (cy - Y0); p = prec[cum[r] + (int)offs[r * Wp + (hx0 - X0)] + (h - roff[j])]; }
    } else if (valid) p = spos[rstart[j] + (h - roff[j])];
    const int cx = valid ? (int)floorf(p.x) + G.nb : hx0;                       // :412
    int z0, z1, y0, y1, x0, x1;
    reach(cx, cy, cz, z0, z1, y0, y1, x0, x1);
    const bool phys = valid && is_phys(cx, cy, cz);
    const int vi = rec_index(p);                      // the velocity stays in arrival order (p3m_internal.h); fetched now, needed after the sums
    float4 vrec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (phys) vrec = vel[vi];
    float ax = 0.f, ay = 0.f, az = 0.f;
    bool heavy = false;                               // this lane's home record is left to the heavy pass
    if (fat) {
      // a row segment holds more records than a 16-bit offset counts: windows and partners straight from global memory
      if (valid)
        for (int zz = z0; zz <= z1; zz++)
          for (int yy = y0; yy <= y1; yy++) {
            const int64_t rb = ((int64_t)zz * E + yy) * E;
            const int a = cs[rb + x0], b = cs[rb + x1 + 1];
            int s0 = 0, s1 = 0;
            if (zz == cz && yy == cy) { s0 = cs[rb + cx]; s1 = cs[rb + cx + 1]; }    // own cell is excluded (:515-516)
            for (int q = a; q < b; q++) {
              if (q >= s0 && q < s1) { q = s1 - 1; continue; }
              const float4 o = spos[q];
              pp_ext_eval(p, o.x, o.y, o.z, F, ax, ay, az);
            }
          }
    } else {
      f32x2 ax2 = {0.f, 0.f}, ay2 = {0.f, 0.f}, az2 = {0.f, 0.f};
      // own cell, excluded (:515-516), as positions of the concatenated sequence
      int own0 = 0, own1 = 0;
      if (valid) {
        const int r = (cz - Z0) * NRY + (cy - Y0);
        own0 = cum[r] + offs[r * Wp + (cx - X0)]; own1 = cum[r] + offs[r * Wp + (cx - X0) + 1];
      }
      // ---- who lists, who walks, who is left to the heavy pass.  A lane lists its partners' positions (one byte each, relative to
      // the start of its first window) when they are at most PP3_LCAP and span at most 256 positions; with few partners over a
      // longer span (7 % of the lanes at the reference's density: the windows of five planes lie ~230 positions apart) it walks
      // its windows itself; the records of a blob and their neighbours (more than PP3_LCAP partners) are HEAVY
      int n = 0, base = 0;
      bool walker = false;
      if (single) {
        // the whole region is staged and the reach is known: the (2r+1)^2 windows unrolled, a window's
        // first three records stored unconditionally at the list's end and the end advanced by a compare (no branches: 25
        // windows cost ~700 instructions per wavefront where a predicated version took 2500); a window of more
        // than three records anywhere in the wavefront sends it through the tail loop
        bool ovf = false;
        // the lane's first window starts the lowest position it will list (the windows come in the order of the staged sequence)
        if (valid) { const int r = (z0 - Z0) * NRY + (y0 - Y0); base = cum[r] + (int)offs[r * Wp + (x0 - X0)]; }
        const int rh = valid ? (cz - Z0) * NRY + (cy - Y0) : 0;
        const unsigned short *wa = offs + rh * Wp + (x0 - X0), *wb = offs + rh * Wp + (x1 + 1 - X0);
        const int *wc = cum + rh;
        bool yok[2 * PPR + 1];
#pragma unroll
        for (int dy = -PPR; dy <= PPR; dy++) yok[dy + PPR] = valid && cy + dy >= y0 && cy + dy <= y1;
        // three consecutive slots take a, a+1, a+2 whatever the count: the next window overwrites what was not a partner.
        // crowded (uniform: the region holds more records than the background's, i.e. part of a blob): a lane that is past the
        // capacity stores nothing more, so a wavefront next to a blob runs the tail loop 35 times in all, not 35 times per window
        const bool crowded = Ptot > PP3_CROWDED;
        auto list_planes = [&](auto CR) {             // the whole loop once per case: the choice costs one branch, not one per window
          constexpr bool CROWDED = decltype(CR)::value;
          auto append = [&](int a, int cnt) {
            unsigned char *o = mylist + min(n, PP3_LCAP);   // a lane past the capacity is heavy, not listed
            const int e = a - base;
            o[0] = (unsigned char)e; o[1] = (unsigned char)(e + 1); o[2] = (unsigned char)(e + 2);
            ovf = ovf || (cnt > 0 && e + cnt > 256);
            if (!CROWDED) {
              if (__any(cnt > 3))
                for (int k = 3; k < min(cnt, PP3_LCAP + 3); k++) mylist[min(n + k, PP3_LCAP + 2)] = (unsigned char)(e + k);
            } else if (__any(cnt > 3 && n <= PP3_LCAP))
              for (int k = 3; k < cnt && n + k < PP3_LCAP + 3; k++) mylist[n + k] = (unsigned char)(e + k);
            n += cnt;
          };
#pragma unroll 1
          for (int dz = -PPR; dz <= PPR; dz++) {      // one plane of windows per trip: the tail loops are not replicated 25 times
            const bool zok = cz + dz >= z0 && cz + dz <= z1;
#pragma unroll
            for (int dy = -PPR; dy <= PPR; dy++) {
              const bool rv = zok && yok[dy + PPR];
              const int d = dz * NRY + dy;           // uniform
              const int cr = wc[d], a = cr + (int)wa[d * Wp], b = cr + (int)wb[d * Wp];
              if (dz == 0 && dy == 0) { append(a, rv ? own0 - a : 0); append(own1, rv ? b - own1 : 0); }   // the own cell splits the own row's window (:515-516)
              else append(a, rv ? b - a : 0);
            }
          }
        };
        if (crowded) list_planes(std::true_type{}); else list_planes(std::false_type{});
        heavy = valid && n > PP3_LCAP;
        walker = valid && !heavy && ovf;
      } else if (valid) {
        // several batches (or any reach): count first -- a lane is heavy or light for the whole task
        int first = 0, last = 0;
        for (int zz = z0; zz <= z1; zz++)
          for (int yy = y0; yy <= y1; yy++) {
            const int r = (zz - Z0) * NRY + (yy - Y0), cr = cum[r];
            const int va = cr + (int)offs[r * Wp + (x0 - X0)], vb = cr + (int)offs[r * Wp + (x1 + 1 - X0)];
            if (zz == z0 && yy == y0) first = va;
            last = vb;
            n += vb - va;
          }
        if (cz >= z0 && cz <= z1) n -= own1 - own0;    // the own cell, if the own row is among the windows at all: a record of the planes
                                                       // above pt + pp_range sweeps downwards only (reach), its own plane is not one of them
        heavy = n > PP3_LCAP;
        walker = !heavy && last - first > 256;
      }
      // ---- the task's heavy lanes, one bit each, for the heavy-task list (thread 0 enters them at the top of the loop)
      unsigned long long hb = __ballot(heavy);
      if (hb) {                                       // uniform per wavefront, and rare away from blobs
        // one or two heavy lanes with moderately many partners (the tail of the background's distribution; the fringe of a blob
        // in the next patch) walk: a task of the heavy pass for them would cost more than their walk
        if (__popcll(hb) <= 2 && wave_max_i(heavy ? n : 0) <= 2 * PP3_LCAP) { walker = walker || heavy; heavy = false; hb = 0ull; }
        else {
          const int bz0 = wave_min_i(heavy ? z0 : 0x7fff), bz1 = wave_max_i(heavy ? z1 : 0), by0 = wave_min_i(heavy ? y0 : 0x7fff), by1 = wave_max_i(heavy ? y1 : 0);
          const int bx0 = wave_min_i(heavy ? x0 : 0x7fff), bx1 = wave_max_i(heavy ? x1 : 0);
          if (lane == 0) {
            misc[8 + 2 * wv] = (int)(unsigned)hb; misc[9 + 2 * wv] = (int)(unsigned)(hb >> 32); misc[4] = 1;
            int *bw = misc + 24 + 6 * wv;
            bw[0] = bz0; bw[1] = bz1; bw[2] = by0; bw[3] = by1; bw[4] = bx0; bw[5] = bx1;
          }
        }
      }
      int NL = 0;
      if (!single) {                                  // several batches: staged (with barriers) only if somebody lists or walks
        const unsigned long long lb = __ballot(valid && !heavy);
        if (lane == 0) misc[16 + wv] = __popcll(lb);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; w++) NL += misc[16 + w];
      }
      // ---- the light records: sums over their lists
      auto eval_list = [&](int nl) {
        const int nmax = wave_max_i(nl);
        for (int k = 0; k < nmax; k += 4) {           // four partners in flight at a time, two per evaluation; no branches: a slot past the list's end reads record 0 and adds zero
          float4 o[4];
          const unsigned e4 = *reinterpret_cast<const unsigned *>(mylist + k);   // four entries
#pragma unroll
          for (int u = 0; u < 4; u++) { const int i = base + (int)((e4 >> (8 * u)) & 255u); o[u] = prec[k + u < nl ? i : 0]; }
          pp_ext_eval2<TAPER_ALL>(p, o[0], o[1], k < nl, k + 1 < nl, F, ax2, ay2, az2);
          pp_ext_eval2<TAPER_ALL>(p, o[2], o[3], k + 2 < nl, k + 3 < nl, F, ax2, ay2, az2);
        }
      };
      // a walker's windows inside the batch [b0, b1), partners from the staged batch, two at a time
      auto walk = [&](int b0, int b1) {
        for (int zz = z0; zz <= z1; zz++)
          for (int yy = y0; yy <= y1; yy++) {
            const int r = (zz - Z0) * NRY + (yy - Y0);
            const int va = max(cum[r] + (int)offs[r * Wp + (x0 - X0)], b0), vb = min(cum[r] + (int)offs[r * Wp + (x1 + 1 - X0)], b1);
            int v = va;
            while (v < vb) {
              if (v >= own0 && v < own1) { v = own1; continue; }            // own cell is excluded (:515-516)
              const int w = v + 1;
              const bool okB = w < vb && !(w >= own0 && w < own1);
              pp_ext_eval2<TAPER_ALL>(p, prec[v - b0], prec[(okB ? w : v) - b0], true, okB, F, ax2, ay2, az2);
              v += 2;
            }
          }
      };
      // one trip for a region staged in one batch (listed above); else batch by batch: stage, list the batch's share, sum
      if (single || NL > 0)
        for (int b0 = 0; b0 < Ptot; b0 += PP3_PCAP) {   // batches of the concatenated partner sequence
          const int b1 = min(b0 + PP3_PCAP, Ptot);
          int nl = (heavy || walker) ? 0 : n;
          if (!single) {
            __syncthreads();                            // the previous batch's readers are done (the first trip: nothing is staged yet)
            stage(b0, b1);
            __syncthreads();
            nl = 0;
            if (valid && !heavy && !walker) {
              // window by window: the positions of this batch, relative to the lane's first position in it
              bool have = false;
              for (int zz = z0; zz <= z1; zz++)
                for (int yy = y0; yy <= y1; yy++) {
                  const int r = (zz - Z0) * NRY + (yy - Y0), cr = cum[r] - b0;
                  int va = cr + (int)offs[r * Wp + (x0 - X0)], vb = cr + (int)offs[r * Wp + (x1 + 1 - X0)];   // positions in the batch
                  va = max(va, 0); vb = min(vb, b1 - b0);
                  const bool ownrow = (zz == cz && yy == cy);
                  // the own cell [o0, o1) splits the own row's window in two (:515-516)
                  const int o0 = ownrow ? min(max(own0 - b0, va), vb) : vb, o1 = ownrow ? min(max(own1 - b0, va), vb) : vb;
#pragma unroll
                  for (int half = 0; half < 2; half++) {
                    const int a = half == 0 ? va : o1, b = half == 0 ? o0 : vb;
                    if (half == 1 && !ownrow) break;
                    if (b > a && !have) { base = a; have = true; }
                    for (int v = a; v < b; v++) { if (nl < PP3_LCAP) mylist[nl] = (unsigned char)(v - base); nl++; }
                  }
                }
            }
          }
          eval_list(nl);
          if (walker) walk(b0, b1);
        }
      ax += ax2.x + ax2.y; ay += ay2.x + ay2.y; az += az2.x + az2.y;
    }
    if (valid && !heavy) {
      if (phys) {                                                                   // :576-582
        float4 v = vrec;
        v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;
        vel[vi] = v;
      }
      mag = sqrtf(ax * ax + ay * ay + az * az);                                     // :617
    }
    } else {
    // ================================================================ PASS 1: the heavy records of the task
    // The task's heavy lanes (bits set by pass 0), compacted in lane order (the order of the home records) into groups of 64: the
    // records and their home rows in LDS.  Every group is swept by all four wavefronts, each taking its share of the partners (of
    // every chunk of 64 union rows: shares of equal length in the row-major partner sequence); the partial sums are added in the
    // order of the shares afterwards.  A lane's partners come in the order of the list path (rows in z, y order, ascending
    // position) within a share.  The sweep itself: pp_sweep_row
    const unsigned long long hb = ((unsigned long long)(unsigned)misc[9 + 2 * wv] << 32) | (unsigned)misc[8 + 2 * wv];
    const bool heavy = valid && ((hb >> lane) & 1ull);
    int hoff = 0, Hn = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { const int c = __popc((unsigned)misc[8 + 2 * w]) + __popc((unsigned)misc[9 + 2 * w]); if (w < wv) hoff += c; Hn += c; }
    if (heavy) {
      const int k = hoff + __popcll(hb & ((1ull << lane) - 1ull));
      hrec[k] = spos[rstart[j] + (h - roff[j])];
      hj[k] = (unsigned char)j;
    }
    const int ngrp = (Hn + 63) >> 6;                  // uniform, <= NW
    PPSweepK SK{F.c1, F.K, F.K34, F.K74, F.big_s, F.nrp_big, F.nbig_t, F.r2t_big, 1.0e-12f};
    // the constants of the swept evaluation in VECTOR registers: an instruction with a scalar-register operand costs 4.3 cycles of
    // a SIMD's issue against 2.6 (tools/valubench.hip, profiles/r04_valubench.txt)
    asm volatile("" : "+v"(SK.c1), "+v"(SK.K), "+v"(SK.K34), "+v"(SK.K74), "+v"(SK.big_s), "+v"(SK.nrp_big), "+v"(SK.nbig_t), "+v"(SK.r2t_big), "+v"(SK.tiny));
    float acc[NW][3];
#pragma unroll
    for (int gi = 0; gi < NW; gi++) { acc[gi][0] = 0.f; acc[gi][1] = 0.f; acc[gi][2] = 0.f; }
    const int nbat = (Ptot + PP3_PCAP - 1) / PP3_PCAP;
    for (int ib = 0; ib < nbat; ib++) {
      const int b0 = ib * PP3_PCAP, b1 = min(b0 + PP3_PCAP, Ptot);
      __syncthreads();                                // hrec / hj are written (first trip); the previous batch's readers are done
      stage(b0, b1);
      __syncthreads();
      const float4 *pr = prec - b0;
#pragma unroll
      for (int gi = 0; gi < NW; gi++) {
        if (gi >= ngrp) break;
        // the group's records, one per lane, their windows and the union of them
        const int k = gi * 64 + lane;
        const bool hvalid = k < Hn;
        float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
        int hcx = hx0, hcy = hy0, hcz = hz0, hz0_ = 0, hz1_ = -1, hy0_ = 0, hy1_ = -1, hx0_ = 0, hx1_ = 0, hown0 = 0, hown1 = 0;
        if (hvalid) {
          hp = hrec[k];
          const int hjj = hj[k];
          hcz = hz0 + hjj / PP3_HY; hcy = hy0 + hjj % PP3_HY; hcx = (int)floorf(hp.x) + G.nb;                      // :412
          reach(hcx, hcy, hcz, hz0_, hz1_, hy0_, hy1_, hx0_, hx1_);
          const int r = (hcz - Z0) * NRY + (hcy - Y0);
          hown0 = cum[r] + offs[r * Wp + (hcx - X0)]; hown1 = cum[r] + offs[r * Wp + (hcx - X0) + 1];
        }
        const float hx = hp.x, hy = hp.y, hz = hp.z;
        const int uz0 = wave_min_i(hvalid ? hz0_ : 0x7fffffff), uz1 = wave_max_i(hvalid ? hz1_ : -0x7fffffff);
        const int uy0 = wave_min_i(hvalid ? hy0_ : 0x7fffffff), uy1 = wave_max_i(hvalid ? hy1_ : -0x7fffffff);
        const int ux0 = wave_min_i(hvalid ? hx0_ : 0x7fffffff), ux1 = wave_max_i(hvalid ? hx1_ : -0x7fffffff);
        float sx_ = acc[gi][0], sy_ = acc[gi][1], sz_ = acc[gi][2];
        // The union's rows, 64 at a time with one row per lane: the stretch [ua, ub) of the row inside the batch, cut down to this
        // wavefront's share.  Rows with an empty stretch are never visited; for the others the lanes' own window ends are fetched
        // one row ahead of the sweep (a blob of 205 spreads over ~40 rows of ~6 partners: fetched row by row in the loop, the
        // three dependent LDS round trips per row cost more than the sweep itself)
        const int nuy = uy1 - uy0 + 1, nur = (uz1 - uz0 + 1) * nuy;
        const int inv = (65536 + nuy - 1) / nuy;                                            // i / nuy = (i * inv) >> 16 for i < 4096
        for (int rc = 0; rc < nur; rc += 64) {
          const int i = rc + lane;
          int ua_l = 0, ub_l = 0, crow_l = 0, zy_l = 0;
          if (i < nur) {
            const int zq = (i * inv) >> 16, zz = uz0 + zq, yy = uy0 + (i - zq * nuy);
            const int r = (zz - Z0) * NRY + (yy - Y0);
            crow_l = cum[r];
            ua_l = max(crow_l + (int)offs[r * Wp + (ux0 - X0)], b0); ub_l = min(crow_l + (int)offs[r * Wp + (ux1 + 1 - X0)], b1);
            zy_l = (zz << 16) | yy;
          }
          {
            // this wavefront's share of the chunk's partners: positions [lo, hi) of the concatenation of the rows' stretches
            const int cnt = max(ub_l - ua_l, 0);
            const int inc = wave_scan_incl_i(cnt);
            const int tot = __builtin_amdgcn_readlane(inc, 63), per_w = (tot + NW - 1) / NW, lo = wv * per_w, hi = min(lo + per_w, tot);
            const int before = inc - cnt;                                                  // partners of the rows before this one
            ua_l += max(lo - before, 0); ub_l -= max(inc - hi, 0);
          }
          unsigned long long todo = __ballot(ua_l < ub_l);
          int n_zy = 0, n_crow = 0, n_ua = 0, n_ub = 0, n_araw = 0, n_braw = 0;
          bool n_rowok = false;
          auto prefetch = [&](int ln) {            // ln: uniform
            n_zy = __builtin_amdgcn_readlane(zy_l, ln); n_crow = __builtin_amdgcn_readlane(crow_l, ln);
            n_ua = __builtin_amdgcn_readlane(ua_l, ln); n_ub = __builtin_amdgcn_readlane(ub_l, ln);
            const int zz = n_zy >> 16, yy = n_zy & 0xffff, r = (zz - Z0) * NRY + (yy - Y0);
            n_rowok = hvalid && zz >= hz0_ && zz <= hz1_ && yy >= hy0_ && yy <= hy1_;
            n_araw = 0; n_braw = 0;
            if (n_rowok) { n_araw = (int)offs[r * Wp + (hx0_ - X0)]; n_braw = (int)offs[r * Wp + (hx1_ + 1 - X0)]; }
          };
          if (todo) prefetch(__builtin_ctzll(todo));
          while (todo) {
            const int zy = n_zy, crow = n_crow, ua = n_ua, ub = n_ub, araw = n_araw, braw = n_braw;
            const bool rowok = n_rowok;
            todo &= todo - 1ull;
            if (todo) prefetch(__builtin_ctzll(todo));
            // this lane's window of the row [a, a + len) and, on its own row, its own cell [own0, own0 + olen) (:515-516), as
            // the centre and half width of an interval test on the position
            const int a = rowok ? crow + araw : ua, len = braw - araw;
            const bool ownrow = rowok && (zy >> 16) == hcz && (zy & 0xffff) == hcy;
            const float lenf = (float)len, olenf = ownrow ? (float)(hown1 - hown0) : 0.0f;
            const float d0 = (float)(ua - a) - 0.5f * (lenf - 1.0f), k4 = 2.0f * lenf - 1.0f;
#ifdef PP_SWEEP_STATS   // diagnostic build: swept partners (x 64 lane slots), rows visited, lanes with a home record, into hcount[8..]
            if (lane == 0) { atomicAdd(reinterpret_cast<unsigned long long *>(hcount + 8), (unsigned long long)(ub - ua)); atomicAdd(reinterpret_cast<unsigned long long *>(hcount + 10), 1ull); }
            { int lo_ = rowok && len > 0 ? max(a, ua) : 0x7fffffff, hi_ = rowok && len > 0 ? min(a + len, ub) : -0x7fffffff;   // the stretch some lane wants
              lo_ = wave_min_i(lo_); hi_ = wave_max_i(hi_);
              if (lane == 0 && hi_ > lo_) atomicAdd(reinterpret_cast<unsigned long long *>(hcount + 14), (unsigned long long)(hi_ - lo_)); }
            { const int acc_ = rowok ? max(min(a + len, ub) - max(a, ua), 0) : 0; unsigned long long t_ = (unsigned long long)acc_;
              for (int o_ = 32; o_ > 0; o_ >>= 1) t_ += __shfl_xor(t_, o_, 64);
              if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(hcount + 12), t_); }
#endif
            if (__any(ownrow)) {
              const float dO = (float)(ua - hown0) - 0.5f * (olenf - 1.0f), k4o = 1.0f - 2.0f * olenf;
              pp_sweep_row<true, TAPER_ALL>(pr, ua, ub, hx, hy, hz, d0, k4, dO, k4o, SK, sx_, sy_, sz_);
            } else pp_sweep_row<false, TAPER_ALL>(pr, ua, ub, hx, hy, hz, d0, k4, 0.f, 0.f, SK, sx_, sy_, sz_);
          }
        }
        acc[gi][0] = sx_; acc[gi][1] = sy_; acc[gi][2] = sz_;
      }
    }
    // the wavefronts' shares of a group's sums, added in the order of the shares by wavefront (group mod NW), which kicks
#pragma unroll
    for (int gi = 0; gi < NW; gi++) {
      if (gi >= ngrp) break;
      __syncthreads();                                // (the previous group's partial sums are read)
      float *ps = part + (wv * 64 + lane) * 3;
      ps[0] = acc[gi][0]; ps[1] = acc[gi][1]; ps[2] = acc[gi][2];
      __syncthreads();
      const int k = gi * 64 + lane;
      if (wv == gi && k < Hn) {
        float sx_ = 0.f, sy_ = 0.f, sz_ = 0.f;
#pragma unroll
        for (int q = 0; q < NW; q++) { const float *pq = part + (q * 64 + lane) * 3; sx_ += pq[0]; sy_ += pq[1]; sz_ += pq[2]; }
        const float4 hp = hrec[k];
        const int hjj = hj[k];
        if (is_phys((int)floorf(hp.x) + G.nb, hy0 + hjj % PP3_HY, hz0 + hjj / PP3_HY)) {      // :576-582
          const int hvi = rec_index(hp);
          float4 v = vel[hvi];
          v.x = v.x + sx_ * a_mid * P3M_G_F * dt; v.y = v.y + sy_ * a_mid * P3M_G_F * dt; v.z = v.z + sz_ * a_mid * P3M_G_F * dt;
          vel[hvi] = v;
        }
        mag = sqrtf(sx_ * sx_ + sy_ * sy_ + sz_ * sz_);                             // :617
      }
    }
    }
    mag = wave_max_nonneg_to_last(mag);
    if (lane == 63 && mag > 0.f) p3m_atomic_max_nonneg(tile_max + tile, mag);
  }
}

// ------------------------------------------------------------------ extended PP: the LEAN light pass (round 6)
// k_pp_ext3's pass 0 spent 2.9 vector wavefront-instructions per pair evaluation at the reference's density (profiles/r05_pp_sq_counters.txt):
// a quarter of them evaluations, the rest staging (bisections, per-row index arithmetic), 28 instructions for each of a record's 25 windows,
// and a window walk by every wavefront for the few lanes whose one-byte list entries overflowed.  k_pp_light is that pass for the common
// task -- pp_range 2, the whole partner region staged in one batch, no row segment of more than 255 records -- with every table laid out
// for compile-time offsets.  A crowded patch whose two halves along x fit is worked as two half patches by a second launch (LISTED); what is
// left then is entered in a list that k_pp_plan_slow + k_pp_ext3<.., 0> work afterwards (nothing at the background's density).
//   * no plan: a task is a patch, drawn from eight counters; its and its tile's coordinates are decoded by thread 0 a task ahead
//     (multiplications by reciprocals); a patch of more than 256 home records runs a second sub-task inside;
//   * geometry: the (8 + 4) x (8 + 4) partner rows of a patch are ALWAYS laid out whole, x boundaries hx0 - 2 ... hx1 + 2; rows outside the
//     tile's extended region are empty rows and boundaries outside it are clamped to it, so no window is ever clipped per lane: a window is
//     5 cells of one row = two bytes of the row's table five apart, at a compile-time offset from the lane's first window;
//   * offset table T: ONE BYTE per boundary (records of the row before it), 36 bytes per row; the row's first staged position times 16 in
//     cum16.  A wavefront loads three planes of 12 rows (36 buffer loads in flight: scalar row offset + lane offset), stores each row
//     minus its first value, and takes the rows' counts back from the table for one DPP scan: three vector instructions per row, no bisection;
//   * staging: a wavefront copies its own 36 rows, (row, slot < 8) items flattened over its lanes (rows of more than 8 records: a second trip);
//   * lists: 16-bit entries = the partner's byte address in the staged records.  A window is appended by storing its first entry -- and
//     five more only in the lanes whose window holds more than one record -- and one add (the next window overwrites what was not a
//     partner); a window of more than six records (1e-6 at the background's density) sends the lanes that use their lists through a rolled
//     general loop.  16-bit stores only: a wider store at this alignment is legal and stalls the LDS.  No entry
//     can overflow, so no lane walks its windows at the background's density;
//   * evaluation: four entries per trip, two per packed evaluation; the list's tail is padded with the home record itself (r = 0: below
//     the soft cut, adds zero), so a trip needs one compare for its four partners.
// Records with more than PPL_LCAP partners are heavy and left to k_pp_ext3<.., 1> exactly as before (same task records, same lane bits).
#define PPL_NRY 12            // partner rows per plane (PP3_HY + 4) and planes (PP3_HZ + 4)
#define PPL_NR 144
#define PPL_TS 36             // bytes per row of T: xbw + 5 boundaries
#define PPL_XBW_MAX (PPL_TS - 5)
#define PPL_PCAP 704
#define PPL_LCAP 32
#define PPL_LENT 40           // entries per lane (LCAP + the spill of a window's six stores): 80 bytes, so that the four entries of an evaluation trip are one
                              // ALIGNED 8-byte read (76 bytes: an odd number of words between the lanes' lists, but every other lane's read misaligned)
typedef __attribute__((address_space(3))) volatile unsigned short lds_vu16;   // an LDS pointer by type: a volatile generic pointer compiles to flat stores
struct PPLShared {
  float4 prec[PPL_PCAP + 6];                      // staged partner records (+ what the spill entries of a window's 12-byte store can address)
  unsigned char T[PPL_NR * PPL_TS + 16];          // T[r][i]: records of row r before boundary i
  unsigned short cum16[PPL_NR + 8];               // 16 * (staged records before row r)
  unsigned short lcum[PPL_NR + 8];                // records of the rows of the same wavefront before row r
  int rowg[PPL_NR + 8];                           // sorted index of the row segment's first record
  unsigned short lists[PP3_NT * PPL_LENT];
  alignas(16) int wtot[4];                        // staged records per wavefront
  int hsum[8];                                    // a crowded patch: the wavefronts' shares of its two halves' staged records
  int roff[PP3_HZ * PP3_HY + 1];                  // exclusive prefix of the home rows' counts
  int misc[48];                                   // [0] task state, [1] slow flag, [4] the task has heavy lanes, [5..8] the task record, [8+8..] see below
};
// (the arguments only thread 0, one lane per task or a rare branch reads -- PPLRare -- were tried behind a pointer into device memory, to free the
// thirty scalar registers they hold: fewer spills, and slower by a fifth -- every scalar load is waited for with the LDS counter at zero)
struct PPLRare {
  int *counter; int *htask, *hcount; int *slow; int *slowcount;   // task counters, heavy-task list (k_pp_ext3<.., 1>), patches left to the general pass
  float *tile_max, *fmax_intra;
  int ntask_cap, fat_limit;
  int npx, npxy, per_tile; unsigned m_npx, m_npxy, m_T, m_T2;      // patches per row of patches, per plane, per tile; reciprocals (mulhi) of those and of T, T^2
  // a crowded patch (more staged records than the LDS holds) whose two HALVES along x both fit is entered here twice, patch * 2 + half, and
  // worked by a second launch of the same kernel over this list (list != null: the tasks are its entries, half patches); only what is
  // crowded even then, or holds a row segment of more than 255 records, is left to the general pass
  int *retry, *retrycount; const int *list, *listcount;
};
struct PPLArgs {
  const float4 *spos; float4 *vel; const int *cs; PPLRare rare;
  // -DPPINT fused (or intra_done = null): the pairs inside a home record's own cell -- the reference's buckets wherever the coarse cell is not
  // flagged (k_pp_intra) -- are summed from the staged records too (:324-361); the records done are flagged for k_pp_intra, which works the rest
  unsigned char *intra_done; const unsigned char *cflag;
  int T, nb, pt, E, ms, xbw, ngroups;
  float c1, K, K34, K74, r2_soft, r2_taper, a_mid, dt;            // pp_force_constants
};
__device__ __forceinline__ void pp_ext_eval_l(const float4 &p, float ox, float oy, float oz, const PPLArgs &F, float &ax, float &ay, float &az) {   // pp_ext_eval on the light pass's constants
  const float sx = p.x - ox, sy = p.y - oy, sz = p.z - oz;               // :551
  const float r2 = sx * sx + sy * sy + sz * sz;
  const float ir = __builtin_amdgcn_rsqf(r2), qq = (r2 * ir) * F.c1;
  const float q2 = qq * qq, q3 = q2 * qq;
  float tp = __builtin_fmaf(q3, __builtin_fmaf(q2, F.K34, F.K74), F.K);    // :559-564
  tp = r2 < F.r2_taper ? tp : F.K;
  float f = tp * ((ir * ir) * ir);
  f = r2 >= F.r2_soft ? f : 0.0f;                                        // :558
  ax = __builtin_fmaf(-sx, f, ax); ay = __builtin_fmaf(-sy, f, ay); az = __builtin_fmaf(-sz, f, az);   // :571
}
template <bool TAPER_ALL>
__device__ __forceinline__ void pp_ext_eval2s(const float4 &p, const float4 &A, const float4 &B, bool ok, const PPLArgs &F, f32x2 &ax, f32x2 &ay, f32x2 &az) {
  const f32x2 ox = {A.x, B.x}, oy = {A.y, B.y}, oz = {A.z, B.z};
  const f32x2 sx = p.x - ox, sy = p.y - oy, sz = p.z - oz;               // :551
  const f32x2 r2 = sx * sx + sy * sy + sz * sz;
  const f32x2 ir = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
  const f32x2 qq = (r2 * ir) * F.c1;
  const f32x2 q2 = qq * qq, q3 = q2 * qq;
  const f32x2 k34 = {F.K34, F.K34}, k74 = {F.K74, F.K74}, kk = {F.K, F.K};
  f32x2 tp = __builtin_elementwise_fma(q3, __builtin_elementwise_fma(q2, k34, k74), kk);   // :559-564
  if (!TAPER_ALL) { tp.x = r2.x < F.r2_taper ? tp.x : F.K; tp.y = r2.y < F.r2_taper ? tp.y : F.K; }
  f32x2 f = tp * ((ir * ir) * ir);
  f.x = (ok && r2.x >= F.r2_soft) ? f.x : 0.0f; f.y = (ok && r2.y >= F.r2_soft) ? f.y : 0.0f;   // :558 (by selection: r = 0 is met here)
  ax = __builtin_elementwise_fma(-sx, f, ax); ay = __builtin_elementwise_fma(-sy, f, ay); az = __builtin_elementwise_fma(-sz, f, az);   // :571
}
template <bool TAPER_ALL, bool FUSE, bool LISTED>   // FUSE: -DPPINT's bucket pairs on the way (PPLArgs::intra_done); LISTED: the second launch (PPLRare::list)
__global__ __launch_bounds__(PP3_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pp_light(const PPLArgs A) {
  __shared__ PPLShared S;
  static_assert(PP3_NT == 256 && PP3_HZ == 8 && PP3_HY == 8, "three planes of twelve rows per wavefront");
  constexpr int NH = PP3_HZ * PP3_HY, NW = PP3_NT / 64, ppr = 2, RW = PPL_NR / NW;   // RW = 36 rows per wavefront = 3 planes
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const PPLArgs &G = A, &F = A;                      // (geometry and force constants by their old names)
  const PPLRare *const R = &A.rare;
  const int e = G.pt + 2 * ppr, E = G.E;
  // a task is a PATCH (no plan: k_pp_plan3's pass over the cell table, the scan and k_pp_fill2 cost 0.13 ms per tile; a patch of more than
  // PP3_NT home records -- two per cent of them at the background's density -- is left to the general pass like a crowded one)
  constexpr bool listed = LISTED;                     // the second launch: half patches from the first one's retry list
  const int ntask = listed ? min(*R->listcount, 2 * A.ngroups) : A.ngroups;
  if ((int)blockIdx.x >= ntask) return;
  const int per = (ntask + PP3_NSEG - 1) / PP3_NSEG;
  int seg = blockIdx.x % PP3_NSEG;
  // the patch's and the tile's coordinates, decoded by thread 0 a task ahead (multiplications by reciprocals; one real division)
  auto decode = [&](int t) {
    const int gh = listed ? R->list[t] : 2 * t, g = gh >> 1, half = listed ? (gh & 1) + 1 : 0;   // half: 0 the whole patch, 1 / 2 its lower / upper half along x
    const int tile = g / R->per_tile; int rem = g - tile * R->per_tile;
    const int gz = R->npxy == 1 ? rem : (int)__umulhi((unsigned)rem, R->m_npxy); rem -= gz * R->npxy;      // (the reciprocal of 1 does not fit 32 bits)
    const int gy = R->npx == 1 ? rem : (int)__umulhi((unsigned)rem, R->m_npx), xb = rem - gy * R->npx;
    const int tz = (int)__umulhi((unsigned)tile, R->m_T2), t2 = tile - tz * A.T * A.T, ty = (int)__umulhi((unsigned)t2, R->m_T), tx = t2 - ty * A.T;
    return make_int4(g, half << 24, (gz << 20) | (gy << 10) | xb, (tz << 20) | (ty << 10) | tx);
  };
  // thread 0: the two-stage pipeline of task draws (see k_pp_ext3)
  int a_tf = 0, a_seg = 0, tried = 0;
  bool a_live = false, b_live = false;
  int4 b_val = make_int4(0, 0, 0, 0), cur = make_int4(0, 0, 0, 0), prev = make_int4(0, 0, 0, 0);
  auto advance = [&]() {
    b_live = false;
    if (a_live) {
      const int sbeg = min(a_seg * per, ntask), send = min(sbeg + per, ntask), t = sbeg + a_tf;
      if (t < send) { b_val = decode(t); b_live = true; tried = 0; }
      else { tried++; seg = (seg + 1) % PP3_NSEG; }
    }
    a_live = tried < PP3_NSEG;
    if (a_live) { a_seg = seg; a_tf = atomicAdd(R->counter + 32 * seg, 1); }
  };
  if (tid == 0) { a_live = true; a_seg = seg; a_tf = atomicAdd(R->counter + 32 * seg, 1); advance(); }
  // Every list entry that is ever READ must address finite numbers: an entry past a lane's list end (stale, or never written) is
  // evaluated with a zero factor, and 0 * (p - NaN) is NaN.  The records area and the lists start as zeros; from then on an entry is
  // zero or a staged address, and a staged slot holds zeros or some task's records
  for (int i = tid; i < (int)(sizeof(S.prec) / 16); i += PP3_NT) S.prec[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < (int)(sizeof(S.lists) / 4); i += PP3_NT) reinterpret_cast<unsigned *>(S.lists)[i] = 0u;
  if (tid < 2 * NW) S.misc[16 + tid] = 0;            // [16..23] heavy lanes (bits) per wavefront, [24..47] the wavefronts' boxes
  if (tid == 0) S.misc[4] = 0;
  const unsigned mybase = (unsigned)(tid * PPL_LENT * 2), mycap = mybase + 2 * PPL_LCAP;   // byte offsets into S.lists
  // cell offsets through a buffer descriptor: a load is then a scalar row offset + a 32-bit lane offset (pp_extended takes this pass
  // only where the cell table and the records are shorter than 4 GB)
  const __amdgpu_buffer_rsrc_t cs_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(A.cs), 0, (int)0xffffffffu, 0x00020000);
#ifndef PPL_STAGE_GLOBAL
  const __amdgpu_buffer_rsrc_t sp_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(A.spos), 0, (int)0xffffffffu, 0x00020000);
#endif
  char *const lbytes = reinterpret_cast<char *>(S.lists);
  const char *const pbytes = reinterpret_cast<const char *>(S.prec);
  // thread 0, behind a barrier that follows the wavefronts' writes: the task (patch g, sub-task sub) just worked has heavy lanes -- its record
  // for k_pp_ext3<.., 1> (same layout: a bit per lane, the box of partner rows and cells those lanes reach)
  auto flush_heavy = [&](int g, int sub) {
    if (!S.misc[4]) return;
    const int slot = atomicAdd(R->hcount, 1);
    int bz0 = 0x7fff, bz1 = 0, by0 = 0x7fff, by1 = 0, bx0 = 0x7fff, bx1 = 0;
#pragma unroll
    for (int w = 0; w < NW; w++)
      if (S.misc[16 + 2 * w] | S.misc[17 + 2 * w]) {
        const int *bw = S.misc + 24 + 6 * w;
        bz0 = min(bz0, bw[0]); bz1 = max(bz1, bw[1]); by0 = min(by0, bw[2]); by1 = max(by1, bw[3]); bx0 = min(bx0, bw[4]); bx1 = max(bx1, bw[5]);
      }
    if (slot < R->ntask_cap) {
      int *rec = R->htask + (size_t)PP3_HREC * slot;
      rec[0] = g; rec[1] = sub; rec[2] = bz0 | (bz1 << 16); rec[3] = by0 | (by1 << 16); rec[4] = bx0 | (bx1 << 16);
#pragma unroll
      for (int k = 0; k < 2 * NW; k++) rec[5 + k] = S.misc[16 + k];
    }
#pragma unroll
    for (int k = 0; k < 2 * NW; k++) S.misc[16 + k] = 0;
    S.misc[4] = 0;
  };
  for (;;) {
    __syncthreads();                                  // the previous task's readers of the tables are done
    if (tid == 0) { S.misc[0] = b_live ? 1 : (a_live ? 0 : -1); S.misc[1] = 0; S.misc[5] = b_val.x; S.misc[6] = b_val.y; S.misc[7] = b_val.z; S.misc[8] = b_val.w; cur = b_val; advance(); }
    __syncthreads();
    const int state = S.misc[0];
    if (tid == 0) { flush_heavy(prev.x, prev.y); prev = cur; }
    if (state < 0) break;
    if (state == 0) continue;
    const int pk = __builtin_amdgcn_readfirstlane(S.misc[7]), tk = __builtin_amdgcn_readfirstlane(S.misc[8]);
    const int xb = pk & 1023, gy = (pk >> 10) & 1023, gz = pk >> 20, tx = tk & 1023, ty = (tk >> 10) & 1023, tz = tk >> 20;
    const int tile = (tz * G.T + ty) * G.T + tx;
    const int lox = tx * G.pt + G.nb - ppr, loy = ty * G.pt + G.nb - ppr, loz = tz * G.pt + G.nb - ppr;
    const int half = LISTED ? __builtin_amdgcn_readfirstlane(S.misc[6]) >> 24 : 0, wl = (A.xbw + 1) >> 1;   // (a half patch: wl cells, or what is left of the patch)
    const int px0 = lox + xb * A.xbw, px1 = min(px0 + A.xbw, lox + e);
    const int hx0 = half == 2 ? min(px0 + wl, px1) : px0, hx1 = half == 1 ? min(px0 + wl, px1) : px1;   // home cells [hx0, hx1)
    const int hz0 = loz + gz * PP3_HZ, hy0 = loy + gy * PP3_HY;
    const int WV = hx1 - hx0 + 5;                      // boundaries hx0 - 2 ... hx1 + 2 (clamped to the region)
    if (hx1 <= hx0) continue;                          // (the upper half of a patch cut short by the region's face)
    // ---- the offset table: this wavefront's three planes of twelve rows
    {
      // every lane loads its boundary of every row through a uniform row pointer + a 32-bit lane offset (rows outside the region: the
      // nearest row inside, so that no load is conditional, and zeros afterwards)
      const int col4 = 4 * (min(max(hx0 - ppr + lane, lox), lox + e) - lox);
      const int zb = hz0 - ppr + 3 * wvu, yb = hy0 - ppr;
      int o[RW];
#pragma unroll
      for (int u = 0; u < RW; u++) asm volatile("" : "=v"(o[u]));   // (defined for the compiler; only the lanes below read them)
      if (lane < PPL_TS) {                            // the 36 boundaries of a row: the other lanes request nothing
#pragma unroll
        for (int u = 0; u < RW; u++) {
          const int zz = min(max(zb + u / PPL_NRY, loz), loz + e - 1), yy = min(max(yb + u % PPL_NRY, loy), loy + e - 1);
          o[u] = __builtin_amdgcn_raw_buffer_load_b32(cs_rsrc, col4, (unsigned)(((zz * E + yy) * E + lox) * 4), 0);   // scalar row offset + lane offset: no vector address arithmetic
        }
      }
      int orv = 0;                                    // OR of the offsets a lane has seen: at the last boundary, of the rows' counts
      unsigned char *trow = S.T + RW * wvu * PPL_TS + lane;
#pragma unroll
      for (int u = 0; u < RW; u++) {
        const int zz = zb + u / PPL_NRY, yy = yb + u % PPL_NRY;
        if (!(zz >= loz && zz < loz + e && yy >= loy && yy < loy + e)) o[u] = 0;      // (uniform, and rare: patches on the region's faces)
        const int d = o[u] - __builtin_amdgcn_readfirstlane(o[u]);
        if (lane < PPL_TS) trow[u * PPL_TS] = (unsigned char)d;
        if (lane == 0) S.rowg[RW * wvu + u] = o[u];    // the row segment's first record
        orv |= d;
      }
      if (lane == WV - 1 && orv > 255) S.misc[1] = 1;   // (counts are >= 0: their OR exceeds 255 exactly when one of them does)
      // the rows' counts back from the table (this wavefront's own stores: in order), one row per lane, and their prefix: three vector
      // instructions per row above where carrying first / count / prefix through the scalar unit and two v_writelane took five
      {
        const unsigned char *const tr = S.T + (RW * wvu + lane) * PPL_TS;
        const int cnt = lane < RW ? (int)tr[WV - 1] : 0;
        const int inc = wave_scan_incl_i(cnt);
        if (lane < RW) S.lcum[RW * wvu + lane] = (unsigned short)(inc - cnt);
        if (lane == RW - 1) S.wtot[wvu] = inc;
        if (!LISTED) {
          // the staged records of the patch's two halves along x (boundaries 0 ... wl + 4 and wl ... WV - 1), should the whole be crowded: summed
          // here, on every task, because the rare branch that needs them is no place for two scans and a barrier (they cost the common path
          // fifty spilled scalars)
          const int wlh = (A.xbw + 1) >> 1;
          const int cl = lane < RW ? (int)tr[min(wlh + 2 * ppr, WV - 1)] : 0, cr = lane < RW ? cnt - (int)tr[min(wlh, WV - 1)] : 0;
          const int sl = wave_scan_incl_i(cl), sr = wave_scan_incl_i(cr);
          if (lane == RW - 1) { S.hsum[wvu] = sl; S.hsum[4 + wvu] = sr; }
        }
      }
      // P3M_PP_FAT_LIMIT (a test switch) lowers the longest row segment this pass takes
      if (R->fat_limit < 255 && lane < RW && (int)S.T[(RW * wvu + lane) * PPL_TS + WV - 1] > R->fat_limit) S.misc[1] = 1;
    }
    __syncthreads();
    const int4 wt = *reinterpret_cast<const int4 *>(S.wtot);
    const int Ptot = wt.x + wt.y + wt.z + wt.w;
    if (S.misc[1] != 0 || Ptot > PPL_PCAP) {          // uniform: a crowded region or a long row segment
      // the two halves' staged records (hsum; valid with the table: no segment beyond 255)
      bool halves = false;
      if (!LISTED && S.misc[1] == 0 && hx1 - hx0 > wl) {
        const int pl = S.hsum[0] + S.hsum[1] + S.hsum[2] + S.hsum[3], pr = S.hsum[4] + S.hsum[5] + S.hsum[6] + S.hsum[7];
        halves = pl <= PPL_PCAP && pr <= PPL_PCAP;
      }
      if (tid == 0) {
        if (halves) { const int s = atomicAdd(R->retrycount, 2); R->retry[s] = 2 * cur.x; R->retry[s + 1] = 2 * cur.x + 1; }
        else { const int s = atomicAdd(R->slowcount, 1); if (s < A.ngroups) R->slow[s] = cur.x; }   // k_pp_ext3's general pass 0 (never a half: see PPLRare)
      }
      continue;
    }
    {
      const int wbase = wvu == 0 ? 0 : wvu == 1 ? wt.x : wvu == 2 ? wt.x + wt.y : wt.x + wt.y + wt.z;
      if (lane < RW) S.cum16[RW * wvu + lane] = (unsigned short)((S.lcum[RW * wvu + lane] + wbase) << 4);
#if !defined(PPL_ABL) || PPL_ABL < 3
      // ---- staging: this wavefront's rows, (row, slot) items over the lanes
      const int rl = lane >> 3, k = lane & 7;
      constexpr int NIT = (RW + 7) / 8;
      float4 q[NIT];
      int cn[NIT], c0[NIT], g0[NIT];
      bool more = false;
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int r = RW * wvu + it * 8 + rl;
        const bool rok = it * 8 + 8 <= RW || it * 8 + rl < RW;
        cn[it] = rok ? (int)S.T[r * PPL_TS + WV - 1] : 0; g0[it] = S.rowg[r]; c0[it] = (int)S.lcum[r] + wbase;
#ifndef PPL_STAGE_GLOBAL   // (the scalar-base global load: 1.58 ms per tile against 1.49 with the descriptor, whose four scalars the allocator spills and reloads)
        if (k < cn[it]) q[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sp_rsrc, (g0[it] + k) * 16, 0, 0));
#else
        if (k < cn[it]) q[it] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(A.spos) + (size_t)((unsigned)(g0[it] + k) * 16u));   // scalar base + 32-bit lane offset
#endif
        more = more || cn[it] > 8;
      }
#pragma unroll
      for (int it = 0; it < NIT; it++) if (k < cn[it]) S.prec[c0[it] + k] = q[it];
      if (__any(more)) {
#pragma unroll 1
        for (int it = 0; it < NIT; it++)
          for (int kk = k + 8; kk < cn[it]; kk += 8) S.prec[c0[it] + kk] = A.spos[g0[it] + kk];
      }
#endif
      // ---- the home rows' counts (wavefront 0): exclusive prefix
      if (wv == 0) {
        const int rj = ((lane >> 3) + ppr) * PPL_NRY + (lane & 7) + ppr;
        const int hc = (int)S.T[rj * PPL_TS + ppr + (hx1 - hx0)] - (int)S.T[rj * PPL_TS + ppr];
        const int inc = wave_scan_incl_i(hc);
        S.roff[lane] = inc - hc;
        if (lane == 63) S.roff[NH] = inc;
      }
    }
    __syncthreads();
    const int total = S.roff[NH];                     // <= Ptot <= 704: tasks of PP3_NT home records (pp3_task_homes)
    if (total == 0) continue;
#if defined(PPL_ABL) && PPL_ABL >= 2   // timing-only ablation builds (tools/variant.sh): 1 no evaluation, 2 nor lists, 3 nor the records' staging
    continue;
#endif
    // sub-tasks of PP3_NT home records (total <= 704: pp3_task_homes = PP3_NT, the numbering k_pp_ext3<.., 1> expects); a second one in two
    // patches of a hundred at the background's density.  Nothing below shares LDS between lanes but the heavy-lane record
#pragma unroll 1
    for (int sub = 0; sub * PP3_NT < total; sub++) {
    if (sub > 0) {
      __syncthreads();
      if (tid == 0) { flush_heavy(cur.x, (cur.y & ~0xffffff) | (sub - 1)); prev.y = (cur.y & ~0xffffff) | sub; }
      __syncthreads();
    }
    const int hraw = sub * PP3_NT + tid;
    const bool valid = hraw < total;
    const int h = min(hraw, total - 1);               // a lane without a home record repeats the last one's lists (and kicks nothing)
    // the home row of record h: the last row whose prefix is <= h, found plane first, then row (two LDS round trips of eight independent
    // values each; a bisection is six dependent ones, and the task is a chain of round trips before it is anything else)
    int j;
    {
      int jzc = 0, jyc = 0;
#pragma unroll
      for (int m = 1; m < PP3_HZ; m++) jzc += S.roff[m * PP3_HY] <= h ? 1 : 0;
#pragma unroll
      for (int m = 1; m < PP3_HY; m++) jyc += S.roff[jzc * PP3_HY + m] <= h ? 1 : 0;
      j = jzc * PP3_HY + jyc;
    }
    const int jz = j >> 3, jy = j & 7, cz = hz0 + jz, cy = hy0 + jy;
    const int rh = (jz + ppr) * PPL_NRY + jy + ppr;
    const unsigned self16 = (unsigned)S.cum16[rh] + (((unsigned)S.T[rh * PPL_TS + ppr] + (unsigned)(h - S.roff[j])) << 4);
    const float4 p = *reinterpret_cast<const float4 *>(pbytes + self16);
    const int cx = (int)floorf(p.x) + G.nb;                                     // :412
    auto is_phys = [&](int ccx, int ccy, int ccz) {                             // :576-582
      return ccx >= lox + ppr && ccx < lox + ppr + G.pt && ccy - loy >= ppr && ccy - loy < ppr + G.pt && ccz - loz >= ppr && ccz - loz < ppr + G.pt;
    };
    const bool phys = valid && is_phys(cx, cy, cz);
    const int vi = rec_index(p);
    float4 vrec = make_float4(0.f, 0.f, 0.f, 0.f);
#ifndef PPL_NOVEL
    if (phys) vrec = A.vel[vi];                        // needed after the sums
#endif
    // the lane's first window (dz = dy = -2): boundary cx - 2 of row (jz, jy) of the table; window (dz, dy) lies (dz + 2) * 12 + dy + 2 rows on
    const unsigned char *tb = S.T + (jz * PPL_NRY + jy) * PPL_TS + (cx - hx0);
    const unsigned short *cb = S.cum16 + jz * PPL_NRY + jy;
    // the reference's half-shell sweep starts only from the planes k <= pt + pp_range (:496): a record of the planes above sweeps downwards only
    const bool rim_task = hz0 + PP3_HZ - 1 - loz >= G.pt + ppr;                 // uniform
    const int dzmax = cz - loz >= G.pt + ppr ? loz + G.pt + ppr - 1 - cz : ppr;
    unsigned loff = mybase;
    int cmax = 0;
    // a window: its first SIX entries in one 12-byte store whatever its count (the next window overwrites what was not a partner); a
    // window of more than six records (1e-6 at the background's density) sends the wavefront through build_slow
    auto append = [&](unsigned a16, int cnt) {
      const unsigned w0 = __umul24(a16, 0x10001u) + 0x100000u;                   // entries a16, a16 + 16
      const unsigned w1 = w0 + 0x200020u, w2 = w0 + 0x400040u;                  //         ... a16 + 80
      // six 16-bit stores (ds_write_b16 / _d16_hi: the three registers serve two each).  One 12-byte store at this 2-byte alignment is
      // legal and was measured: the LDS then stalls on the misalignment for most of the kernel's time (SQ_LDS_UNALIGNED_STALL 589 M of
      // 963 M busy cycles, profiles/r06_pp_lds_counters.txt)
      lds_vu16 *const lp = (lds_vu16 *)(lbytes + min(loff, mycap));          // (volatile: the compiler merges plain stores back into one misaligned store)
      // the first entry by every lane; the other five only by the lanes whose window holds more than one record (13 % at the background's
      // density): 64 lanes storing to 64 unrelated addresses conflict four or five deep on the banks, and the LDS -- not the vector units --
      // was what the kernel waited for (SQ_LDS_BANK_CONFLICT 363 M of 657 M busy cycles with six full-width stores per window)
      lp[0] = (unsigned short)w0;
      if (cnt > 1) { lp[1] = (unsigned short)(w0 >> 16); lp[2] = (unsigned short)w1; lp[3] = (unsigned short)(w1 >> 16); lp[4] = (unsigned short)w2; lp[5] = (unsigned short)(w2 >> 16); }
      loff += 2 * cnt;
      cmax = max(cmax, cnt);
    };
    // the general form, rolled: any counts, the half-shell rule of the rim planes (tasks of a tile's last patch along z).  A lane stores
    // what a list holds and no more (next to a blob a window has a hundred records and the lane is heavy anyway), but counts everything
    auto build_slow = [&](bool mine) {
      loff = mybase;
#pragma unroll 1
      for (int dz = -ppr; dz <= ppr; dz++)
#pragma unroll 1
        for (int dy = -ppr; dy <= ppr; dy++) {
          const int ro = (dz + ppr) * PPL_NRY + dy + ppr;
          const int wa = tb[ro * PPL_TS], wb = mine && dz <= dzmax ? (int)tb[ro * PPL_TS + 2 * ppr + 1] : wa;
          const bool ownrow = dz == 0 && dy == 0;
          const int o0 = ownrow ? min((int)tb[ro * PPL_TS + ppr], wb) : wb, o1 = ownrow ? min((int)tb[ro * PPL_TS + ppr + 1], wb) : wb;
          const unsigned c16 = ((lds_vu16 *)cb)[ro];
#pragma unroll
          for (int half = 0; half < 2; half++) {       // the own cell [o0, o1) splits the own row's window (:515-516)
            const int a = half == 0 ? wa : o1, b = half == 0 ? o0 : wb;
            for (int v = a; v < b && loff + 2 * (unsigned)(v - a) <= mycap + 10; v++) {
              const unsigned short en = (unsigned short)(c16 + ((unsigned)v << 4));
              *(lds_vu16 *)(lbytes + loff + 2 * (v - a)) = en;
            }
            loff += 2 * max(b - a, 0);
          }
        }
    };
#ifndef PPL_NOBUILD
    if (!rim_task) {
      // (the table reads of two planes requested together before the appends -- 30 loads in flight instead of the compiler's four -- were
      // measured: no faster, 77 more live registers and spills)
#pragma unroll
      for (int dz = -ppr; dz <= ppr; dz++) {
#pragma unroll
        for (int dy = -ppr; dy <= ppr; dy++) {
          const int ro = (dz + ppr) * PPL_NRY + dy + ppr;
          const int wa = tb[ro * PPL_TS], wb = tb[ro * PPL_TS + 2 * ppr + 1];
          const unsigned c16 = ((lds_vu16 *)cb)[ro];   // (volatile: five neighbours would be merged into misaligned wide reads)
          if (dz == 0 && dy == 0) {                    // the own cell splits the own row's window (:515-516)
            const int o0 = tb[ro * PPL_TS + ppr], o1 = tb[ro * PPL_TS + ppr + 1];
            append(c16 + ((unsigned)wa << 4), o0 - wa);
            append(c16 + ((unsigned)o1 << 4), wb - o1);
          } else append(c16 + ((unsigned)wa << 4), wb - wa);
        }
      }
    }
    {
      // a list that is USED and has a window of more than six records (or any list of a rim task): the general loop for those lanes
      const bool redo = rim_task || (cmax > 6 && (int)(loff - mybase) <= 2 * PPL_LCAP);
      if (__any(redo)) { const unsigned keep = loff; build_slow(redo); if (!redo) loff = keep; }
    }
#endif
    const int n_all = (int)(loff - mybase) >> 1;
    bool heavy = valid && n_all > PPL_LCAP;
    bool walker = false;
    unsigned long long hb = __ballot(heavy);
    if (hb) {                                         // uniform per wavefront, and rare away from blobs
      if (__popcll(hb) <= 2 && wave_max_i(heavy ? n_all : 0) <= 2 * PPL_LCAP) { walker = heavy; heavy = false; hb = 0ull; }
      else {
        int z0 = max(cz - ppr, loz), z1 = min(min(cz + ppr, loz + e - 1), cz + dzmax);
        const int y0 = max(cy - ppr, loy), y1 = min(cy + ppr, loy + e - 1), x0 = max(cx - ppr, lox), x1 = min(cx + ppr, lox + e - 1);
        const int bz0 = wave_min_i(heavy ? z0 : 0x7fff), bz1 = wave_max_i(heavy ? z1 : 0), by0 = wave_min_i(heavy ? y0 : 0x7fff), by1 = wave_max_i(heavy ? y1 : 0);
        const int bx0 = wave_min_i(heavy ? x0 : 0x7fff), bx1 = wave_max_i(heavy ? x1 : 0);
        if (lane == 0) {
          S.misc[16 + 2 * wv] = (int)(unsigned)hb; S.misc[17 + 2 * wv] = (int)(unsigned)(hb >> 32); S.misc[4] = 1;
          int *bw = S.misc + 24 + 6 * wv;
          bw[0] = bz0; bw[1] = bz1; bw[2] = by0; bw[3] = by1; bw[4] = bx0; bw[5] = bx1;
        }
      }
    }
    const int n = (valid && !heavy && !walker) ? n_all : 0;
    { lds_vu16 *const lp = (lds_vu16 *)(lbytes + mybase + 2 * n);   // the tail of the last trip: the home record itself
      lp[0] = (unsigned short)self16; lp[1] = (unsigned short)self16; lp[2] = (unsigned short)self16; }
    f32x2 ax2 = {0.f, 0.f}, ay2 = {0.f, 0.f}, az2 = {0.f, 0.f};
#if defined(PPL_ABL) && PPL_ABL == 1
    if (n < -1)
#endif
    for (int k = 0;; k += 4) {
      const bool okm = k < n;
      if (!__any(okm)) break;
      const uint2 e4 = *reinterpret_cast<const uint2 *>(lbytes + mybase + 2 * k);
      const float4 o0 = *reinterpret_cast<const float4 *>(pbytes + (e4.x & 0xffffu)), o1 = *reinterpret_cast<const float4 *>(pbytes + (e4.x >> 16));
      const float4 o2 = *reinterpret_cast<const float4 *>(pbytes + (e4.y & 0xffffu)), o3 = *reinterpret_cast<const float4 *>(pbytes + (e4.y >> 16));
      pp_ext_eval2s<TAPER_ALL>(p, o0, o1, okm, F, ax2, ay2, az2);
      pp_ext_eval2s<TAPER_ALL>(p, o2, o3, okm, F, ax2, ay2, az2);
    }
    float ax = ax2.x + ax2.y, ay = ay2.x + ay2.y, az = az2.x + az2.y;
    // ---- -DPPINT on the way: the pairs inside the own cell (physical records of cells that are the reference's buckets: k_pp_intra)
    float aix = 0.f, aiy = 0.f, aiz = 0.f;
    bool idone = false;
    if (FUSE) {
      if (phys && !heavy) {                           // (a tile's physical cells lie inside the rank's volume: k_pp_intra's own test)
        const float fms = (float)A.ms, ims = 1.0f / fms;
        const bool pow2 = (A.ms & (A.ms - 1)) == 0;   // (x / mesh_scale is x * (1 / mesh_scale) bit for bit when mesh_scale is a power of two: k_pp_intra)
        const int c0 = (int)floorf(pow2 ? p.x * ims : p.x / fms), c1 = (int)floorf(pow2 ? p.y * ims : p.y / fms), c2 = (int)floorf(pow2 ? p.z * ims : p.z / fms);
        const int Ec = E / A.ms, cbf = G.nb / A.ms;
        idone = A.cflag[((c2 + cbf) * Ec + (c1 + cbf)) * Ec + (c0 + cbf)] == 0;
      }
      const int ro = ppr * PPL_NRY + ppr;
      const int ob0 = tb[ro * PPL_TS + ppr], ob1 = tb[ro * PPL_TS + ppr + 1];
      const unsigned own16 = (unsigned)((lds_vu16 *)cb)[ro] + ((unsigned)ob0 << 4);
      const int nown = idone ? ob1 - ob0 : 0;         // (the home record itself among them: r = 0, below the soft cut)
      for (int k = 0; __any(k < nown); k++) {
        const float4 o = *reinterpret_cast<const float4 *>(pbytes + (k < nown ? own16 + 16 * k : self16));
        const float sx = p.x - o.x, sy = p.y - o.y, sz = p.z - o.z;                 // :336
        const float r2 = sx * sx + sy * sy + sz * sz;
        const float ir = __builtin_amdgcn_rsqf(r2);
        float f = F.K * ((ir * ir) * ir);                                           // mass_p / (r pp_bias)^3 (:344)
        f = (k < nown && r2 >= F.r2_soft) ? f : 0.0f;                               // :340
        aix = __builtin_fmaf(-sx, f, aix); aiy = __builtin_fmaf(-sy, f, aiy); aiz = __builtin_fmaf(-sz, f, aiz);   // :346-347
      }
      if (idone) {
        // the record's sorted index (k_pp_intra's): its row segment's first record + the records of the row before its cell + its rank
        const int sidx = S.rowg[rh] + (int)S.T[rh * PPL_TS + ppr] + (h - S.roff[j]);
        A.intra_done[sidx] = 1;
      }
    }
    if (__any(walker)) {                              // one or two lanes of the wavefront with 33 ... 64 partners: they walk their windows
      if (walker) {
        const unsigned own0 = (unsigned)cb[2 * PPL_NRY + 2 + 0] + ((unsigned)tb[(2 * PPL_NRY + 2) * PPL_TS + ppr] << 4);
        const unsigned own1 = (unsigned)cb[2 * PPL_NRY + 2 + 0] + ((unsigned)tb[(2 * PPL_NRY + 2) * PPL_TS + ppr + 1] << 4);
        for (int dz = -ppr; dz <= dzmax; dz++)
          for (int dy = -ppr; dy <= ppr; dy++) {
            const int ro = (dz + ppr) * PPL_NRY + dy + ppr;
            const unsigned c16 = cb[ro];
            unsigned a = c16 + ((unsigned)tb[ro * PPL_TS] << 4);
            const unsigned b = c16 + ((unsigned)tb[ro * PPL_TS + 2 * ppr + 1] << 4);
            for (; a < b; a += 16) {
              if (dz == 0 && dy == 0 && a >= own0 && a < own1) continue;          // own cell is excluded (:515-516)
              const float4 o = *reinterpret_cast<const float4 *>(pbytes + a);
              pp_ext_eval_l(p, o.x, o.y, o.z, F, ax, ay, az);
            }
          }
      }
    }
    float mag = 0.f;
    if (valid && !heavy) {
      if (phys) {                                                                   // :576-582
        float4 v = vrec;
        if (idone) { v.x = v.x + aix * A.a_mid * P3M_G_F * A.dt; v.y = v.y + aiy * A.a_mid * P3M_G_F * A.dt; v.z = v.z + aiz * A.a_mid * P3M_G_F * A.dt; }   // :349-350, before the extended kick as there
        v.x = v.x + ax * A.a_mid * P3M_G_F * A.dt; v.y = v.y + ay * A.a_mid * P3M_G_F * A.dt; v.z = v.z + az * A.a_mid * P3M_G_F * A.dt;
#ifdef PPL_NOVEL   // ablation: no velocity read-modify-write
        if (v.x == 1.2345f)
#endif
        A.vel[vi] = v;
      }
      mag = sqrtf(ax * ax + ay * ay + az * az);                                     // :617
    }
    mag = wave_max_nonneg_to_last(mag);
    if (lane == 63 && mag > 0.f) p3m_atomic_max_nonneg(R->tile_max + tile, mag);
    if (FUSE) {                                       // pp_force_max (:356): one record in nine has a bucket mate at the background's density
      float magi = sqrtf(aix * aix + aiy * aiy + aiz * aiz);
      if (__any(magi > 0.f)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) magi = fmaxf(magi, __shfl_xor(magi, o, 64));
        if (lane == 0) p3m_atomic_max_nonneg(R->fmax_intra + p3m_slot() * 16, magi);
      }
    }
    }
  }
}

// the patches k_pp_light left to the general pass: their tasks (k_pp_plan3 + k_pp_fill2 for a list of patches; one wavefront per patch, one
// lane per home row)
__global__ __launch_bounds__(256) void k_pp_plan_slow(const int *__restrict__ cs, PPGeo G, int npz, int npy, int npx, int xbw, const int *__restrict__ slow, const int *__restrict__ slowcount,
                                                      int ngroups, int4 *__restrict__ task4, int *__restrict__ ntask, int cap) {
  const int nslow = min(*slowcount, ngroups), j = threadIdx.x & 63;
  const int e = G.pt + 2 * G.ppr;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nslow; i += gridDim.x * 4) {
    const int g = slow[i];
    const int xb = g % npx, gy = (g / npx) % npy, gz = (g / (npx * npy)) % npz, tile = g / (npx * npy * npz);
    const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
    const int lox = tx * G.pt + G.nb - G.ppr, loy = ty * G.pt + G.nb - G.ppr, loz = tz * G.pt + G.nb - G.ppr;
    const int hx0 = lox + xb * xbw, hx1 = min(hx0 + xbw, lox + e);
    int count = 0;
    const int rz = gz * PP3_HZ + j / PP3_HY, ry = gy * PP3_HY + j % PP3_HY;
    if (rz < e && ry < e) { const int64_t rb = ((int64_t)(loz + rz) * G.E + (loy + ry)) * G.E; count = cs[rb + hx1] - cs[rb + hx0]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o, 64);
    const int nt = pp3_task_homes(count), n = (count + nt - 1) / nt;
    int base = 0;
    if (j == 0 && n > 0) base = atomicAdd(ntask, n);
    base = __shfl(base, 0, 64);
    for (int k = j; k < n; k += 64) if (base + k < cap) task4[base + k] = make_int4(g, k, 0, 0);
  }
}

// the smallest float r2 with sqrtf(r2) > t (sqrtf is correctly rounded and monotone): "rmag > t" becomes "r2 >= this"
static float first_r2_with_root_above(float t) {
  float r2 = t * t;
  while (sqrtf(r2) > t) r2 = nextafterf(r2, 0.0f);
  while (!(sqrtf(r2) > t)) r2 = nextafterf(r2, INFINITY);
  return r2;
}

int pp_extended(p3m_ctx *c, float a_mid, float dt, float mass_p, bool fuse_intra) {
  P3M_TRY(particles_full_cells(c));
  c->pp_intra_fused = false;
  const Geometry &g = c->g;
  if (g.pp_range == 0) return P3M_OK;
  PPGeo G{g.T, g.nb, g.pt, g.E, g.Nn, g.ms, g.pp_range, c->p.rsoft, c->p.pp_bias, (float)g.ncut};
  const int e = g.pt + 2 * g.pp_range;
  // P3M_PP_EXT_REF=1: the plain kernel k_pp_ext -- one lane per record of a row, partners straight from global memory, the force
  // in the reference's own arithmetic (sqrt and divisions, :551-571) -- instead of the LDS-staged kernel with its reciprocal
  // square root and fused multiply-adds.  A test switch (tests/test_gpu_parity.py FALLBACKS): both have to meet the oracle
  static const bool ref_arith = getenv("P3M_PP_EXT_REF") && getenv("P3M_PP_EXT_REF")[0] == '1';
  if (ref_arith) {
    const unsigned blocks = (unsigned)((int64_t)g.ntiles * e * e);
    hipLaunchKernelGGL(k_pp_ext, dim3(blocks), dim3(64), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, mass_p, a_mid,
                       dt, c->d_tile_ext);
    HIP_TRY(hipGetLastError());
    return P3M_OK;
  }
  // patches of PP3_HZ x PP3_HY rows x xbw cells holding 7/8 of PP3_NT home records at the mean density (one task)
  const double rho_mean = (double)c->np_all / ((double)g.E * g.E * g.E);
  int xbw = (int)std::lround((0.875 * PP3_NT) / std::max(1e-9, rho_mean * PP3_HZ * PP3_HY));
  // the lean light pass (k_pp_light): the reference's reach, record indices that fit a 32-bit byte offset; P3M_PP_LIGHT_OFF=1 (a test
  // switch) sends every task through the general pass 0 of k_pp_ext3, which otherwise only works the tasks the lean pass leaves to it
  static const bool light_off = getenv("P3M_PP_LIGHT_OFF") && getenv("P3M_PP_LIGHT_OFF")[0] == '1';
  const bool light = !light_off && g.pp_range == 2 && (int64_t)c->cap < (1ll << 27) && (int64_t)g.E * g.E * g.E < (1ll << 29);   // (32-bit byte offsets into both arrays)
  xbw = std::max(4, std::min(std::min(xbw, e), light ? PPL_XBW_MAX : 64 - 2 * g.pp_range - 1));      // one load instruction per partner row
  const int npx = (e + xbw - 1) / xbw, npy = (e + PP3_HY - 1) / PP3_HY, npz = (e + PP3_HZ - 1) / PP3_HZ;
  const int64_t ngroups64 = (int64_t)g.ntiles * npz * npy * npx;
  // a record is a home record of every tile whose extended region holds its cell: per axis at most 2 + 2*ppr/pt tiles
  const int64_t mult1 = std::min<int64_t>(g.T, 2 + (2 * g.pp_range) / g.pt), mult = mult1 * mult1 * mult1;
  const int64_t ngroups_max = (int64_t)g.ntiles * npz * npy * ((e + 3) / 4);
  const int64_t ntask_cap64 = mult * (c->cap / PP3_NTD + 1) + ngroups_max + 64;
  if (ngroups_max > 0x3fffffff || ntask_cap64 > 0x7fffffff || npx > 1023 || npy > 1023 || npz > 1023 || g.T > 1023) { p3m_set_error("extended PP: too many patches"); return P3M_EINVAL; }
  const int ngroups = (int)ngroups64, ntask_cap = (int)ntask_cap64;
  // each buffer under its own check: a failed allocation must not leave the others looking ready
  if (!c->pp_plan) HIP_TRY(hipMalloc(&c->pp_plan, sizeof(int) * ((size_t)ngroups_max + 8)));
  if (!c->pp_task_group) HIP_TRY(hipMalloc(&c->pp_task_group, sizeof(int4) * (size_t)ntask_cap64));   // the task records (k_pp_fill2)
  if (!c->pp_slow) HIP_TRY(hipMalloc(&c->pp_slow, sizeof(int) * (3 * (size_t)ngroups_max + 24)));   // the patches the lean light pass leaves to the general one | its retry list (two halves per patch)
  if (!c->pp_htask) HIP_TRY(hipMalloc(&c->pp_htask, sizeof(int) * PP3_HREC * (size_t)ntask_cap64));    // the heavy-task list (a task enters it once at most)
  // task counters of the three launches (PP3_NSEG each, on cache lines of their own), the length of the heavy-task list and of the slow-task list
  constexpr int NCNT = 4 * 32 * PP3_NSEG + 128;
  if (!c->pp_counter) HIP_TRY(hipMalloc(&c->pp_counter, sizeof(int) * NCNT));
  P3M_TRY(scan_reserve(c, ngroups_max + 8));
  HIP_TRY(hipMemsetAsync(c->pp_counter, 0, sizeof(int) * NCNT, c->stream));
  int *hcount = c->pp_counter + 4 * 32 * PP3_NSEG, *slowcount = hcount + 32, *slow_ntask = hcount + 64, *retrycount = hcount + 96;
  if (!light) {                                        // the plan of every patch: tasks of PP3_NT (PP3_NTD in a blob) home records
    hipLaunchKernelGGL(k_pp_plan3, dim3(cdiv(ngroups, 4)), dim3(256), 0, c->stream, (const int *)c->cell_end, G, npy, npx, xbw, ngroups, c->pp_plan);
    HIP_TRY(hipGetLastError());
    P3M_TRY(exclusive_scan_i32(c, c->pp_plan, ngroups));
    hipLaunchKernelGGL(k_pp_fill2, dim3(cdiv(ngroups, 256)), dim3(256), 0, c->stream, (const int *)c->pp_plan, ngroups, reinterpret_cast<int4 *>(c->pp_task_group), ntask_cap, npz, npy, npx, g.T);
    HIP_TRY(hipGetLastError());
  }
  const PPForce F = pp_force_constants(mass_p, G.pp_bias, G.ncut, first_r2_with_root_above(G.rsoft), first_r2_with_root_above(G.ncut + sqrtf(3.0f)));
  const int Wp = (xbw + 2 * g.pp_range + 2) & ~1, NRmax = (PP3_HZ + 2 * g.pp_range) * (PP3_HY + 2 * g.pp_range);
  const size_t lds = sizeof(float4) * PP3_PCAP + sizeof(unsigned short) * (size_t)NRmax * Wp + (size_t)(PP3_NT / 64) * 64 * PP3_LSTR +
                     sizeof(int) * ((size_t)2 * NRmax + 1 + 2 * PP3_HZ * PP3_HY + 1 + 48);   // 48: misc
  // a partner row segment of more than 65 534 records does not fit the 16-bit offsets: its task takes the per-lane path over
  // global memory.  P3M_PP_FAT_LIMIT=n lowers the limit (a test switch: ordinary inputs then run that path)
  static const int fat_limit = getenv("P3M_PP_FAT_LIMIT") ? std::max(1, std::min(65534, atoi(getenv("P3M_PP_FAT_LIMIT")))) : 65534;
  // no two records within reach of each other are further apart than sqrt(3) (pp_range + 1) cells: when that is inside the taper's
  // range (the rule: nf_cutoff = 16) the taper needs no switch
  const float far2 = 3.0f * (float)((g.pp_range + 1) * (g.pp_range + 1));
  const bool taper_all = far2 < F.r2_taper;
  const int *plan_total = c->pp_plan + ngroups;        // the scan's total: the number of tasks
  if (light) {
    int *const retry = c->pp_slow + ngroups_max + 8;
    PPLRare rare{c->pp_counter + 2 * 32 * PP3_NSEG, c->pp_htask, hcount, c->pp_slow, slowcount, c->d_tile_ext, c->d_red + 1 * P3M_RED_SPAN, ntask_cap, fat_limit,
                 npx, npx * npy, npx * npy * npz, fdiv_magic(npx), fdiv_magic(npx * npy), fdiv_magic(g.T), fdiv_magic(g.T * g.T), retry, retrycount, nullptr, nullptr};
    PPLArgs A{(const float4 *)c->spos, c->vel, (const int *)c->cell_end, rare, nullptr, (const unsigned char *)c->cflag,
              g.T, g.nb, g.pt, g.E, g.ms, xbw, ngroups, F.c1, F.K, F.K34, F.K74, F.r2_soft, F.r2_taper, a_mid, dt};
    // -DPPINT in the same pass (NGP builds): built, at parity, and NOT the default -- P3M_PP_INTRA_FUSED=1 turns it on.  Measured on the
    // headline's geometry with PPINT + PP_EXT: 46.4 ms per step fused against 45.9 with k_pp_intra's own pass over every record.  The
    // light pass pays 0.13 ms per tile for it (a coarse-cell flag gather, a byte store and three more sums per lane in a kernel that is
    // out of registers already) and k_pp_intra, which still has to find the records NOT done, only falls from 0.26 to 0.15 ms: a row of
    // 70 records holds six ghosts that no patch ever visits, so no wavefront of it can leave early
    static const bool fused = getenv("P3M_PP_INTRA_FUSED") && getenv("P3M_PP_INTRA_FUSED")[0] == '1';
    if (fuse_intra && fused) {
      if (!c->pp_intra_done) HIP_TRY(hipMalloc(&c->pp_intra_done, (size_t)c->cap + 16));
      HIP_TRY(hipMemsetAsync(c->pp_intra_done, 0, (size_t)c->np_all, c->stream));
      A.intra_done = c->pp_intra_done;
      c->pp_intra_fused = true;
    }
    const int wpl = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / sizeof(PPLShared)));
    auto kl = A.intra_done ? (taper_all ? k_pp_light<true, true, false> : k_pp_light<false, true, false>) : (taper_all ? k_pp_light<true, false, false> : k_pp_light<false, false, false>);
    auto kl2 = A.intra_done ? (taper_all ? k_pp_light<true, true, true> : k_pp_light<false, true, true>) : (taper_all ? k_pp_light<true, false, true> : k_pp_light<false, false, true>);
    hipLaunchKernelGGL(kl, dim3(256 * wpl), dim3(PP3_NT), 0, c->stream, A);
    HIP_TRY(hipGetLastError());
    // the crowded patches whose halves fit, as half patches (none at the background's density: the launch then costs its dispatch)
    A.rare.counter = c->pp_counter + 3 * 32 * PP3_NSEG; A.rare.list = retry; A.rare.listcount = retrycount;
    hipLaunchKernelGGL(kl2, dim3(256 * wpl), dim3(PP3_NT), 0, c->stream, A);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_pp_plan_slow, dim3(128), dim3(256), 0, c->stream, (const int *)c->cell_end, G, npz, npy, npx, xbw, (const int *)c->pp_slow, (const int *)slowcount, ngroups,
                       reinterpret_cast<int4 *>(c->pp_task_group), slow_ntask, ntask_cap);
    HIP_TRY(hipGetLastError());
  }
  const int wpc = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / lds));   // resident workgroups per CU by LDS
  for (int pass = 0; pass < 2; pass++) {               // resident workgroups per CU: by LDS, the heavy pass also by its registers
    auto kern = pass == 0 ? (g.pp_range == 2 ? (taper_all ? k_pp_ext3<2, true, 0> : k_pp_ext3<2, false, 0>) : (taper_all ? k_pp_ext3<0, true, 0> : k_pp_ext3<0, false, 0>))
                          : (g.pp_range == 2 ? (taper_all ? k_pp_ext3<2, true, 1> : k_pp_ext3<2, false, 1>) : (taper_all ? k_pp_ext3<0, true, 1> : k_pp_ext3<0, false, 1>));
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // pass 0: the whole plan, or (behind the lean light pass) the tasks it left
    hipLaunchKernelGGL(kern, dim3(256 * (pass == 0 ? wpc : std::min(wpc, PP3_WPE - 1))), dim3(PP3_NT), lds, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, F, a_mid, dt,
                       c->d_tile_ext, (const int4 *)c->pp_task_group, light ? (const int *)slow_ntask : plan_total, npy, npx, xbw, ntask_cap,
                       c->pp_counter + pass * 32 * PP3_NSEG, Wp, NRmax, fat_limit, c->pp_htask, hcount);
    HIP_TRY(hipGetLastError());
  }
  return P3M_OK;
}


// ------------------------------------------------------------------ measurement hook (bench.py: pairs/s of the two PP kernels)
// Pair EVALUATIONS as the kernels perform them: every kicked record sums over all its partners, so a pair of two
// kicked records is evaluated twice (the reference's loops visit it once and update both members).
//   intra    : sum over physical fine cells of n*(n-1)
//   extended : sum over the records of every tile's extended region of the records in their partner cells (same
//              clipping and the same half-shell reach as k_pp_ext)
__global__ __launch_bounds__(256) void k_pp_count_intra(const float4 *__restrict__ spos, const int *__restrict__ cs, int n, PPGeo G, unsigned long long *__restrict__ out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  unsigned long long c = 0;
  if (s < n) {
    const float4 p = spos[s];
    const float fNn = (float)G.Nn;
    if (p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn) {
      const int bx = (int)floorf(p.x) + G.nb, by = (int)floorf(p.y) + G.nb, bz = (int)floorf(p.z) + G.nb;
      const int64_t cell = ((int64_t)bz * G.E + by) * G.E + bx;
      c = (unsigned long long)(cs[cell + 1] - cs[cell] - 1);
    }
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
__global__ __launch_bounds__(64) void k_pp_count_ext(const float4 *__restrict__ spos, const int *__restrict__ cs, PPGeo G, unsigned long long *__restrict__ out) {
  const int e = G.pt + 2 * G.ppr;
  const int ry = blockIdx.x % e, rz = (blockIdx.x / e) % e, tile = blockIdx.x / (e * e);
  const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
  const int lox = tx * G.pt + G.nb - G.ppr, loy = ty * G.pt + G.nb - G.ppr, loz = tz * G.pt + G.nb - G.ppr;
  const int cy = loy + ry, cz = loz + rz;
  const int64_t rowb = ((int64_t)cz * G.E + cy) * G.E;
  const int p0 = cs[rowb + lox], p1 = cs[rowb + lox + e];
  unsigned long long c = 0;
  for (int s = p0 + threadIdx.x; s < p1; s += 64) {
    const int cx = (int)floorf(spos[s].x) + G.nb;
    int z0 = max(cz - G.ppr, loz), z1 = min(cz + G.ppr, loz + e - 1);
    if (cz - loz >= G.pt + G.ppr) z1 = min(z1, loz + G.pt + G.ppr - 1);
    const int y0 = max(cy - G.ppr, loy), y1 = min(cy + G.ppr, loy + e - 1);
    const int x0 = max(cx - G.ppr, lox), x1 = min(cx + G.ppr, lox + e - 1);
    for (int zz = z0; zz <= z1; zz++)
      for (int yy = y0; yy <= y1; yy++) {
        const int64_t rb = ((int64_t)zz * G.E + yy) * G.E;
        c += (unsigned long long)(cs[rb + x1 + 1] - cs[rb + x0]);
        if (zz == cz && yy == cy) c -= (unsigned long long)(cs[rb + cx + 1] - cs[rb + cx]);
      }
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if (threadIdx.x == 0 && c) atomicAdd(out, c);
}

extern "C" int p3m_hip_time_pp(p3m_ctx *c, float a_mid, float dt, float mass_p, int32_t reps, float *ms_intra, float *ms_ext, int64_t *evals_intra,
                               int64_t *evals_ext) {
  if (!c || reps < 1 || !ms_intra || !ms_ext || !evals_intra || !evals_ext) return P3M_EINVAL;
  P3M_TRY(need_particles(c, "p3m_hip_time_pp"));
  HIP_TRY(hipSetDevice(c->device));
  P3M_TRY(particles_full_cells(c));
  const Geometry &g = c->g;
  PPGeo G{g.T, g.nb, g.pt, g.E, g.Nn, g.ms, g.pp_range, c->p.rsoft, c->p.pp_bias, (float)g.ncut};
  unsigned long long *d_cnt = nullptr, h_cnt[2] = {0, 0};
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  float a = 0.f, b = 0.f;
  auto body = [&]() -> int {   // every early return leaves through the clean-up below
    HIP_TRY(hipMalloc(&d_cnt, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), c->stream));
    if (c->np_all > 0) {
      hipLaunchKernelGGL(k_pp_count_intra, dim3(cdiv(c->np_all, 256)), dim3(256), 0, c->stream, (const float4 *)c->spos, (const int *)c->cell_end, c->np_all, G, d_cnt);
      const int e = g.pt + 2 * g.pp_range;
      if (g.pp_range > 0)
        hipLaunchKernelGGL(k_pp_count_ext, dim3((unsigned)((int64_t)g.ntiles * e * e)), dim3(64), 0, c->stream, (const float4 *)c->spos, (const int *)c->cell_end, G, d_cnt + 1);
    }
    HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2));
    P3M_TRY(pp_intra(c, a_mid, dt, mass_p)); P3M_TRY(pp_extended(c, a_mid, dt, mass_p, false));   // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; i++) P3M_TRY(pp_intra(c, a_mid, dt, mass_p));
    HIP_TRY(hipEventRecord(e1, c->stream));
    for (int i = 0; i < reps; i++) P3M_TRY(pp_extended(c, a_mid, dt, mass_p, false));
    HIP_TRY(hipEventRecord(e2, c->stream));
    HIP_TRY(hipEventSynchronize(e2));
    HIP_TRY(hipEventElapsedTime(&a, e0, e1)); HIP_TRY(hipEventElapsedTime(&b, e1, e2));
    if (getenv("P3M_PP_STATS") && c->pp_counter && c->pp_plan) {   // diagnostic: tasks of the last launch, those left to the general pass, heavy tasks
      int h[3] = {0, 0, 0};
      HIP_TRY(hipMemcpy(h, c->pp_counter + 4 * 32 * PP3_NSEG, sizeof(int), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(h + 1, c->pp_counter + 4 * 32 * PP3_NSEG + 32, sizeof(int), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(h + 2, c->pp_counter + 4 * 32 * PP3_NSEG + 96, sizeof(int), hipMemcpyDeviceToHost));
      fprintf(stderr, "[pp stats] heavy tasks %d, patches left to the general pass %d, half patches %d\n", h[0], h[1], h[2]);
#ifdef PP_SWEEP_STATS
      { unsigned long long q[4]; HIP_TRY(hipMemcpy(q, c->pp_counter + 4 * 32 * PP3_NSEG + 8, sizeof(q), hipMemcpyDeviceToHost));
        fprintf(stderr, "[pp stats] sweep: %llu partners swept per wavefront-row visit x 64 lanes = %.3g lane slots, %llu row visits, %llu accepted (incl. own cell); tight stretches %llu\n", q[0], 64.0 * (double)q[0], q[1], q[2], q[3]); }
#endif
      if (h[0] > 0 && c->pp_htask) {   // heavy lanes per heavy task
        std::vector<int> rec((size_t)h[0] * PP3_HREC);
        HIP_TRY(hipMemcpy(rec.data(), c->pp_htask, rec.size() * sizeof(int), hipMemcpyDeviceToHost));
        long hist[10] = {0}, lanes = 0; const int edge[10] = {1, 2, 4, 8, 16, 32, 64, 128, 192, 257};
        for (int t = 0; t < h[0]; t++) {
          int n = 0;
          for (int k = 0; k < 8; k++) n += __builtin_popcount((unsigned)rec[(size_t)t * PP3_HREC + 5 + k]);
          lanes += n;
          for (int b = 0; b < 10; b++) if (n <= edge[b]) { hist[b]++; break; }
        }
        fprintf(stderr, "[pp stats] heavy lanes %ld; tasks by heavy lanes (<=1 2 4 8 16 32 64 128 192 256):", lanes);
        for (int b = 0; b < 10; b++) fprintf(stderr, " %ld", hist[b]);
        fprintf(stderr, "\n");
      }
    }
    return P3M_OK;
  };
  const int rc = body();
  if (rc != P3M_OK) (void)hipStreamSynchronize(c->stream);   // h_cnt lives on this frame
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e2) (void)hipEventDestroy(e2);
  if (d_cnt) (void)hipFree(d_cnt);
  P3M_TRY(rc);
  *ms_intra = a / reps; *ms_ext = b / reps; *evals_intra = (int64_t)h_cnt[0]; *evals_ext = (int64_t)h_cnt[1];
  return P3M_OK;
}
