// pp.hip -- short-range particle-particle forces on the cell-sorted records.
//   pp_intra    : -DPPINT, particle_mesh_threaded.f90:274-285 (bucketing) + :324-361 (pairs)
//   pp_extended : -DPP_EXT, particle_mesh_threaded.f90:378-624
// The sorted order (records of one fine cell contiguous, cells of one x-row contiguous) replaces
// llf / hoc_fine / ll_fine: a cell's partners are index ranges.  FP32-ALU bound (about 20 flop per
// pair incl. the reciprocal square root); reported as pairs/s, not against the HBM roofline.
#include "p3m_internal.h"
#include <algorithm>
#include <cmath>

struct PPGeo { int T, nb, pt, E, Nn, ms, ppr; float rsoft, pp_bias, ncut; };

__device__ __forceinline__ float3 pair_force(const float4 &a, const float4 &b, float mass_p, float rsoft, float pp_bias) {
  // :336-344 : sep = x1-x2 ; rmag ; if (rmag>rsoft) force_pp = mass_p*(sep/(rmag*pp_bias)**3)
  const float sx = a.x - b.x, sy = a.y - b.y, sz = a.z - b.z;
  const float rmag = sqrtf(sx * sx + sy * sy + sz * sz);
  if (!(rmag > rsoft)) return make_float3(0.f, 0.f, 0.f);
  const float rb = rmag * pp_bias, rb3 = rb * rb * rb;
  return make_float3(mass_p * (sx / rb3), mass_p * (sy / rb3), mass_p * (sz / rb3));
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
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

// Dense cells: when a wavefront's records sit in cells of more than PP_INTRA_DENSE records, the wavefront walks the union of
// its lanes' cell ranges (contiguous in the sorted order) 64 partners at a time -- one coalesced load, then one broadcast
// (v_readlane) per partner, each lane keeping the partners of its own cell -- instead of every lane streaming its whole
// cell from global memory.  Same partner order per lane (ascending sorted index) as the per-lane loop.
#define PP_INTRA_DENSE 12
__global__ __launch_bounds__(256) void k_pp_intra(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs,
                                                  const unsigned char *__restrict__ cflag, int n, PPGeo G, float mass_p, float a_mid, float dt,
                                                  float *__restrict__ fmax_out, float r2_soft) {
  const int s = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  float mag = 0.f;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  bool phys = false, slow = false;
  int cc[3] = {0, 0, 0}, sub[3] = {0, 0, 0}, q0 = 0, q1 = 0;
  // The records of a fine cell are neighbours in the sorted order: a record's cell mates are found by comparing cell indices along
  // the wavefront (plus the record before and the one after it) instead of two look-ups in cell_end per record, which at the mean
  // density cost four times the bytes of the records themselves.  Only a run that crosses the wavefront's ends reads cell_end
  int64_t cell = -1, ncell = -1;
  {
    const int t = lane == 0 ? s - 1 : s + 1;           // the neighbours beyond the wavefront's ends
    if ((lane == 0 || lane == 63) && t >= 0 && t < n) {
      const float4 o = spos[t];
      ncell = ((int64_t)((int)floorf(o.z) + G.nb) * G.E + ((int)floorf(o.y) + G.nb)) * G.E + ((int)floorf(o.x) + G.nb);
    }
  }
  if (s < n) {
    p = spos[s];
    const float fNn = (float)G.Nn;
    phys = p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn;
    cell = ((int64_t)((int)floorf(p.z) + G.nb) * G.E + ((int)floorf(p.y) + G.nb)) * G.E + ((int)floorf(p.x) + G.nb);
    if (phys) {
      // the hoc coarse cell alone decides the path (ref_bucket's three float and six integer divisions were a third of this
      // kernel's instructions; x / mesh_scale is x * (1 / mesh_scale) bit for bit when mesh_scale is a power of two)
      const float fms = (float)G.ms, ims = 1.0f / fms;
      const bool pow2 = (G.ms & (G.ms - 1)) == 0;
      cc[0] = (int)floorf(pow2 ? p.x * ims : p.x / fms); cc[1] = (int)floorf(pow2 ? p.y * ims : p.y / fms); cc[2] = (int)floorf(pow2 ? p.z * ims : p.z / fms);
      const int Ec = G.E / G.ms, cb = G.nb / G.ms;
      slow = cflag[((cc[2] + cb) * Ec + (cc[1] + cb)) * Ec + (cc[0] + cb)] != 0;
      if (slow) ref_bucket(p, G, cc, sub);             // the sub-cell is only compared on the slow path
    }
  }
  {
    const int64_t before = __shfl_up(cell, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || cell != before);            // bit l: a run of equal cells starts at lane l
    const bool open_l = __shfl(cell == ncell ? 1 : 0, 0, 64) != 0, open_r = __shfl(cell == ncell ? 1 : 0, 63, 64) != 0;
    const int start = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));
    const unsigned long long after = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
    const int end = after ? __ffsll((long long)after) - 1 : 64;
    if (phys && !slow) {
      if ((start == 0 && open_l) || (end == 64 && open_r)) { q0 = cs[cell]; q1 = cs[cell + 1]; }
      else { const int s0 = s - lane; q0 = s0 + start; q1 = s0 + end; }
    }
  }
  float ax = 0.f, ay = 0.f, az = 0.f;
  const bool fast = phys && !slow;
  const int maxc = wave_max_i(fast ? q1 - q0 : 0);
  if (maxc > PP_INTRA_DENSE) {
    const int Q0 = wave_min_i(fast ? q0 : 0x7fffffff), Q1 = wave_max_i(fast ? q1 : 0);
    const float ibias = 1.0f / G.pp_bias;
    for (int base = Q0; base < Q1; base += 64) {
      const int m = min(64, Q1 - base);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < m) o = spos[base + lane];
      for (int jj = 0; jj < m; jj++) {
        const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), jj));
        const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), jj));
        const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), jj));
        const int q = base + jj;
        if (fast && q >= q0 && q < q1 && q != s) {
          const float sx = p.x - px, sy = p.y - py, sz = p.z - pz;                 // :336
          const float r2 = sx * sx + sy * sy + sz * sz;
          if (r2 >= r2_soft) {                                                     // :340 rmag > rsoft, decided exactly on r^2
            const float ib = __builtin_amdgcn_rsqf(r2) * ibias, irb3 = ib * ib * ib;
            ax -= mass_p * (sx * irb3); ay -= mass_p * (sy * irb3); az -= mass_p * (sz * irb3);   // :344-347
          }
        }
      }
    }
  } else if (fast) {
    for (int q = q0; q < q1; q++) {
      if (q == s) continue;
      const float3 f = pair_force(p, spos[q], mass_p, G.rsoft, G.pp_bias);
      ax -= f.x; ay -= f.y; az -= f.z;                            // :346-347
    }
  }
  // Records of flagged coarse cells (some record's reference bucket differs from its sorted cell): partners are the records
  // of the whole coarse cell with the same reference bucket.  Wavefront-cooperative: one flagged coarse cell at a time, its
  // ms*ms x-rows 64 candidates at a time, each candidate's bucket computed once by the lane that loaded it and broadcast
  // with its position (a thousand-particle cell made every lane stream and re-bucket the whole coarse cell on its own).
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
  if (phys) {
    const int vi = __float_as_int(spos[s].w); float4 v = vel[vi];   // the velocity stays in arrival order (p3m_internal.h)
    v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;  // :349-350
    vel[vi] = v;
    mag = sqrtf(ax * ax + ay * ay + az * az);                       // :356
  }
  for (int o = 32; o > 0; o >>= 1) mag = fmaxf(mag, __shfl_down(mag, o, 64));
  if ((threadIdx.x & 63) == 0 && mag > 0.f) p3m_atomic_max_nonneg(fmax_out + p3m_slot() * 16, mag);
}

static float first_r2_with_root_above(float t);
int pp_intra(p3m_ctx *c, float a_mid, float dt, float mass_p) {
  P3M_TRY(particles_full_cells(c));
  const Geometry &g = c->g;
  if (c->np_all == 0) return P3M_OK;
  PPGeo G{g.T, g.nb, g.pt, g.E, g.Nn, g.ms, g.pp_range, c->p.rsoft, c->p.pp_bias, (float)g.ncut};
  hipLaunchKernelGGL(k_pp_intra, dim3(cdiv(c->np_all, 256)), dim3(256), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end,
                     (const unsigned char *)c->cflag, c->np_all, G, mass_p, a_mid, dt, c->d_red + 1 * P3M_RED_SPAN, first_r2_with_root_above(G.rsoft));
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

// The same sums from LDS: one 256-thread workgroup per block of PB_X x PB_Y x PB_Z home cells of a tile's extended
// region.  The records of the block's halo (home rows +- pp_range, x range +- pp_range) and the cell offsets of
// those row segments are staged in LDS with coalesced loads; every home record then walks its partners -- same
// rows and cells as k_pp_ext -- without touching global memory.  A block whose
// halo does not fit the staging area (strong clustering) falls back to global loads for what was not staged.
#define PB_Y 4
#define PB_Z 4
#define PPT_CAP 2048    // staged records per block
#define PP_LPH 2        // lanes per home record
__global__ __launch_bounds__(256) void k_pp_ext_tiled(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, PPGeo G,
                                                      float mass_p, float a_mid, float dt, float *__restrict__ tile_max, int bx_cells, int nbx, int nby,
                                                      float r2_soft, float r2_taper) {
  extern __shared__ int lds_i[];
  const int ppr = G.ppr, e = G.pt + 2 * ppr;
  const int HR = (PB_Y + 2 * ppr) * (PB_Z + 2 * ppr);        // halo rows
  const int wseg = bx_cells + 2 * ppr + 1;                     // cell offsets per staged row segment
  int *rp0 = lds_i;                 // [HR] first record of the staged segment (global sorted index)
  int *rcnt = rp0 + HR;             // [HR] records staged of it
  int *roff = rcnt + HR;            // [HR] their offset in lp
  int *rtot = roff + HR;            // [HR] records in the segment
  int *hpre = rtot + HR;            // [PB_Y*PB_Z + 1] prefix of home record counts
  int *cseg = hpre + PB_Y * PB_Z + 1;   // [HR][wseg] cell offsets relative to rp0
  float4 *lp = reinterpret_cast<float4 *>(cseg + ((HR * wseg + 3) & ~3));
  __shared__ float wmax[4];
  int b = blockIdx.x;
  const int ibx = b % nbx; b /= nbx;
  const int iby = b % nby; b /= nby;
  const int nbz = nby, ibz = b % nbz, tile = b / nbz;
  const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
  const int lox = tx * G.pt + G.nb - ppr, loy = ty * G.pt + G.nb - ppr, loz = tz * G.pt + G.nb - ppr;   // extended region [lo, lo+e)
  const int hx0 = lox + ibx * bx_cells, hx1 = min(hx0 + bx_cells, lox + e);      // home cells in x
  const int hy0 = loy + iby * PB_Y, hz0 = loz + ibz * PB_Z;
  const int sx0 = max(hx0 - ppr, lox), sx1 = min(hx1 + ppr, lox + e);           // staged cells in x
  const int ny = PB_Y + 2 * ppr;
  // ---- stage: segment ranges and cell offsets
  for (int r = threadIdx.x; r < HR; r += 256) {
    const int yy = hy0 - ppr + r % ny, zz = hz0 - ppr + r / ny;
    int p0 = 0, p1 = 0;
    if (yy >= loy && yy < loy + e && zz >= loz && zz < loz + e) { const int64_t rb = ((int64_t)zz * G.E + yy) * G.E; p0 = cs[rb + sx0]; p1 = cs[rb + sx1]; }
    rp0[r] = p0; rtot[r] = p1 - p0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int r = 0; r < HR; r++) { roff[r] = run; const int c = min(rtot[r], PPT_CAP - run); rcnt[r] = c; run += c; }
    hpre[PB_Y * PB_Z] = run;   // borrowed until the home prefix is built: number of staged records
  }
  for (int r = threadIdx.x >> 6; r < HR; r += 4) {   // one wavefront per row segment: no per-element index divisions
    const int ry_ = r % ny, yy = hy0 - ppr + ry_, zz = hz0 - ppr + r / ny;
    const bool inside = yy >= loy && yy < loy + e && zz >= loz && zz < loz + e;
    const int *src = cs + ((int64_t)zz * G.E + yy) * G.E + sx0;
    const int base = rp0[r];
    for (int k = threadIdx.x & 63; k < wseg; k += 64) cseg[r * wseg + k] = (inside && sx0 + k <= sx1) ? src[k] - base : 0;
  }
  __syncthreads();
  {
    const int nst = hpre[PB_Y * PB_Z];
    for (int t = threadIdx.x; t < nst; t += 256) {   // staged record t belongs to the last row r with roff[r] <= t
      int lo_r = 0, hi_r = HR - 1;
      while (lo_r < hi_r) { const int mid = (lo_r + hi_r + 1) >> 1; if (roff[mid] <= t) lo_r = mid; else hi_r = mid - 1; }
      lp[t] = spos[rp0[lo_r] + (t - roff[lo_r])];
    }
  }
  __syncthreads();
  // home records: rows (jy, jz) of the block, cells [hx0, hx1)
  if (threadIdx.x == 0) {
    int run = 0;
    for (int hr = 0; hr < PB_Y * PB_Z; hr++) {
      hpre[hr] = run;
      const int r = (hr / PB_Y + ppr) * ny + (hr % PB_Y + ppr);
      const int yy = hy0 + hr % PB_Y, zz = hz0 + hr / PB_Y;
      if (yy < loy + e && zz < loz + e && hx1 > hx0) run += cseg[r * wseg + (hx1 - sx0)] - cseg[r * wseg + (hx0 - sx0)];
    }
    hpre[PB_Y * PB_Z] = run;
  }
  __syncthreads();
  const int nhome = hpre[PB_Y * PB_Z];
  float mymax = 0.f;
  auto partner = [&](int r, int q) -> float4 { const int k = q - rp0[r]; return k < rcnt[r] ? lp[roff[r] + k] : spos[q]; };
  // PP_LPH lanes share one home record: lane u takes the partner rows u, u+PP_LPH, ... of the (zz,yy) sweep and walks
  // them with a flat cursor (a wavefront pays for its busiest lane, not for the busiest lane of every row); the
  // partial sums are added across the lanes at the end.
  const float incut = 1.0f / G.ncut, ibias = 1.0f / G.pp_bias;
  const int nround = (nhome * PP_LPH + 255) / 256;
  for (int rd = 0; rd < nround; rd++) {
    const int t = rd * 256 + (int)threadIdx.x, hidx = t / PP_LPH, u = t - hidx * PP_LPH;
    const bool live = hidx < nhome;
    float ax = 0.f, ay = 0.f, az = 0.f;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0, cx = 0, cy = 0, cz = 0;
    if (live) {
      int hr = 0;
#pragma unroll
      for (int k = 1; k < PB_Y * PB_Z; k++) hr += (hidx >= hpre[k]) ? 1 : 0;
      const int jy = hr % PB_Y, jz = hr / PB_Y;
      const int r_own = (jz + ppr) * ny + (jy + ppr);
      s = rp0[r_own] + cseg[r_own * wseg + (hx0 - sx0)] + (hidx - hpre[hr]);
      p = partner(r_own, s);
      cy = hy0 + jy; cz = hz0 + jz;
      cx = (int)floorf(p.x) + G.nb;                                // :412 (floor(xv)+1, global)
      int z0 = max(cz - ppr, loz), z1 = min(cz + ppr, loz + e - 1);
      // the reference's half-shell sweep starts only from the planes k <= pt + pp_range (:496): a pair whose two cells
      // both lie in the top pp_range planes of the region is never formed
      if (cz - loz >= G.pt + ppr) z1 = min(z1, loz + G.pt + ppr - 1);
      const int y0 = max(cy - ppr, loy), y1 = min(cy + ppr, loy + e - 1);
      const int x0 = max(cx - ppr, lox), x1 = min(cx + ppr, lox + e - 1);
      const int nyr = y1 - y0 + 1, nrow = (z1 - z0 + 1) * nyr;
      int zi = 0, yi = u - PP_LPH, q = 0, q1 = 0, s0 = 0, s1 = 0, r = 0;   // (zi, yi): the row ordinal u, u+PP_LPH, ... as a mixed-radix counter
      const int nzr = z1 - z0 + 1;
      (void)nrow;
      for (;;) {
        if (q >= q1) {
          yi += PP_LPH;
          while (yi >= nyr) { yi -= nyr; zi++; }
          if (zi >= nzr) break;
          const int zz = z0 + zi, yy = y0 + yi;
          r = (zz - hz0 + ppr) * ny + (yy - hy0 + ppr);
          const int *cr = cseg + r * wseg - sx0;
          q = rp0[r] + cr[x0]; q1 = rp0[r] + cr[x1 + 1];
          const bool own = (zz == cz && yy == cy);
          s0 = own ? rp0[r] + cr[cx] : 0; s1 = own ? rp0[r] + cr[cx + 1] : 0;   // own cell is excluded (:515-516)
          continue;
        }
        if (q >= s0 && q < s1) { q = s1; continue; }
        const float4 o = partner(r, q);
        q++;
        const float sx = p.x - o.x, sy = p.y - o.y, sz = p.z - o.z;            // :551
        const float r2 = sx * sx + sy * sy + sz * sz;
        // rmag = sqrt(r2) > rsoft decided EXACTLY on r2 (r2_soft = the smallest float whose correctly rounded root exceeds
        // rsoft, found on the host), the magnitudes from the hardware reciprocal square root: no IEEE sqrt or division in
        // the pair loop (about a third of its instructions); the force differs from sep/(rmag*pp_bias)^3 by a few ulp
        if (r2 >= r2_soft) {                                                    // :558
          const float ir = __builtin_amdgcn_rsqf(r2), rb1 = (r2 * ir) * G.pp_bias, ib = ir * ibias, irb3 = ib * ib * ib;
          float fx = mass_p * (sx * irb3), fy = mass_p * (sy * irb3), fz = mass_p * (sz * irb3);
          if (r2 < r2_taper) {                                                  // :559-564, not (rmag > ncut + sqrt(3))
            const float qq = rb1 * incut;
            const float taper = 1.f - (7.0f / 4.0f) * (qq * qq * qq) + (3.0f / 4.0f) * (qq * qq * qq * qq * qq);
            fx *= taper; fy *= taper; fz *= taper;
          }
          ax -= fx; ay -= fy; az -= fz;                                         // :571
        }
      }
    }
#pragma unroll
    for (int o = 1; o < PP_LPH; o <<= 1) { ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); az += __shfl_xor(az, o, 64); }
    if (live && u == 0) {
      const int ry = cy - loy, rz = cz - loz;
      const bool phys = (cx >= lox + ppr && cx < lox + ppr + G.pt && ry >= ppr && ry < ppr + G.pt && rz >= ppr && rz < ppr + G.pt);
      if (phys) {                                                                   // :576-582
        const int vi = __float_as_int(spos[s].w); float4 v = vel[vi];   // the velocity stays in arrival order (p3m_internal.h)
        v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;
        vel[vi] = v;
      }
      mymax = fmaxf(mymax, sqrtf(ax * ax + ay * ay + az * az));                     // :617
    }
  }
  for (int o = 32; o > 0; o >>= 1) mymax = fmaxf(mymax, __shfl_down(mymax, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mymax;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m4 = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m4 > 0.f) p3m_atomic_max_nonneg(tile_max + tile, m4);
  }
}

// ------------------------------------------------------------------ extended PP, version 2: one wavefront per 64 home records
// No block-level staging, no barriers between wavefronts: at the reference's density (1/8 particle per fine cell, ~15
// partners per record) k_pp_ext_tiled spent 94 % of its instructions on staging and on walking 25 mostly empty row
// windows per record (rocprofv3: 530 lane-instructions per pair evaluation), and with strong clustering a few workgroups
// did all the work.  Here a TASK is 64 consecutive home records of a group of PP_RG x-rows of one tile's extended region;
// k_pp_plan counts the tasks of every group, a scan turns the counts into first-task numbers, k_pp_fill writes the
// task -> group table, and persistent wavefronts draw tasks from a counter (dense tasks take thousands of times longer
// than sparse ones: no static split balances them).  One lane per home record:
//   sparse path: the lane collects the sorted indices of its partners (the clipped row windows of k_pp_ext, read straight
//     from cell_end through L1/L2) in a list in LDS, then sums over the list -- every lane busy with its own partners;
//   dense path (a lane's list would overflow): the wavefront walks the union of its lanes' windows row by row, loads 64
//     partners at a time (one per lane, coalesced) and broadcasts them one by one (v_readlane); each lane keeps the
//     partners inside its own window.  Neighbouring home records share almost all partners, so each partner is loaded once
//     per 64 home records.
#define PP_RG 16
#define PP_LCAP 40
#define PP_CHUNK 4
#define PP_NSEG 64
__global__ __launch_bounds__(256) void k_pp_plan(const int *__restrict__ cs, PPGeo G, int ngy, int nxb, int xbw, int ngroups, int *__restrict__ plan) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const int e = G.pt + 2 * G.ppr;
  const int xb = g % nxb, gy = (g / nxb) % ngy, rz = (g / (nxb * ngy)) % e, tile = g / (nxb * ngy * e);
  const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
  const int lox = tx * G.pt + G.nb - G.ppr, loy = ty * G.pt + G.nb - G.ppr, loz = tz * G.pt + G.nb - G.ppr;
  const int hx0 = lox + xb * xbw, hx1 = min(hx0 + xbw, lox + e);   // the group's home cells
  int count = 0;
  for (int j = 0; j < PP_RG; j++) {
    const int ry = gy * PP_RG + j;
    if (ry < e) { const int64_t rb = ((int64_t)(loz + rz) * G.E + (loy + ry)) * G.E; count += cs[rb + hx1] - cs[rb + hx0]; }
  }
  plan[g] = (count + 63) >> 6;
}
__global__ __launch_bounds__(256) void k_pp_fill(const int *__restrict__ plan, int ngroups, int *__restrict__ task_group, int cap) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const int k1 = min(plan[g + 1], cap);
  for (int k = plan[g]; k < k1; k++) task_group[k] = g;
}
__global__ __launch_bounds__(256) void k_pp_fill2(const int *__restrict__ plan, int ngroups, int2 *__restrict__ task2, int cap) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const int k0 = plan[g], k1 = min(plan[g + 1], cap);
  for (int k = k0; k < k1; k++) task2[k] = make_int2(g, k - k0);
}
struct PPForce { float mass_p, pp_bias, ibias, incut, r2_soft, r2_taper; };
// One partner of the extended sweep.  The hard cut (:558) is decided on r^2 computed in the reference's order; the force itself is
// formed with fused multiply-adds and the taper in Horner form, 26 instructions instead of 37 (the kick differs from the
// reference's association in the last bit: 1e-7 relative against the 1e-5 bar), without branches: a lane outside the cut adds zero
__device__ __forceinline__ void pp_ext_eval(const float4 &p, float ox, float oy, float oz, const PPForce &F, float &ax, float &ay, float &az) {
  const float sx = p.x - ox, sy = p.y - oy, sz = p.z - oz;               // :551
  const float r2 = sx * sx + sy * sy + sz * sz;
  const float ir = __builtin_amdgcn_rsqf(r2), qq = (r2 * ir) * (F.pp_bias * F.incut), ib = ir * F.ibias;
  const float q2 = qq * qq, q3 = q2 * qq;
  float taper = __builtin_fmaf(q3, __builtin_fmaf(0.75f, q2, -1.75f), 1.0f);   // 1 - 7/4 q^3 + 3/4 q^5 (:559-564)
  taper = r2 < F.r2_taper ? taper : 1.0f;
  float f = (F.mass_p * taper) * (ib * ib * ib);
  f = r2 >= F.r2_soft ? f : 0.0f;                                        // :558, decided exactly on r^2 (see k_pp_ext_tiled)
  ax = __builtin_fmaf(-sx, f, ax); ay = __builtin_fmaf(-sy, f, ay); az = __builtin_fmaf(-sz, f, az);   // :571
}
// Two partners at once in the halves of packed registers (v_pk_add / v_pk_mul / v_pk_fma_f32: one instruction slot for both):
// 17 slots per partner.  A half that is not a partner (okA / okB false) adds zero.  Every half goes through the operations of
// pp_ext_eval; a home record's sum is formed as two partial sums, added at the end
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pp_ext_eval2(const float4 &p, const float4 &A, const float4 &B, bool okA, bool okB, const PPForce &F,
                                             f32x2 &ax, f32x2 &ay, f32x2 &az) {
  const f32x2 ox = {A.x, B.x}, oy = {A.y, B.y}, oz = {A.z, B.z};
  const f32x2 sx = p.x - ox, sy = p.y - oy, sz = p.z - oz;               // :551
  const f32x2 r2 = sx * sx + sy * sy + sz * sz;
  const f32x2 ir = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
  const f32x2 qq = (r2 * ir) * (F.pp_bias * F.incut), ib = ir * F.ibias;
  const f32x2 q2 = qq * qq, q3 = q2 * qq;
  const f32x2 c34 = {0.75f, 0.75f}, c74 = {-1.75f, -1.75f}, one = {1.0f, 1.0f};
  f32x2 taper = __builtin_elementwise_fma(q3, __builtin_elementwise_fma(c34, q2, c74), one);   // :559-564
  taper.x = r2.x < F.r2_taper ? taper.x : 1.0f; taper.y = r2.y < F.r2_taper ? taper.y : 1.0f;
  f32x2 f = (F.mass_p * taper) * (ib * ib * ib);
  f.x = (okA && r2.x >= F.r2_soft) ? f.x : 0.0f; f.y = (okB && r2.y >= F.r2_soft) ? f.y : 0.0f;   // :558
  ax = __builtin_elementwise_fma(-sx, f, ax); ay = __builtin_elementwise_fma(-sy, f, ay); az = __builtin_elementwise_fma(-sz, f, az);   // :571
}
template <int PPR>   // PPR > 0: pp_range known at compile time (the reference's default 2); 0: any
__global__ __launch_bounds__(64) void k_pp_ext2(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, PPGeo G, PPForce F,
                                                float a_mid, float dt, float *__restrict__ tile_max, const int *__restrict__ plan,
                                                const int *__restrict__ task_group, int ngroups, int ngy, int nxb, int xbw, int ntask_cap, int *__restrict__ counter) {
  __shared__ int list[PP_LCAP][64];
  __shared__ int rstart[PP_RG], roff[PP_RG];
  const int lane = threadIdx.x;
  const int ppr = G.ppr, e = G.pt + 2 * ppr, E = G.E;
  const int ntask = min(plan[ngroups], ntask_cap);
  // PP_NSEG task counters on cache lines of their own, each handing out the tasks of one contiguous segment (atomics on ONE
  // address serialise at ~12 ns each: 145 000 fetches from a single counter were half of this kernel's run time); a
  // wavefront starts at its own segment and moves on round robin when a segment runs dry
  const int per = (ntask + PP_NSEG - 1) / PP_NSEG;
  int seg = blockIdx.x % PP_NSEG;
  for (int tried = 0; tried < PP_NSEG;) {
    const int sbeg = min(seg * per, ntask), send = min(sbeg + per, ntask);
    int tf = 0;
    if (lane == 0) {   // a plain (agent-coherent) load first: a dry segment costs no read-modify-write
      tf = __hip_atomic_load(counter + 32 * seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tf < send - sbeg) tf = atomicAdd(counter + 32 * seg, PP_CHUNK);
    }
    tf = sbeg + __builtin_amdgcn_readfirstlane(tf);
    if (tf >= send) { seg = (seg + 1) % PP_NSEG; tried++; continue; }
    tried = 0;
    const int tl = min(tf + PP_CHUNK, send);
    for (int t = tf; t < tl; t++) {
      const int g = task_group[t], sub = t - plan[g];
      const int xb = g % nxb, gy = (g / nxb) % ngy, rz = (g / (nxb * ngy)) % e, tile = g / (nxb * ngy * e);
      const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
      const int lox = tx * G.pt + G.nb - ppr, loy = ty * G.pt + G.nb - ppr, loz = tz * G.pt + G.nb - ppr;
      const int cz = loz + rz;
      const int hx0 = lox + xb * xbw, hx1 = min(hx0 + xbw, lox + e);   // the group's home cells: a patch of PP_RG rows x xbw cells
      // the group's rows: first record and exclusive prefix of the home counts
      int cnt = 0, st = 0;
      if (lane < PP_RG) {
        const int ry = gy * PP_RG + lane;
        if (ry < e) { const int64_t rb = ((int64_t)cz * E + (loy + ry)) * E; st = cs[rb + hx0]; cnt = cs[rb + hx1] - st; }
      }
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < PP_RG; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
      const int total = __shfl(inc, PP_RG - 1, 64);
      __syncthreads();                                    // the previous task's readers of rstart / roff / list are done
      if (lane < PP_RG) { rstart[lane] = st; roff[lane] = inc - cnt; }
      __syncthreads();
      const int h = sub * 64 + lane;
      const bool valid = h < total;
      int j = 0;
#pragma unroll
      for (int k = 1; k < PP_RG; k++) j += (roff[k] <= h) ? 1 : 0;
      const int s = valid ? rstart[j] + (h - roff[j]) : 0;
      const float4 p = valid ? spos[s] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int cx = valid ? (int)floorf(p.x) + G.nb : lox + ppr;                // :412
      const int cy = loy + gy * PP_RG + (valid ? j : 0);
      int z0 = max(cz - ppr, loz), z1 = min(cz + ppr, loz + e - 1);               // uniform over the wavefront
      // the reference's half-shell sweep starts only from the planes k <= pt + pp_range (:496): see k_pp_ext
      if (cz - loz >= G.pt + ppr) z1 = min(z1, loz + G.pt + ppr - 1);
      const int y0 = max(cy - ppr, loy), y1 = min(cy + ppr, loy + e - 1);
      const int x0 = max(cx - ppr, lox), x1 = min(cx + ppr, lox + e - 1);
      float ax = 0.f, ay = 0.f, az = 0.f;
      // ---- sparse attempt: list the partners
      int n = 0;
      if (PPR > 0) {
        // compile-time reach: every window of the (2 PPR + 1)^2 partner rows is loaded before the first one is used
        constexpr int NW = PPR > 0 ? (2 * PPR + 1) * (2 * PPR + 1) : 1, ND = 2 * PPR + 1;
        int wa[NW], wb[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const int zz = cz - PPR + w / ND, yy = cy - PPR + w % ND;
          const bool ok = valid && zz >= z0 && zz <= z1 && yy >= y0 && yy <= y1;
          const int64_t rb = ((int64_t)(ok ? zz : cz) * E + (ok ? yy : cy)) * E;
          wa[w] = 0; wb[w] = 0;
          if (ok) { wa[w] = cs[rb + x0]; wb[w] = cs[rb + x1 + 1]; }
        }
        int s0 = 0, s1 = 0;
        if (valid) { const int64_t rb = ((int64_t)cz * E + cy) * E; s0 = cs[rb + cx]; s1 = cs[rb + cx + 1]; }   // own cell is excluded (:515-516)
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const bool own = (w == NW / 2);
          for (int q = wa[w]; q < wb[w]; q++) {
            if (own && q >= s0 && q < s1) { q = s1 - 1; continue; }
            if (n < PP_LCAP) list[n][lane] = q;
            n++;
          }
        }
      } else {
        for (int zz = z0; zz <= z1; zz++)
          for (int dy = -ppr; dy <= ppr; dy++) {
            const int yy = cy + dy;
            const bool yok = valid && yy >= y0 && yy <= y1;
            const int64_t rb = ((int64_t)zz * E + (yok ? yy : cy)) * E;
            int a = 0, b = 0, s0 = 0, s1 = 0;
            if (yok) { a = cs[rb + x0]; b = cs[rb + x1 + 1]; }
            if (yok && zz == cz && dy == 0) { s0 = cs[rb + cx]; s1 = cs[rb + cx + 1]; }    // own cell is excluded (:515-516)
            for (int q = a; q < b; q++) {
              if (q >= s0 && q < s1) { q = s1 - 1; continue; }
              if (n < PP_LCAP) list[n][lane] = q;
              n++;
            }
          }
      }
      // lanes whose partners fit their list sum over it; the others (records in or next to dense cells) are served in
      // GROUPS of lanes that are close in x -- a blob's members -- by the broadcast path over the union of the group's
      // windows only: a row that crosses a blob holds a few dozen blob members and a few dozen background records spread
      // over hundreds of cells, and one union over all of them would test every lane against every record of 25 whole rows
      const bool big = n > PP_LCAP;
      const int nl = big ? 0 : n;
      const int nmax = wave_max_i(nl);
      for (int k = 0; k < nmax; k += 4) {   // four partners in flight at a time
        float4 o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { o[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (k + u < nl) o[u] = spos[list[k + u][lane]]; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (k + u < nl) pp_ext_eval(p, o[u].x, o[u].y, o[u].z, F, ax, ay, az);
      }
      bool todo = big;
      unsigned long long pend;
      while ((pend = __ballot(todo)) != 0ull) {
        const int lead = __ffsll((long long)pend) - 1;
        const int xa = __builtin_amdgcn_readlane(cx, lead);
        const bool mine = todo && cx >= xa - 4 && cx <= xa + 8;          // the group: dense lanes within a dozen cells of the first one
        const int Y0 = wave_min_i(mine ? y0 : 0x7fffffff), Y1 = wave_max_i(mine ? y1 : -1);
        const int X0 = wave_min_i(mine ? x0 : 0x7fffffff), X1 = wave_max_i(mine ? x1 : -1);
        for (int zz = z0; zz <= z1; zz++)
          for (int yy = Y0; yy <= Y1; yy++) {
            const int64_t rb = ((int64_t)zz * E + yy) * E;
            const int A = cs[rb + X0], B = cs[rb + X1 + 1];
            const bool rowok = mine && yy >= y0 && yy <= y1;
            const bool ownrow = (zz == cz && yy == cy);
            for (int base = A; base < B; base += 64) {
              const int m = min(64, B - base);
              float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
              if (lane < m) o = spos[base + lane];
              const int ocx = (int)floorf(o.x) + G.nb;
              for (int jj = 0; jj < m; jj++) {
                const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), jj));
                const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), jj));
                const float pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), jj));
                const int pcx = __builtin_amdgcn_readlane(ocx, jj);
                if (rowok && pcx >= x0 && pcx <= x1 && !(ownrow && pcx == cx)) pp_ext_eval(p, px, py, pz, F, ax, ay, az);
              }
            }
          }
        todo = todo && !mine;
      }
      float mag = 0.f;
      if (valid) {
        const int ry = cy - loy;
        const bool phys = (cx >= lox + ppr && cx < lox + ppr + G.pt && ry >= ppr && ry < ppr + G.pt && rz >= ppr && rz < ppr + G.pt);
        if (phys) {                                                                   // :576-582
          const int vi = __float_as_int(spos[s].w); float4 v = vel[vi];   // the velocity stays in arrival order (p3m_internal.h)
          v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;
          vel[vi] = v;
        }
        mag = sqrtf(ax * ax + ay * ay + az * az);                                     // :617
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mag = fmaxf(mag, __shfl_xor(mag, o, 64));
      if (lane == 0 && mag > 0.f) p3m_atomic_max_nonneg(tile_max + tile, mag);
    }
  }
}


// ------------------------------------------------------------------ extended PP, version 3: the partner region of a patch through LDS
// k_pp_ext2 at the reference's density is bound by cache-line traffic: every home record reads 50 cell offsets and ~15 partner
// records straight from global memory, 44 cache lines per load instruction -- ~200 KB of lines per 64 home records through an
// L2 that hits 44 % of the time.  Here a task is up to PP3_NT home records of a 3-D PATCH (PP3_HZ planes x PP3_HY rows x xbw
// cells, 7/8 of PP3_NT home records at the mean density) worked by PP3_NT / 64 wavefronts, and everything its homes can reach -- the
// (HZ+2r) x (HY+2r) partner rows clipped to the patch's x range +-r -- is brought into LDS ONCE per task by row-wise coalesced
// loads: the cell offsets of every partner row (one load instruction per row, kept as 16-bit offsets from the row's first
// record) and the partner records themselves (a flat copy over the concatenated row segments).  A home lane then finds its
// (2r+1)^2 row windows in the LDS offsets, lists the partners' LDS indices (lanes with few partners) or walks the windows
// directly (dense cells), and reads the partners from LDS: a partner record is fetched from HBM once per ~3 home records it
// serves instead of once per home record, the offsets once per patch.  Regions with more records than the staging area holds
// (blobs) are worked off in batches of PP3_PCAP records of the concatenated row segments; a row segment of more than 65 534
// records sends the task down a plain per-lane path over global memory.  The partner order per home record (rows in z, y
// order, ascending sorted index) is that of k_pp_ext.
#define PP3_HZ 8        // patch: 8 planes x 8 rows (4 x 16 measured 4.5 % slower: 160 partner rows per task against 144)
#define PP3_HY 8
#define PP3_NT 256
#define PP3_PCAP 704
#define PP3_LCAP 32      // list entries per lane
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
  if (j == 0) plan[g] = (count + PP3_NT - 1) / PP3_NT;
}
template <int PPR>   // PPR > 0: pp_range known at compile time (the reference's default 2); 0: any
__global__ __launch_bounds__(PP3_NT) __attribute__((amdgpu_waves_per_eu(PP3_WPE, PP3_WPE))) void k_pp_ext3(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, PPGeo G, PPForce F,
                                                    float a_mid, float dt, float *__restrict__ tile_max, const int *__restrict__ plan,
                                                    const int2 *__restrict__ task2, int ngroups, int npy, int npx, int xbw, int ntask_cap, int *__restrict__ counter,
                                                    int Wp, int NRmax, int fat_limit) {   // fat_limit: 65534 (see "fat" below); task2: {group, sub-task} of every task; Wp: entries per row of the offset table (xbw + 2r + 1 rounded up to even); NRmax: partner rows
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
  int *misc = roff + NH + 1;                        // [0] task, [1] fat flag, [2..5] wave maxima
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned char *mylist = list + (wv * 64 + lane) * PP3_LSTR;                 // this lane's entries
  const int ppr = G.ppr, e = G.pt + 2 * ppr, E = G.E, npz = (e + PP3_HZ - 1) / PP3_HZ;
  const int ntask = min(plan[ngroups], ntask_cap);
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
  auto advance = [&]() {   // thread 0 only
    b_live = false;
    if (a_live) {
      const int sbeg = min(a_seg * per, ntask), send = min(sbeg + per, ntask), t = sbeg + a_tf;
      if (t < send) { b_val = task2[t]; b_live = true; tried = 0; }
      else { tried++; seg = (seg + 1) % PP3_NSEG; }
    }
    a_live = tried < PP3_NSEG;
    if (a_live) { a_seg = seg; a_tf = atomicAdd(counter + 32 * seg, 1); }
  };
  if (tid == 0) { a_live = true; a_seg = seg; a_tf = atomicAdd(counter + 32 * seg, 1); advance(); }
  for (;;) {
    __syncthreads();                                  // the previous task's readers of the LDS tables (and of misc) are done
    if (tid == 0) { misc[0] = b_live ? 1 : (a_live ? 0 : -1); misc[1] = 0; misc[6] = b_val.x; misc[7] = b_val.y; advance(); }
    __syncthreads();
    const int state = misc[0];
    if (state < 0) break;                             // every segment has run dry
    if (state == 0) continue;                         // a dry segment: the next draw is on its way
    const int g = misc[6], sub = misc[7];
    const int xb = g % npx, gy = (g / npx) % npy, gz = (g / (npx * npy)) % npz, tile = g / (npx * npy * npz);
    const int tz = tile / (G.T * G.T), ty = (tile / G.T) % G.T, tx = tile % G.T;
    const int lox = tx * G.pt + G.nb - ppr, loy = ty * G.pt + G.nb - ppr, loz = tz * G.pt + G.nb - ppr;
    const int hx0 = lox + xb * xbw, hx1 = min(hx0 + xbw, lox + e);           // home cells [hx0, hx1)
    const int hz0 = loz + gz * PP3_HZ, hy0 = loy + gy * PP3_HY;
    // the partner region: rows [Z0, Z1] x [Y0, Y1], cells [X0, X1]
    const int Z0 = max(hz0 - ppr, loz), Z1 = min(hz0 + PP3_HZ - 1 + ppr, loz + e - 1);
    const int Y0 = max(hy0 - ppr, loy), Y1 = min(hy0 + PP3_HY - 1 + ppr, loy + e - 1);
    const int X0 = max(hx0 - ppr, lox), X1 = min(hx1 - 1 + ppr, lox + e - 1);
    const int NRY = Y1 - Y0 + 1, NR = (Z1 - Z0 + 1) * NRY, W = X1 - X0 + 2;
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
      int hinc = hcnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(hinc, o, 64); if (lane >= o) hinc += u; }
      const int total = __shfl(hinc, NH - 1, 64);
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
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (r < NR) cum[r] = carry + inc - cnt;
        carry += __shfl(inc, 63, 64);
      }
      if (lane == 0) cum[NR] = carry;
    }
    // this thread's home record
    const int h = sub * PP3_NT + tid;
    const bool valid = h < total;
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
    int z0 = max(cz - ppr, loz), z1 = min(cz + ppr, loz + e - 1);
    // the reference's half-shell sweep starts only from the planes k <= pt + pp_range (:496): see k_pp_ext
    if (cz - loz >= G.pt + ppr) z1 = min(z1, loz + G.pt + ppr - 1);
    const int y0 = max(cy - ppr, loy), y1 = min(cy + ppr, loy + e - 1);
    const int x0 = max(cx - ppr, lox), x1 = min(cx + ppr, lox + e - 1);
    const bool phys = valid && (cx >= lox + ppr && cx < lox + ppr + G.pt && cy - loy >= ppr && cy - loy < ppr + G.pt && cz - loz >= ppr && cz - loz < ppr + G.pt);
    const int vi = rec_index(p);                      // the velocity stays in arrival order (p3m_internal.h); fetched now, needed after the sums
    float4 vrec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (phys) vrec = vel[vi];
    float ax = 0.f, ay = 0.f, az = 0.f;
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
      for (int b0 = 0; b0 < max(Ptot, 1); b0 += PP3_PCAP) {   // batches of the concatenated partner sequence (one, unless a blob sits here)
        const int b1 = min(b0 + PP3_PCAP, Ptot);
        if (!single) {
          if (b0 > 0) __syncthreads();                // the previous batch's readers are done
          stage(b0, b1);
          __syncthreads();
        }
        // pass 1: list the LDS indices of this lane's partners inside the batch.  A window holds 0.6 partners on average
        int n = 0;
        bool ovf = false;
        // the lane's first window starts the lowest position it will list (the windows come in the order of the staged sequence)
        int base = 0;
        if (valid) { const int r = (z0 - Z0) * NRY + (y0 - Y0); base = max(cum[r] + (int)offs[r * Wp + (x0 - X0)] - b0, 0); }
        if (single) {
          // the whole region is staged and the reach is known: the (2r+1)^2 windows unrolled, a window's
          // first three records stored unconditionally at the list's end and the end advanced by a compare (no branches: 25
          // windows cost ~700 instructions per wavefront where the predicated version below took 2500); a window of more
          // than three records anywhere in the wavefront sends it through the tail loop
          const int rh = valid ? (cz - Z0) * NRY + (cy - Y0) : 0;
          const unsigned short *wa = offs + rh * Wp + (x0 - X0), *wb = offs + rh * Wp + (x1 + 1 - X0);
          const int *wc = cum + rh;
          bool yok[2 * PPR + 1];
#pragma unroll
          for (int dy = -PPR; dy <= PPR; dy++) yok[dy + PPR] = valid && cy + dy >= y0 && cy + dy <= y1;
          // three consecutive slots take a, a+1, a+2 whatever the count: the next window overwrites what was not a partner
          auto append = [&](int a, int cnt) {
            unsigned char *o = mylist + min(n, PP3_LCAP);   // a lane past the capacity is walked, not listed
            const int e = a - base;
            o[0] = (unsigned char)e; o[1] = (unsigned char)(e + 1); o[2] = (unsigned char)(e + 2);
            ovf = ovf || (cnt > 0 && e + cnt > 256);
            if (__any(cnt > 3))
              for (int k = 3; k < cnt; k++) mylist[min(n + k, PP3_LCAP + 2)] = (unsigned char)(e + k);
            n += cnt;
          };
#pragma unroll 1
          for (int dz = -PPR; dz <= PPR; dz++) {        // one plane of windows per trip: the tail loops are not replicated 25 times
            const bool zok = cz + dz >= z0 && cz + dz <= z1;
#pragma unroll
            for (int dy = -PPR; dy <= PPR; dy++) {
              const bool rv = zok && yok[dy + PPR];
              const int d = dz * NRY + dy;             // uniform
              const int cr = wc[d], a = cr + (int)wa[d * Wp], b = cr + (int)wb[d * Wp];
              if (dz == 0 && dy == 0) { append(a, rv ? own0 - a : 0); append(own1, rv ? b - own1 : 0); }   // the own cell splits the own row's window (:515-516)
              else append(a, rv ? b - a : 0);
            }
          }
        } else if (valid) {
          // any reach, any batch: window by window
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
                if (b > a && b - base > 256) ovf = true;
                for (int v = a; v < b; v++) { if (n < PP3_LCAP) mylist[n] = (unsigned char)(v - base); n++; }
              }
            }
        }
        const bool listed = !ovf && n <= PP3_LCAP;
        const int nl = listed ? n : 0;
        const int nmax = wave_max_i(nl);
        for (int k = 0; k < nmax; k += 4) {           // four partners in flight at a time, two per evaluation; no branches: a slot past the list's end reads record 0 and adds zero
          float4 o[4];
          const unsigned e4 = *reinterpret_cast<const unsigned *>(mylist + k);   // four entries
#pragma unroll
          for (int u = 0; u < 4; u++) { const int i = base + (int)((e4 >> (8 * u)) & 255u); o[u] = prec[k + u < nl ? i : 0]; }
          pp_ext_eval2(p, o[0], o[1], k < nl, k + 1 < nl, F, ax2, ay2, az2);
          pp_ext_eval2(p, o[2], o[3], k + 2 < nl, k + 3 < nl, F, ax2, ay2, az2);
        }
        // dense lanes and multi-batch regions: walk the windows, partners from the staged batch, two at a time
        if (valid && !listed) {
          for (int zz = z0; zz <= z1; zz++)
            for (int yy = y0; yy <= y1; yy++) {
              const int r = (zz - Z0) * NRY + (yy - Y0);
              const int va = max(cum[r] + (int)offs[r * Wp + (x0 - X0)], b0), vb = min(cum[r] + (int)offs[r * Wp + (x1 + 1 - X0)], b1);
              int v = va;
              while (v < vb) {
                if (v >= own0 && v < own1) { v = own1; continue; }            // own cell is excluded (:515-516)
                const int w = v + 1;
                const bool okB = w < vb && !(w >= own0 && w < own1);
                pp_ext_eval2(p, prec[v - b0], prec[(okB ? w : v) - b0], true, okB, F, ax2, ay2, az2);
                v += 2;
              }
            }
        }
      }
      ax += ax2.x + ax2.y; ay += ay2.x + ay2.y; az += az2.x + az2.y;
    }
    float mag = 0.f;
    if (valid) {
      if (phys) {                                                                   // :576-582
        float4 v = vrec;
        v.x = v.x + ax * a_mid * P3M_G_F * dt; v.y = v.y + ay * a_mid * P3M_G_F * dt; v.z = v.z + az * a_mid * P3M_G_F * dt;
        vel[vi] = v;
      }
      mag = sqrtf(ax * ax + ay * ay + az * az);                                     // :617
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mag = fmaxf(mag, __shfl_xor(mag, o, 64));
    if (lane == 0 && mag > 0.f) p3m_atomic_max_nonneg(tile_max + tile, mag);
  }
}

// the smallest float r2 with sqrtf(r2) > t (sqrtf is correctly rounded and monotone): "rmag > t" becomes "r2 >= this"
static float first_r2_with_root_above(float t) {
  float r2 = t * t;
  while (sqrtf(r2) > t) r2 = nextafterf(r2, 0.0f);
  while (!(sqrtf(r2) > t)) r2 = nextafterf(r2, INFINITY);
  return r2;
}

int pp_extended(p3m_ctx *c, float a_mid, float dt, float mass_p) {
  P3M_TRY(particles_full_cells(c));
  const Geometry &g = c->g;
  if (g.pp_range == 0) return P3M_OK;
  PPGeo G{g.T, g.nb, g.pt, g.E, g.Nn, g.ms, g.pp_range, c->p.rsoft, c->p.pp_bias, (float)g.ncut};
  const int e = g.pt + 2 * g.pp_range;
  static const bool v1 = getenv("P3M_PP_EXT_V1") && getenv("P3M_PP_EXT_V1")[0] == '1';   // A/B switch: the LDS-tiled kernel of round 1
  // The LDS-staged kernel (k_pp_ext3) is the default.  P3M_PP_EXT_V2=1 selects the gather kernel k_pp_ext2 (ms per launch on a 560 tile,
  // uniform / blobs of 205 / blobs of 13 000: 3.2 / 9.0 / 636 against 2.0 / 7.5 / 422 for k_pp_ext3)
  static const bool v2 = getenv("P3M_PP_EXT_V2") && getenv("P3M_PP_EXT_V2")[0] == '1';
  const bool v3 = !v2;
  if (!v1 && v3) {
    // patches of PP3_HZ x PP3_HY rows x xbw cells holding 7/8 of PP3_NT home records at the mean density (one task)
    static const int xbw_env = getenv("P3M_PP_XBW") ? atoi(getenv("P3M_PP_XBW")) : 0;
    const double rho_mean = (double)c->np_all / ((double)g.E * g.E * g.E);
    int xbw = xbw_env > 0 ? xbw_env : (int)std::lround((0.875 * PP3_NT) / std::max(1e-9, rho_mean * PP3_HZ * PP3_HY));
    xbw = std::max(4, std::min(std::min(xbw, e), 64 - 2 * g.pp_range - 1));      // one load instruction per partner row
    const int npx = (e + xbw - 1) / xbw, npy = (e + PP3_HY - 1) / PP3_HY, npz = (e + PP3_HZ - 1) / PP3_HZ;
    const int64_t ngroups64 = (int64_t)g.ntiles * npz * npy * npx;
    const int64_t mult1 = std::min<int64_t>(g.T, 2 + (2 * g.pp_range) / g.pt), mult = mult1 * mult1 * mult1;
    const int64_t ngroups_max = (int64_t)g.ntiles * npz * npy * ((e + 3) / 4);
    const int64_t ntask_cap64 = mult * (c->cap / PP3_NT + 1) + ngroups_max + 64;
    if (ngroups_max > 0x3fffffff || ntask_cap64 > 0x7fffffff) { p3m_set_error("extended PP: too many patches"); return P3M_EINVAL; }
    const int ngroups = (int)ngroups64, ntask_cap = (int)ntask_cap64;
    if (!c->pp_plan) HIP_TRY(hipMalloc(&c->pp_plan, sizeof(int) * ((size_t)ngroups_max + 8)));
    if (!c->pp_task_group) HIP_TRY(hipMalloc(&c->pp_task_group, sizeof(int2) * (size_t)ntask_cap64));   // {group, sub-task} per task
    if (!c->pp_counter) HIP_TRY(hipMalloc(&c->pp_counter, sizeof(int) * 32 * PP_NSEG));
    P3M_TRY(scan_reserve(c, ngroups_max + 8));
    HIP_TRY(hipMemsetAsync(c->pp_counter, 0, sizeof(int) * 32 * PP_NSEG, c->stream));
    hipLaunchKernelGGL(k_pp_plan3, dim3(cdiv(ngroups, 4)), dim3(256), 0, c->stream, (const int *)c->cell_end, G, npy, npx, xbw, ngroups, c->pp_plan);
    HIP_TRY(hipGetLastError());
    P3M_TRY(exclusive_scan_i32(c, c->pp_plan, ngroups));
    hipLaunchKernelGGL(k_pp_fill2, dim3(cdiv(ngroups, 256)), dim3(256), 0, c->stream, (const int *)c->pp_plan, ngroups, reinterpret_cast<int2 *>(c->pp_task_group), ntask_cap);
    HIP_TRY(hipGetLastError());
    PPForce F{mass_p, G.pp_bias, 1.0f / G.pp_bias, 1.0f / G.ncut, first_r2_with_root_above(G.rsoft), first_r2_with_root_above(G.ncut + sqrtf(3.0f))};
    const int Wp = (xbw + 2 * g.pp_range + 2) & ~1, NRmax = (PP3_HZ + 2 * g.pp_range) * (PP3_HY + 2 * g.pp_range);
    const size_t lds = sizeof(float4) * PP3_PCAP + sizeof(unsigned short) * (size_t)NRmax * Wp + (size_t)(PP3_NT / 64) * 64 * PP3_LSTR +
                       sizeof(int) * ((size_t)2 * NRmax + 1 + 2 * PP3_HZ * PP3_HY + 1 + 8);   // 8: misc
    // a partner row segment of more than 65 534 records does not fit the 16-bit offsets: its task takes the per-lane path over
    // global memory.  P3M_PP_FAT_LIMIT=n lowers the limit (a test switch: ordinary inputs then run that path)
    static const int fat_limit = getenv("P3M_PP_FAT_LIMIT") ? std::max(1, std::min(65534, atoi(getenv("P3M_PP_FAT_LIMIT")))) : 65534;
    auto kern = g.pp_range == 2 ? k_pp_ext3<2> : k_pp_ext3<0>;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const int wpc3 = getenv("P3M_PP_WPC") ? atoi(getenv("P3M_PP_WPC")) : 0;
    int wpc = wpc3 > 0 ? wpc3 : (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / lds));   // resident workgroups per CU by LDS
    hipLaunchKernelGGL(kern, dim3(256 * wpc), dim3(PP3_NT), lds, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, F, a_mid, dt,
                       c->d_tile_ext, (const int *)c->pp_plan, (const int2 *)c->pp_task_group, ngroups, npy, npx, xbw, ntask_cap, c->pp_counter, Wp, NRmax, fat_limit);
    HIP_TRY(hipGetLastError());
    return P3M_OK;
  }
  if (!v1) {
    // a group is a patch of PP_RG rows x xbw cells holding ~192 records (three tasks) at the mean density.  Measured on a 560
    // tile (ms per launch, uniform / 30 % of the particles in blobs of 205): whole rows 3.85 / 13.1, 96 cells 4.02 / 9.9,
    // 24 cells 4.83 / 10.0: compact patches keep a blob's members in one wavefront (the broadcast path then serves dozens
    // of lanes at once instead of a handful), whole rows keep the window loads of a uniform task on the fewest rows
    static const int xbw_env = getenv("P3M_PP_XBW") ? atoi(getenv("P3M_PP_XBW")) : 0;
    const double rho_mean = (double)c->np_all / ((double)g.E * g.E * g.E);
    int xbw = xbw_env > 0 ? xbw_env : (int)std::lround(192.0 / std::max(1e-9, rho_mean * PP_RG));
    xbw = std::max(4, std::min(xbw, e));
    const int nxb = (e + xbw - 1) / xbw;
    const int ngy = (e + PP_RG - 1) / PP_RG;
    const int64_t ngroups64 = (int64_t)g.ntiles * e * ngy * nxb;
    // a record is a home record of every tile whose extended region holds its cell: per axis at most 2 + 2*ppr/pt tiles
    const int64_t mult1 = std::min<int64_t>(g.T, 2 + (2 * g.pp_range) / g.pt), mult = mult1 * mult1 * mult1;
    const int64_t ntask_cap64 = mult * (c->cap / 64 + 1) + ngroups64 + 64;
    if (ngroups64 > 0x3fffffff || ntask_cap64 > 0x7fffffff) { p3m_set_error("extended PP: too many row groups"); return P3M_EINVAL; }
    const int ngroups = (int)ngroups64, ntask_cap = (int)ntask_cap64;
    const int64_t ngroups_max = (int64_t)g.ntiles * e * ngy * ((e + 3) / 4);
    if (ngroups_max > 0x3fffffff) { p3m_set_error("extended PP: too many row groups"); return P3M_EINVAL; }
    // each buffer under its own check: a failed allocation must not leave the others looking ready
    if (!c->pp_plan) HIP_TRY(hipMalloc(&c->pp_plan, sizeof(int) * ((size_t)ngroups_max + 8)));
    if (!c->pp_task_group) HIP_TRY(hipMalloc(&c->pp_task_group, sizeof(int) * (size_t)(mult * (c->cap / 64 + 1) + ngroups_max + 64)));   // any patch width
    if (!c->pp_counter) HIP_TRY(hipMalloc(&c->pp_counter, sizeof(int) * 32 * PP_NSEG));
    P3M_TRY(scan_reserve(c, ngroups_max + 8));
    HIP_TRY(hipMemsetAsync(c->pp_counter, 0, sizeof(int) * 32 * PP_NSEG, c->stream));
    hipLaunchKernelGGL(k_pp_plan, dim3(cdiv(ngroups, 256)), dim3(256), 0, c->stream, (const int *)c->cell_end, G, ngy, nxb, xbw, ngroups, c->pp_plan);
    HIP_TRY(hipGetLastError());
    P3M_TRY(exclusive_scan_i32(c, c->pp_plan, ngroups));
    hipLaunchKernelGGL(k_pp_fill, dim3(cdiv(ngroups, 256)), dim3(256), 0, c->stream, (const int *)c->pp_plan, ngroups, c->pp_task_group, ntask_cap);
    HIP_TRY(hipGetLastError());
    PPForce F{mass_p, G.pp_bias, 1.0f / G.pp_bias, 1.0f / G.ncut, first_r2_with_root_above(G.rsoft), first_r2_with_root_above(G.ncut + sqrtf(3.0f))};
    static const int wpc = getenv("P3M_PP_WPC") ? atoi(getenv("P3M_PP_WPC")) : 15;          // resident wavefronts per CU (10.4 KB of LDS each)
    static const bool unr = getenv("P3M_PP_UNROLL") && getenv("P3M_PP_UNROLL")[0] == '1';    // compile-time reach: all 25 windows loaded up front (116 VGPRs)
    if (g.pp_range == 2 && unr)
      hipLaunchKernelGGL(k_pp_ext2<2>, dim3(256 * wpc), dim3(64), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, F, a_mid, dt,
                         c->d_tile_ext, (const int *)c->pp_plan, (const int *)c->pp_task_group, ngroups, ngy, nxb, xbw, ntask_cap, c->pp_counter);
    else
      hipLaunchKernelGGL(k_pp_ext2<0>, dim3(256 * wpc), dim3(64), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, F, a_mid, dt,
                         c->d_tile_ext, (const int *)c->pp_plan, (const int *)c->pp_task_group, ngroups, ngy, nxb, xbw, ntask_cap, c->pp_counter);
    HIP_TRY(hipGetLastError());
    return P3M_OK;
  }
  if (g.pp_range <= 4) {
    // x extent of a block: about 200 busy lanes (home records x PP_LPH) at the mean density, at most 128 cells
    const double rho = (double)c->np_all / ((double)g.E * g.E * g.E);
    int nbx = (int)std::ceil((double)e * PB_Y * PB_Z * rho * PP_LPH / 200.0);
    nbx = std::max(nbx, (e + 127) / 128); nbx = std::min(nbx, std::max(1, e / 8));
    const int bx_cells = (e + nbx - 1) / nbx, nby = (e + PB_Y - 1) / PB_Y;
    nbx = (e + bx_cells - 1) / bx_cells;
    const int HR = (PB_Y + 2 * g.pp_range) * (PB_Z + 2 * g.pp_range), wseg = bx_cells + 2 * g.pp_range + 1;
    const size_t lds = sizeof(int) * (4 * HR + PB_Y * PB_Z + 1 + ((HR * wseg + 3) & ~3) + 4) + sizeof(float4) * PPT_CAP;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pp_ext_tiled), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned blocks = (unsigned)((int64_t)g.ntiles * nbx * nby * nby);
    hipLaunchKernelGGL(k_pp_ext_tiled, dim3(blocks), dim3(256), lds, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, mass_p, a_mid,
                       dt, c->d_tile_ext, bx_cells, nbx, nby, first_r2_with_root_above(G.rsoft), first_r2_with_root_above(G.ncut + sqrtf(3.0f)));
    HIP_TRY(hipGetLastError());
    return P3M_OK;
  }
  const unsigned blocks = (unsigned)((int64_t)g.ntiles * e * e);
  hipLaunchKernelGGL(k_pp_ext, dim3(blocks), dim3(64), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, mass_p, a_mid,
                     dt, c->d_tile_ext);
  HIP_TRY(hipGetLastError());
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
    P3M_TRY(pp_intra(c, a_mid, dt, mass_p)); P3M_TRY(pp_extended(c, a_mid, dt, mass_p));   // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; i++) P3M_TRY(pp_intra(c, a_mid, dt, mass_p));
    HIP_TRY(hipEventRecord(e1, c->stream));
    for (int i = 0; i < reps; i++) P3M_TRY(pp_extended(c, a_mid, dt, mass_p));
    HIP_TRY(hipEventRecord(e2, c->stream));
    HIP_TRY(hipEventSynchronize(e2));
    HIP_TRY(hipEventElapsedTime(&a, e0, e1)); HIP_TRY(hipEventElapsedTime(&b, e1, e2));
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
