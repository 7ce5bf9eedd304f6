// kick_fused.hip -- the last pass of the fine-mesh force and the NGP kick in ONE kernel (round 5).
//
// The NGP force box exists to be read once: k_fft_x_inv2 (fft.hip) writes the three components of every box row
// (particle_mesh_threaded.f90:197-204), k_fine_kick_rows (fine_mesh.hip) reads them back to form max |F|^2 (:208-223) and to
// give every record the force of its own cell (:244-270) -- 1.6 GB written and 1.6 GB read per rank and step at the
// bench's size.  An NGP record needs only its OWN box row, and the records of a cell row are one contiguous range of the
// sorted store.  k_fft_x_inv2_kick therefore transforms the three components of a batch of box rows in ONE trip (slot
// 3*row + comp of the x pass's row batch: at n = 560 a wavefront holds exactly the three components of one row), leaves
// the real rows in LDS (over the gather staging buffer, which is free by then) instead of in the box, forms the maximum
// from LDS and kicks the rows' records from LDS -- velocity gather, fine kick, coarse kick (coarse_velocity.f90:137-179) and the
// survivor count of delete_particles as in k_fine_kick_rows, same expressions, same order.  The box is not written.
//
// A record whose reference cell floor(xv + offset_tile) lies in a box row of ANOTHER batch than the row it is sorted into
// (rounding at a cell face, particle_mesh_threaded.f90:248-249) is left out here: k_ngp_fixup (fine_mesh.hip) has flagged that
// row during the deposit, the pass stores flagged rows to the box as k_fft_x_inv2 would, and k_kick_fix (fine_mesh.hip)
// kicks the handful of such records from there afterwards.
//
// Pipeline of one workgroup (persistent, grid-stride over the batches).  Trip i:
//   A  gathered LY elements of batch i (registers, requested in trip i-1) -> B
//      ; requests: the records of batch i (the chain range -> record -> velocity costs no exposed round trip: ranges two trips ahead)
//   B  requests: LY elements of batch i+1; stage 1 (dft<R1>) B -> X
//   C  requests: the velocities of batch i's records; stage 2 (dft<R2>) X -> F (= B's memory): real box rows, 1/n^3 applied
//   D  max |F|^2 from F; the records of batch i take their force from F; both kicks; velocities stored; the last wavefront
//      writes the tables of batch i+2 (row ranges requested in trip i-1) and requests the row ranges of batch i+3
// with a barrier after each step where the workgroup is four wavefronts (the rows in LDS are then read across wavefronts), a
// wavefront-wide fence where it is one (KCfg).  LDS: 18.5 KB per wavefront at n = 560, eight wavefronts per CU.  Measured with shader-clock stamps per step (-DKF_TRACE): with the coarse gathers inside
// step D that step was two exposed round trips long (the second slot of records belongs to the first wavefront alone).
#include "p3m_internal.h"
#include "fft_core.h"
#include "fft_x2.h"
#include "kick_fused.h"

// timing-only ablations (tools/kf_ab.sh; the results are wrong): KF_AB_NOKICK, KF_AB_NOMAX, KF_AB_NOCOARSE; KF_TRACE: clock stamps

#pragma clang fp contract(off)   // cell indices, CIC weights and the kicks: the reference's unfused fp32 expressions
__device__ __forceinline__ void kf_tile_xyz(int tile, int T, int &tx, int &ty, int &tz) {
  tx = ty = tz = 0;
  if (T > 1) { tz = tile / (T * T); const int r = tile - tz * T * T; ty = r / T; tx = r - ty * T; }
}
// coarse_velocity.f90:143: the coarse cell below a record (its first corner) and the distances to it
struct CoarseCell { unsigned o0; float dx1, dy1, dz1; };
__device__ __forceinline__ CoarseCell kf_coarse_cell(const float4 &p, const KickFuseArgs &a) {
  const float inv = 1.0f / (float)a.ms;
  const float cx_ = inv * p.x - 0.5f, cy_ = inv * p.y - 0.5f, cz_ = inv * p.z - 0.5f;
  const int ci = (int)floorf(cx_) + 1, cj = (int)floorf(cy_) + 1, ck = (int)floorf(cz_) + 1;
  const unsigned m = (unsigned)(a.ncn + 2);
  return CoarseCell{((unsigned)ck * m + (unsigned)cj) * m + (unsigned)ci, (float)ci - cx_, (float)cj - cy_, (float)ck - cz_};
}
// the 24 values the coarse kick of a record interpolates between: corner (cz,cy,cx), component c at cf[3*(4*cz + 2*cy + cx) + c]
__device__ __forceinline__ void kf_coarse_gather(float (&cf)[24], unsigned o0, const KickFuseArgs &a) {
  const unsigned m = (unsigned)(a.ncn + 2), ccs = m * m * m;   // 3 (ncn+2)^3 < 2^31 (fine_kick_fusable)
#pragma unroll
  for (int cz = 0; cz < 2; cz++)
#pragma unroll
    for (int cy = 0; cy < 2; cy++)
#pragma unroll
      for (int cx = 0; cx < 2; cx++) {
        const unsigned o = o0 + (cz ? m * m : 0u) + (cy ? m : 0u) + (cx ? 1u : 0u);
        const int e = 3 * (4 * cz + 2 * cy + cx);
        if (cx) continue;   // the two x corners are neighbours in memory: one 8-byte load (4-byte aligned: the hardware takes it)
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        const char *fb_ = reinterpret_cast<const char *>(a.fc);
        const f2u v0 = *reinterpret_cast<const f2u *>(fb_ + (size_t)(o << 2)), v1 = *reinterpret_cast<const f2u *>(fb_ + (size_t)((o + ccs) << 2)),
                  v2 = *reinterpret_cast<const f2u *>(fb_ + (size_t)((o + 2u * ccs) << 2));
        cf[e] = v0.x; cf[e + 3] = v0.y; cf[e + 1] = v1.x; cf[e + 4] = v1.y; cf[e + 2] = v2.x; cf[e + 5] = v2.y;
      }
}
// :265-266 and coarse_velocity.f90:153-168 -- the arithmetic of k_fine_kick_rows, term by term, in its order
template <bool COARSE>
__device__ __forceinline__ void kf_kick(float4 &v, float fx, float fy, float fz, const CoarseCell &cc, const float (&cf)[24], const KickFuseArgs &a) {
  v.x = v.x + fx * a.a_mid * P3M_G_F * a.dt;
  v.y = v.y + fy * a.a_mid * P3M_G_F * a.dt;
  v.z = v.z + fz * a.a_mid * P3M_G_F * a.dt;
#ifndef KF_AB_NOCOARSE
  if (COARSE) {
    const float dx1 = cc.dx1, dy1 = cc.dy1, dz1 = cc.dz1, dx2 = 1.0f - dx1, dy2 = 1.0f - dy1, dz2 = 1.0f - dz1;
#pragma unroll
    for (int cz = 0; cz < 2; cz++)
#pragma unroll
      for (int cy = 0; cy < 2; cy++)
#pragma unroll
        for (int cx = 0; cx < 2; cx++) {
          const float dV = a.a_mid * P3M_G_F * a.dt * (cx ? dx2 : dx1) * (cy ? dy2 : dy1) * (cz ? dz2 : dz1);
          const int e = 3 * (4 * cz + 2 * cy + cx);
          v.x = v.x + cf[e] * dV; v.y = v.y + cf[e + 1] * dV; v.z = v.z + cf[e + 2] * dV;
        }
  }
#endif
}
// max(m, |F|^2 of four box points): f, f + pitch, f + 2 pitch hold the three components (:217-218, unfused as in k_fine_kick_rows).  The
// maximum as plain v_max3 instructions: fmaxf carries a canonicalisation of each operand that nothing here needs (no NaN source)
__device__ __forceinline__ float kf_max2(float m, const float *f, int pitch) {
  const float4 x = *reinterpret_cast<const float4 *>(f), y = *reinterpret_cast<const float4 *>(f + pitch), z = *reinterpret_cast<const float4 *>(f + 2 * pitch);
  const float a = x.x * x.x + y.x * y.x + z.x * z.x, b = x.y * x.y + y.y * y.y + z.y * z.y, c = x.z * x.z + y.z * y.z + z.z * z.z, d = x.w * x.w + y.w * y.w + z.w * z.w;
  asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b));
  asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(c), "v"(d));
  return m;
}
// one box point (the columns past the last whole group of four)
__device__ __forceinline__ float kf_max1(float m, const float *f, int pitch) {
  const float x = f[0], y = f[pitch], z = f[2 * pitch];
  const float v = x * x + y * y + z * z;
  asm("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(v));
  return m;
}
#pragma clang fp contract(fast)  // the butterflies may fuse, as in fft.hip

#ifdef KF_TRACE   // diagnostic build (tools/variant.sh ... -DKF_TRACE): shader-clock stamps of the steps of one workgroup's trips
__device__ long long kf_trace_buf[8 * 64 * 8];
#define KF_STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 5 && blockIdx.x < 512 && trip < 64) kf_trace_buf[((blockIdx.x >> 6) * 64 + trip) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define KF_STAMP(k) do { } while (0)
#endif

// Workgroup shape.  A wavefront of the x pass holds RPW = 64 / max(R1, R2) whole rows.  Where that is exactly three (n = 304, 560, 608)
// a wavefront owns the three components of its box rows from the gather to the kick, nothing is shared between wavefronts, and the
// workgroup IS one wavefront: no s_barrier anywhere (the four steps of a trip are separated by wavefront-wide fences), and the eight
// wavefronts of a CU drift apart -- one gathers coarse forces while another runs its butterflies.  (As one workgroup of four wavefronts
// with four barriers per trip the pass was 55 % VALU-busy and scaled 1.6x from one workgroup per CU to two: latency, not throughput.)
// Other shapes keep 256 threads: the slots 3*row + comp then straddle wavefronts.
template <int TB> __device__ __forceinline__ void kf_sync() {
  if constexpr (TB == 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // LDS operations of a wavefront complete in order
  else __syncthreads();
}
template <int R1, int R2> struct KCfg {
  using X = X2Cfg<R1, R2>;
  static constexpr int h = X::h, Q = X::Q, RPW = X::RPW, R2P = X::R2P, P = X::P, NCH = X::NCH;
  static constexpr int TB = (RPW == 3) ? 64 : 256, RB = RPW * (TB / 64), NLD = (RB * NCH * 8 + TB - 1) / TB;   // TB: the threads that share a batch
  // threads of a workgroup; batches in flight per workgroup.  KF_WG4: the one-wavefront shape as workgroups of four independent wavefronts
  // sharing the twiddle tables (33 KB for four: 16 wavefronts per CU fit) -- measured 878 us against 857 at three wavefronts per SIMD,
  // and 920 capped at 128 VGPRs for four (64 bytes of scratch): the pass does not want more wavefronts, it wants fewer instructions
#ifdef KF_WG4
  static constexpr int LT = 256, NW = LT / TB;
#else
  static constexpr int LT = TB, NW = 1;
#endif
  // One wavefront per workgroup: the gather buffer B, the exchange buffer X and the real rows F are ONE buffer (each is dead when the next is
  // written: a wavefront-wide fence between the last read and the first write suffices) -- 11.8 KB per wavefront at n = 560 instead of
  // 18.5, i.e. room for three wavefronts per SIMD.  Four wavefronts: B | X apart (F over B), the steps separated by barriers.
  static constexpr bool WAVE = TB == 64;
  static constexpr size_t ube = WAVE ? ((size_t)RB * P > (size_t)RB * R1 * R2P ? (size_t)RB * P : (size_t)RB * R1 * R2P) : (size_t)RB * P + (size_t)RB * R1 * R2P;
  static constexpr size_t lds = sizeof(float2) * (NW * ube + h + (size_t)R2 * (R1 | 1));   // B (| X) of every batch in flight | tw | twl
};
template <int R1, int R2, bool COARSE>
__global__ __launch_bounds__((KCfg<R1, R2>::LT)) void k_fft_x_inv2_kick(KickFuseArgs a) {
  using C = KCfg<R1, R2>;
  constexpr int TB = C::TB;
  constexpr int h = C::h, Q = C::Q, RB = C::RB, R2P = C::R2P, P = C::P, NLD = C::NLD, NR = RB / 3, R1P = R1 | 1;
  extern __shared__ float2 lds[];
  const int wid = C::WAVE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // one wavefront per batch: the workgroup's wavefronts share the twiddle tables, nothing else
  c32 *B = reinterpret_cast<c32 *>(lds) + wid * C::ube, *X = C::WAVE ? B : B + RB * P, *tw = reinterpret_cast<c32 *>(lds) + C::NW * C::ube, *twl = tw + h;
  float *F = reinterpret_cast<float *>(B);            // [3*NR][FP] real box rows of the batch, over B
  // three sets of the per-batch tables: trip i reads the sets of batch i (steps C, D) and of batch i+1 (the requests of step B), and its
  // LAST wavefront writes the set of batch i+2 in step D -- where it would otherwise wait for the first one, which alone holds the
  // batch's second slot of records (in step A the tables were on the path every wavefront waits for: 2300 of a trip's 20 000 clocks)
  __shared__ int64_t src_row[3][RB], box_off[3][NR];
  __shared__ int rp0[3][NR], rp1[3][NR], rtile[3][NR], rflag[3][NR];
  for (int i = threadIdx.x; i < h; i += C::LT) tw[i] = reinterpret_cast<const c32 *>(a.tw_g)[i];
  const int n = a.n, fb = a.fb, fbp = a.fbp, lo = a.lo, FP = a.FP, nchunk = a.px / BXC;
  const int tid = C::WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, rw = lane / Q, q = lane - rw * Q;   // tid: thread of the batch
  const int r = (tid >> 6) * C::RPW + rw;              // slot of the row batch: box row r / 3, component r % 3
  const bool act = rw < C::RPW && r < 3 * NR, s1 = act && q < R2, s2 = act && q < R1;
  const int rbr = r / 3, rcomp = r - 3 * rbr;
  const int nbatch = (a.rows_total + NR - 1) / NR;
  const float rscale = 1.0f / a.inv_scale;
  const int64_t cstride = (int64_t)n * BXC;
  const float fNn = (float)a.Nn;
  const int nct = a.pt / a.ms;
  const int FR = C::WAVE ? (lo + 3) & ~3 : 0;   // front pad of a real row in LDS (one wavefront per workgroup: see step C)
  // W_h^{q*k1} at twl[q*R1P + k1] (k_fft_x_inv2 keeps the R1 factors of its q in registers; here they would push the pass to one
  // wavefront per SIMD: R1P is odd, the R2 lanes of a row read distinct banks, the rows of a wavefront the same words)
  for (int i = threadIdx.x; i < R1 * R2; i += C::LT) { const int qq = i / R1, k1 = i - qq * R1; twl[qq * R1P + k1] = reinterpret_cast<const c32 *>(a.tw_g)[2 * qq * k1]; }
  // box row -> (tile, plane kk, row jj); rows_total * fb < 2^32 (kick_fused_impl): the divisions are one mulhi each
  const fdiv_t d_fb{a.m_fb, fb};
  auto decode = [&](int brow, int &tile, int &kk, int &jj) {
    const int t2 = fdiv(brow, d_fb); tile = fdiv(t2, d_fb);
    jj = brow - t2 * fb; kk = t2 - tile * fb;
  };
  // Per-batch tables.  One wavefront per workgroup (NR = 1): the batch IS one box row, the same for every lane -- its geometry is
  // wavefront-uniform arithmetic on the work item (scalar registers), its record range and flag are requested a trip ahead by every
  // lane from the same address (rgn*), and no table lives in LDS.  Four wavefronts: thread 3*br of the last wavefront requests the
  // range and the flag of box row br of a batch (rg*) and writes them, with the rows' addresses, into the batch's LDS set a trip later.
  const int tt = tid - (C::TB - 64), tbr = tt / 3, tcomp = tt - 3 * tbr;
  int rg0 = 0, rg1 = 0, rgf = 0, rgn0 = 0, rgn1 = 0, rgnf = 0;
  const int64_t comp_src = (int64_t)a.ntile * n * nchunk * n * BXC;        // LY elements between two components of a row
  auto row_src = [&](int brow) {   // LY offset of component 0 of box row brow
    int tile, kk, jj; decode(brow, tile, kk, jj);
    return ((((int64_t)tile * n + (kk + lo)) * nchunk) * n + (jj + lo)) * BXC;
  };
  auto rangeload = [&](int w) {
    if constexpr (C::WAVE) {
      rgn0 = 0; rgn1 = 0; rgnf = 0;
      if (w < nbatch) {
        int tile, kk, jj; decode(w, tile, kk, jj);
        int tx, ty, tz; kf_tile_xyz(tile, a.T, tx, ty, tz);
        const int64_t erow = (int64_t)(tz * a.pt + kk + lo) * a.E + (ty * a.pt + jj + lo);
        if (a.crow) { const int *t = a.crow + erow * a.crow_w + a.ncn + 2 + 2 * tx; rgn0 = t[0]; rgn1 = t[1]; }
        else { const int *t = a.cs + erow * a.E + tx * a.pt + lo; rgn0 = t[0]; rgn1 = t[fb]; }
        rgnf = a.rowflag[w];
      }
      return;
    }
    rg0 = 0; rg1 = 0; rgf = 0;
    const int brow = w * NR + tbr;
    if (tt >= 0 && tt < 3 * NR && tcomp == 0 && w < nbatch && brow < a.rows_total) {
      int tile, kk, jj; decode(brow, tile, kk, jj);
      int tx, ty, tz; kf_tile_xyz(tile, a.T, tx, ty, tz);
      const int64_t erow = (int64_t)(tz * a.pt + kk + lo) * a.E + (ty * a.pt + jj + lo);
      if (a.crow) { const int *t = a.crow + erow * a.crow_w + a.ncn + 2 + 2 * tx; rg0 = t[0]; rg1 = t[1]; }
      else { const int *t = a.cs + erow * a.E + tx * a.pt + lo; rg0 = t[0]; rg1 = t[fb]; }
      rgf = a.rowflag[brow];
    }
  };
  auto tables = [&](int w, int set) {
    if constexpr (C::WAVE) return;
    if (tt >= 0 && tt < 3 * NR) {
      const int brow = w * NR + tbr;
      if (brow < a.rows_total) {
        src_row[set][tt] = row_src(brow) + tcomp * comp_src;
        int tile, kk, jj; decode(brow, tile, kk, jj);
        if (tcomp == 0) { box_off[set][tbr] = (int64_t)brow * fbp; rp0[set][tbr] = rg0; rp1[set][tbr] = rg1; rtile[set][tbr] = tile; rflag[set][tbr] = rgf; }
      } else if (tcomp == 0) { rp0[set][tbr] = 0; rp1[set][tbr] = 0; rtile[set][tbr] = 0; rflag[set][tbr] = 0; }
    }
  };
  // gather item e = tid + u*TB: l4 = e & 7, slot = (e >> 3) % RB, chunk = (e >> 3) / RB
  int grc[NLD];   // slot | chunk << 8 | l4 << 16, or -1
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int e = tid + u * C::TB, t = e >> 3, ch = t / RB;
    grc[u] = ch < C::NCH ? ((t - ch * RB) | (ch << 8) | ((e & 7) << 16)) : -1;
  }
  // One wavefront per workgroup: item u of a lane is slot ((tid >> 3) + 8u) % 3 and chunk ((tid >> 3) + 8u) / 3 -- the slot repeats every
  // three items, eight chunks on: three per-lane offsets (gcl: source, lcl: LDS) serve all items, the rest is a uniform stride, and only
  // the LAST item holds chunks past the row (its predicates are the only ones: per-item lane masks of a loop-invariant condition are
  // kept in scalar register pairs, and 40 of them were spilled to vector lanes and read back every trip)
  int64_t gcl[3] = {0, 0, 0}; int lcl[3] = {0, 0, 0};
  if constexpr (C::WAVE) {
#pragma unroll
    for (int cl = 0; cl < 3; cl++) {
      const int t0 = (tid >> 3) + 8 * cl, ch0 = t0 / 3, rr0 = t0 - 3 * ch0;
      gcl[cl] = rr0 * comp_src + ch0 * cstride + 2 * (tid & 7);
      lcl[cl] = rr0 * P + ch0 * BXC + 2 * (tid & 7);
    }
  }
  auto item_full = [](int u) { return ((7 + 8 * (u % 3)) / 3 + 8 * (u / 3)) * BXC + 15 <= h; };   // every lane's two elements are columns <= h
  auto item_chunk = [&](int u) { return ((tid >> 3) + 8 * (u % 3)) / 3 + 8 * (u / 3); };
  float4 g4[NLD];
  auto fetch = [&](int w, int set) {
    if constexpr (C::WAVE) {   // (w < nbatch)
      const c32 *sb = reinterpret_cast<const c32 *>(a.src) + row_src(w);
#pragma unroll
      for (int u = 0; u < NLD; u++) {
        const float4 *ps = reinterpret_cast<const float4 *>(sb + gcl[u % 3] + (u / 3) * 8 * cstride);
        if (item_full(u)) g4[u] = *ps;
        else { g4[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (item_chunk(u) < C::NCH) g4[u] = *ps; }
      }
      return;
    }
    const int nsl = 3 * min(NR, a.rows_total - w * NR);
    const int64_t wsrc = 0;
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      g4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int rr = grc[u] & 255, ch = (grc[u] >> 8) & 255, l4 = grc[u] >> 16;
      if (grc[u] >= 0 && rr < nsl) {
        const int64_t sr = C::WAVE ? wsrc + rr * comp_src : src_row[set][rr];   // (one wavefront: slot rr = component rr of the one row)
        g4[u] = reinterpret_cast<const float4 *>(a.src + sr + ch * cstride)[l4];
      }
    }
  };
  // the records of a batch, flattened over the threads: record f of the batch = record f - (records of the rows before) of its row.
  // The rows' ranges are the same for every lane: they are held in scalar registers (cum[br] = records of the rows before br)
  int rb0[NR], cum[NR + 1];
  auto ranges = [&](int set) {
    cum[0] = 0;
#pragma unroll
    for (int br = 0; br < NR; br++) {
      rb0[br] = __builtin_amdgcn_readfirstlane(C::WAVE ? rg0 : rp0[set][br]);
      cum[br + 1] = cum[br] + __builtin_amdgcn_readfirstlane(C::WAVE ? rg1 : rp1[set][br]) - rb0[br];
    }
  };
  auto locate = [&](int f, int &row, int &idx) {   // after ranges(set)
    row = 0; int d = rb0[0];
#pragma unroll
    for (int br = 1; br < NR; br++) if (f >= cum[br]) { row = br; d = rb0[br] - cum[br]; }
    idx = f + d;
    return f < cum[NR];
  };
  // The batch's records, flattened over the threads (one per thread; a batch of NR rows holds ~TB of them at the reference's density: the
  // rest, and the whole of a crowded batch, take the unprefetched loop in step D).  Requested in step A, their velocities -- reached
  // through the arrival index in the record -- in step C, both used in step D: the chain costs no exposed round trip.
  float4 pf = make_float4(-1.f, -1.f, -1.f, 0.f), vf = make_float4(0.f, 0.f, 0.f, 0.f); int ps = -1, pr = 0;   // record, velocity, sorted index (-1: none), row of the batch
  auto records = [&](int set) {
    ranges(set);
    pf = make_float4(-1.f, -1.f, -1.f, 0.f); ps = -1; pr = 0;
    int row, idx;
    if (locate(tid, row, idx)) { pf = a.spos[idx]; ps = idx; pr = row; }
  };
  auto physical = [&](const float4 &p) { return p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn; };   // chains of hoc(1..ncn) only (:234-236)
  float fmax2 = 0.f;
  // one record: physical, owned by the tile of its row, reference cell in a row of this batch -> force from F, both kicks, count.
  // cc / cf: its coarse cell and the 24 coarse-force values around it (requested in step C; have_c false: gathered here)
  auto kick_one = [&](bool valid, const float4 &p, float4 v, bool have_v, CoarseCell cc, float (&cf)[24], bool have_c, int s, int row, int set, int row0, int nbr) {
    if (__ballot(valid) == 0) return;   // nothing in this wavefront (the last slot of a batch belongs to its first wavefront alone)
    bool go = valid && physical(p);
    int tx = 0, ty = 0, tz = 0, tile = 0;
    if (a.T > 1) {
      // a record sits in the range of every tile whose box row covers it: only the row of its owner tile kicks it -- the tile of
      // the coarse cell of the chain the particle sits in (link_list.f90:19-21)
      if (C::WAVE) { int kk_, jj_; decode(row0, tile, kk_, jj_); } else tile = rtile[set][row];
      kf_tile_xyz(tile, a.T, tx, ty, tz);
      go = go && ((int)floorf(p.x / (float)a.ms)) / nct == tx && ((int)floorf(p.y / (float)a.ms)) / nct == ty && ((int)floorf(p.z / (float)a.ms)) / nct == tz;
    }
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (go) {
      const float offx = (float)a.nb - (float)(tx * a.pt), offy = (float)a.nb - (float)(ty * a.pt), offz = (float)a.nb - (float)(tz * a.pt);  // :227
      const float x = p.x + offx, y = p.y + offy, z = p.z + offz;                                       // :248
      const int i1 = (int)floorf(x) - lo, j1 = (int)floorf(y) - lo, k1 = (int)floorf(z) - lo;         // index into the force box
      const int lr = (tile * fb + k1) * fb + j1 - row0;
      go = (unsigned)lr < (unsigned)nbr && (unsigned)i1 < (unsigned)fb;                                // else: k_kick_fix (fine_mesh.hip)
      if (go) { const float *f = F + (3 * lr) * FP + FR + i1; fx = f[0]; fy = f[FP]; fz = f[2 * FP]; }
    }
    if (a.cnt256) {
      // every physical record passes here exactly once: the survivors of delete_particles, counted per block of 256 sorted records
      const int blk = go ? (s >> 8) : -1;
      unsigned long long rem = __ballot(go);
      while (rem) {
        const int first = __builtin_amdgcn_readlane(blk, __ffsll((long long)rem) - 1)   /* (a uniform lane: v_readlane, not the LDS round trip of __shfl) */;
        const unsigned long long m1 = __ballot(blk == first);
        if (lane == __ffsll((long long)m1) - 1) atomicAdd(&a.cnt256[first], __popcll(m1));
        rem &= ~m1;
      }
    }
    if (go) {
#ifdef KF_VEL_BY_S   // timing-only experiment: the velocity at the SORTED index (what a sorted velocity array would cost)
      const int vi = s;
#else
      const int vi = rec_index(p);   // the velocity stays in arrival order (p3m_internal.h)
#endif
      if (!have_v) v = a.vel[vi];
      if (COARSE && !have_c) { cc = kf_coarse_cell(p, a); kf_coarse_gather(cf, cc.o0, a); }
      kf_kick<COARSE>(v, fx, fy, fz, cc, cf, a);
      if (!a.dry) a.vel[vi] = v;
    }
  };

  const int G = (int)gridDim.x * C::NW;
  int w = (int)blockIdx.x * C::NW + wid, set = 0;
  if (C::WAVE) rangeload(w);   // (consumed at the top of the first trip)
  else if (w < nbatch) {
    rangeload(w); tables(w, 0);
    rangeload(w + G); if (w + G < nbatch) tables(w + G, 1);
    rangeload(w + 2 * G);
  }
  __syncthreads();   // the twiddle tables (the only barrier of the one-wavefront shape: its wavefronts part here)
  if (w < nbatch) fetch(w, 0);
  int trip = 0; (void)trip;
  for (; w < nbatch; w += G, set = set == 2 ? 0 : set + 1, trip++) {
    KF_STAMP(0);
    const int nxt = set == 2 ? 0 : set + 1, nn = nxt == 2 ? 0 : nxt + 1, wn = w + G, row0 = w * NR, nbr = min(NR, a.rows_total - row0);
    const bool rowok = r < 3 * nbr;
    // ---- A
    if constexpr (C::WAVE) {
#pragma unroll
      for (int u = 0; u < NLD; u++) {
        c32 *pb = B + lcl[u % 3] + (u / 3) * 8 * BXC;
        if (item_full(u)) { pb[0] = (c32){g4[u].x, g4[u].y}; pb[1] = (c32){g4[u].z, g4[u].w}; }
        else {
          const int ch = item_chunk(u), k = ch * BXC + 2 * (tid & 7);
          if (ch < C::NCH && k <= h) pb[0] = (c32){g4[u].x, g4[u].y};
          if (ch < C::NCH && k + 1 <= h) pb[1] = (c32){g4[u].z, g4[u].w};
        }
      }
    } else
#pragma unroll
    for (int u = 0; u < NLD; u++)
      if (grc[u] >= 0) {
        const int rr = grc[u] & 255, k = ((grc[u] >> 8) & 255) * BXC + 2 * (grc[u] >> 16);
        c32 *pb = B + rr * P + k;
        if (k <= h) pb[0] = (c32){g4[u].x, g4[u].y};
        if (k + 1 <= h) pb[1] = (c32){g4[u].z, g4[u].w};
      }
    if (C::WAVE) { rg0 = rgn0; rg1 = rgn1; rgf = rgnf; rangeload(wn); }   // this row's range (requested a trip ago); the next row's
    records(set);
    KF_STAMP(1);
    kf_sync<TB>();
    KF_STAMP(2);
    // ---- B
    if (wn < nbatch) fetch(wn, nxt);
    if (s1 && rowok) {
      const c32 *pk = B + r * P + q, *pm = B + r * P + (h - R2 * (R1 - 1)) - q, *pt = tw + q;
      c32 v[R1];
#pragma unroll
      for (int i = 0; i < R1; i++) {
        v[i] = c2r_pre(pk[R2 * i], pm[R2 * (R1 - 1 - i)], pt[R2 * i]);   // X[m], X[h-m], exp(-2 pi i m / n), m = R2*i + q: conj(e + i o), the forward machinery then yields conj(IFFT)
      }
      dft<R1>(v);
      if (C::WAVE) kf_sync<TB>();   // X is B's memory: every lane has read its elements of B
      c32 *pxw = X + (r * R1) * R2P + q;
#pragma unroll
      for (int k1 = 0; k1 < R1; k1++) pxw[k1 * R2P] = k1 ? vmulr(v[k1], twl[q * R1P + k1]) : v[0];
    } else if (C::WAVE) kf_sync<TB>();
    KF_STAMP(3);
    kf_sync<TB>();
    KF_STAMP(4);
    // ---- C
    vf = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef KF_VEL_BY_S
    if (ps >= 0) vf = a.vel[ps];
#else
    if (ps >= 0) vf = a.vel[rec_index(pf)];
#endif
    if (s2 && rowok) {
      c32 u[R2];
      const c32 *pxr = X + (r * R1 + q) * R2P;
#pragma unroll
      for (int b = 0; b < R2; b++) u[b] = pxr[b];
      dft<R2>(u);
      if (C::WAVE) kf_sync<TB>();   // F is X's memory: every lane has read its elements of X
      const int x0 = 2 * q - lo;   // box column of real element 2j for k2 = 0; lo is even
      float *pd = F + r * FP + FR + x0;
      if constexpr (C::WAVE) {
        // the WHOLE real row goes to LDS, box column i at F[FR + i] behind a front pad of FR >= lo floats: no lane of no store is masked
        // (14 x 3 lane masks of loop-invariant column tests otherwise); a flagged row is copied to the box from there in step D
#pragma unroll
        for (int k2 = 0; k2 < R2; k2++) *reinterpret_cast<float2 *>(pd + 2 * R1 * k2) = make_float2(u[k2].x * rscale, -u[k2].y * rscale);
      } else {
      const bool tobox = rflag[set][rbr] != 0;
      float *pg = a.box + rcomp * a.bcs + box_off[set][rbr] + x0;
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) {
        const int x = x0 + 2 * R1 * k2;
        if ((unsigned)x < (unsigned)fbp) {
          float2 o2 = make_float2(u[k2].x * rscale, -u[k2].y * rscale);
          if (x >= fb) o2.x = 0.f;
          if (x + 1 >= fb) o2.y = 0.f;
          *reinterpret_cast<float2 *>(pd + 2 * R1 * k2) = o2;
          if (tobox) *reinterpret_cast<float2 *>(pg + 2 * R1 * k2) = o2;
        }
      }
      }
    } else if (C::WAVE) kf_sync<TB>();
    KF_STAMP(5);
    kf_sync<TB>();
    KF_STAMP(6);
    // ---- D
    if (!C::WAVE) { if (w + 2 * G < nbatch) tables(w + 2 * G, nn); rangeload(w + 3 * G); }   // (the last wavefront: see the tables)
    {
      if constexpr (C::WAVE) {
        if (rgf) {   // a flagged row (k_ngp_fixup): k_kick_fix takes its forces from the box; pad columns are zero there
          for (int cmp = 0; cmp < 3; cmp++)
            for (int x = lane; x < fbp; x += 64) a.box[cmp * a.bcs + (int64_t)row0 * fbp + x] = x < fb ? F[cmp * FP + FR + x] : 0.f;
        }
      }
#ifndef KF_AB_NOMAX
      if constexpr (C::WAVE) {   // :217-218 over the columns [0, fb) of the one row
        for (int c4 = lane; c4 < (fb >> 2); c4 += 64) fmax2 = kf_max2(fmax2, F + FR + 4 * c4, FP);
        if (lane < (fb & 3)) fmax2 = kf_max1(fmax2, F + FR + (fb & ~3) + lane, FP);
      } else
      // :217-218 (pad columns are zero): wavefront w takes the rows w, w + 4, ... of the batch
      for (int br = tid >> 6; br < nbr; br += C::TB / 64)
        for (int c4 = lane; c4 < (fbp >> 2); c4 += 64) fmax2 = kf_max2(fmax2, F + (3 * br) * FP + 4 * c4, FP);
#endif
      ranges(set);
      const int total = cum[NR];
#ifndef KF_AB_NOKICK
      CoarseCell cc{0u, 0.f, 0.f, 0.f}; float cf[24];
      kick_one(ps >= 0, pf, vf, true, cc, cf, false, ps, pr, set, row0, nbr);
      for (int f0 = TB; f0 < total; f0 += TB) {   // a crowded batch (a blob): the records past the prefetched ones
        int row, idx;
        const bool have = locate(f0 + tid, row, idx);
        float4 p = make_float4(-1.f, -1.f, -1.f, 0.f);
        if (have) p = a.spos[idx];
        kick_one(have, p, p, false, cc, cf, false, idx, row, set, row0, nbr);
      }
#endif
    }
    KF_STAMP(7);
    kf_sync<TB>();
  }
  for (int o = 32; o > 0; o >>= 1) fmax2 = fmaxf(fmax2, __shfl_down(fmax2, o, 64));
  if (lane == 0 && fmax2 > 0.f) p3m_atomic_max_nonneg(a.fmax_out + p3m_slot() * 16, fmax2);
}

// ------------------------------------------------------------------ host side
// pitch of the rows in LDS: = 8 (mod 32) floats where that fits -- the (up to) RPW row groups of a wavefront's 8-byte stores then start
// 8 banks apart; 0: the box rows of a batch do not fit the staging buffer they are laid over
template <int R1, int R2> static int fused_row_pitch(int fbp, int lo) {
  using C = KCfg<R1, R2>;
  constexpr int NR = C::RB / 3;
  if (NR < 1) return 0;
  // one wavefront per workgroup: the whole real row behind its front pad (step C), over the larger of B and X
  const int need = C::WAVE ? (((lo + 3) & ~3) + 2 * C::h - lo + 3) & ~3 : fbp;
  const int limit = C::WAVE ? (int)(2 * C::ube / 3) & ~3 : (int)(((size_t)C::RB * C::P * 2) / (3 * NR)) & ~3;
  int fp = need + ((8 - need % 32 + 32) % 32);
  if (fp > limit) fp = need;
  return fp <= limit ? fp : 0;
}
template <int R1, int R2, bool COARSE> static int kick_fused_impl(p3m_ctx *c, KickFuseArgs &a) {
  using C = KCfg<R1, R2>;
  constexpr int NR = C::RB / 3;
  a.FP = fused_row_pitch<R1, R2>(a.fbp, a.lo);
  if (a.FP == 0) { p3m_set_error("fused kick: the box rows do not fit the staging buffer"); return P3M_EINVAL; }
  if ((int64_t)a.rows_total * a.fb >= 0xffffffffLL) { p3m_set_error("fused kick: too many box rows"); return P3M_EINVAL; }
  a.m_fb = fdiv_magic(a.fb);
  auto kern = k_fft_x_inv2_kick<R1, R2, COARSE>;
#ifdef KF_OCC1   // diagnostic build: half the workgroups per CU (how much of the pass is hidden by the others?)
  constexpr size_t lds = 2 * C::lds;
#else
  constexpr size_t lds = C::lds;
#endif
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  static int occ = 0;
  if (occ == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(kern), C::LT, lds));
    if (occ < 1) occ = 1;
  }
  const int64_t nwg = cdiv(cdiv(a.rows_total, NR), C::NW), g = (int64_t)256 * occ;   // a workgroup works NW batches at a time
  hipLaunchKernelGGL(kern, dim3((unsigned)(g < nwg ? g : nwg)), dim3(C::LT), lds, c->stream, a);
  HIP_TRY(hipGetLastError());
#ifdef KF_TRACE
  { static int calls = 0;
    if (++calls == 20) {
      static long long h[8 * 64 * 8];
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(kf_trace_buf), sizeof(h)));
      for (int b = 0; b < 8; b++) {
        double s[8] = {0}; int nt = 0;
        for (int t = 2; t < 60; t++) { const long long *p = h + (b * 64 + t) * 8; if (p[0] == 0 || p[8] == 0) continue; nt++;
          for (int k2 = 0; k2 < 7; k2++) s[k2] += (double)(p[k2 + 1] - p[k2]); s[7] += (double)(p[8] - p[7]); }
        if (nt) fprintf(stderr, "KF_TRACE wg %d (%d trips) clocks: A %.0f bar %.0f B %.0f bar %.0f C %.0f bar %.0f D %.0f bar %.0f\n", b * 64 + 5, nt, s[0] / nt, s[1] / nt, s[2] / nt, s[3] / nt, s[4] / nt, s[5] / nt, s[6] / nt, s[7] / nt);
      }
    } }
#endif
  return P3M_OK;
}

// box rows per batch of the fused pass for this line length and box row pitch (0: no fused pass): what k_ngp_fixup, k_fft_x_inv2_kick and
// k_kick_fix must agree on
int kick_fused_rows(int n, int fbp, int lo) {
#define X(H, A, B) if (n == 2 * H) return fused_row_pitch<A, B>(fbp, lo) ? KCfg<A, B>::RB / 3 : 0;
  P3M_X2_SIZES(X)
#undef X
  return 0;
}

int kick_fused_launch(p3m_ctx *c, KickFuseArgs &a, bool coarse) {
#define X(H, A, B) if (a.n == 2 * H) return coarse ? kick_fused_impl<A, B, true>(c, a) : kick_fused_impl<A, B, false>(c, a);
  P3M_X2_SIZES(X)
#undef X
  p3m_set_error("fused kick: n=%d has no two-register-stage x pass", a.n); return P3M_EINVAL;
}
