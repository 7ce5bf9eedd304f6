// fine_mesh.hip -- per-tile fine particle-mesh force (particle_mesh_threaded.f90:85-319), all tiles
// of a batch in one launch.
//
// Particles are sorted by extended fine cell (particles.hip); cs[c]..cs[c+1] is the record range of
// cell c, and a whole x-row of cells is one contiguous record range.  Deposits are therefore
// GATHERS: each output row of a tile is produced by one workgroup from the records of the (up to
// four) cell rows that can touch it -- no global atomics, no memset, coalesced row stores.
//
// Cell assignment follows the reference's fp32 expressions exactly: the fine deposit and kick use
// floor(xv + offset) with the TILE-dependent real offset = nf_buf - tile*nf_physical_tile_dim
// (:134,:227,:248-249), which can round a coordinate within half an ulp below an integer up into
// the next cell.  Since the parity metric (1e-5 relative rms over all particles) does not tolerate
// even one mis-binned particle, every kernel here recomputes the reference expression per record.
#include "p3m_internal.h"
#include <algorithm>
#include <stdlib.h>

int fft_inverse3_box(p3m_ctx *c, const FftPlan &pl, const float *rho_hat, float *work, const float *kern3, int batch, float *box, int fb, int lo,
                     int64_t bcs, bool zfwd);
int fft3d_forward_xy(p3m_ctx *c, const FftPlan &pl, float *data, float *scratch, int batch, bool u8, float mass_p);
int fft_inverse3_box_z(p3m_ctx *c, const FftPlan &pl, const float *rho_hat, float *work, const float *kern3, int batch, int fb, int lo, bool zfwd);
int fft_inverse3_box_y(p3m_ctx *c, const FftPlan &pl, float *work, int batch, int fb, int lo);
bool fft_x2_box_pass(int n, int lo);   // fft.hip: the force-box inverse x pass of this size is the two-register-stage kernel
#include "kick_fused.h"
#include "fft_x2.h"   // BXC: the bundle width of the LY / LZ layouts
#include "fft_core.h" // fdiv: division by a run-time constant as one mulhi

struct TileGeo { int T, nf, nb, pt, E, fb, rp, fbp; };  // rp: real row pitch of a fine array (2*px); fbp: force box row pitch

__device__ __forceinline__ void tile_xyz(int tile, int T, int &tx, int &ty, int &tz) {  // :86-90
  tz = tile / (T * T); const int r = tile - tz * T * T; ty = r / T; tx = r - ty * T;
}
__device__ __forceinline__ float block_sum_f(float v, float *sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) for (int i = 0; i < nw; i++) s += sh[i];
  __syncthreads();
  return s;
}

// ------------------------------------------------------------------ fine deposit, NGP (:119-151) and CIC (:153-160)
// One 64-lane workgroup per output row (tile,k,j).  NGP: a record lands in this row iff its
// tile-local cell (jt,kt) == (j,k); such records live in cell rows {j-1,j}x{k-1,k} (round-up).
// Window of chains deposited: NGP x_loc in [4,nf-4), CIC x_loc in [0,nf) (cic_l/cic_h, :119-124),
// membership by coarse cell = exact test on x because the boundaries are multiples of mesh_scale.
template <bool NGP>
__global__ __launch_bounds__(64) void k_fine_deposit(const float4 *__restrict__ spos, const int *__restrict__ cs, float *__restrict__ rho,
                                                     int tile0, TileGeo G, float mass_p, double *__restrict__ sum_interior, int nrows) {
  extern __shared__ __align__(16) float row[];  // rp floats
  const int nf = G.nf, E = G.E, pt = G.pt, nb = G.nb;
  // workgroups go to the eight XCDs in turn: XCD x works the rows x * ceil(nrows / 8) ... of the launch, a slab in z, so that the
  // sorted cell rows a CIC output row gathers from (each feeds four output rows) are fetched into ONE XCD's L2
  const int per8 = (nrows + 7) >> 3, brow = (int)(blockIdx.x & 7u) * per8 + (int)(blockIdx.x >> 3);
  if (brow >= nrows) return;
  const int j = brow % nf, k = (brow / nf) % nf, tl = brow / (nf * nf);
  int tx, ty, tz; tile_xyz(tile0 + tl, G.T, tx, ty, tz);
  for (int i = 4 * threadIdx.x; i < G.rp; i += 256) *reinterpret_cast<float4 *>(row + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const float offx = (float)(-tx * pt + nb), offy = (float)(-ty * pt + nb), offz = (float)(-tz * pt + nb);  // :134
  const int wlo = NGP ? 4 : 0, whi = NGP ? nf - 4 : nf;                                                  // window [wlo,whi)
  const bool row_possible = NGP ? (j >= wlo && j <= whi && k >= wlo && k <= whi) : true;
  if (row_possible && NGP) {
    for (int dk = -1; dk <= 0; dk++) {
      const int ks = k + dk; if (ks < wlo || ks >= whi) continue;   // source cell row must be inside the window
      for (int dj = -1; dj <= 0; dj++) {
        const int js = j + dj; if (js < wlo || js >= whi) continue;
        const int64_t rb = ((int64_t)(tz * pt + ks) * E + (ty * pt + js)) * E + tx * pt;
        const int p0 = cs[rb + wlo], p1 = cs[rb + whi];
        for (int s = p0 + threadIdx.x; s < p1; s += 64) {
          const float4 p = spos[s];
          const float x = p.x + offx, y = p.y + offy, z = p.z + offz;                                 // :139
          const int i1 = (int)floorf(x), j1 = (int)floorf(y), k1 = (int)floorf(z);                   // 0-based (:143 minus 1)
          if (j1 == j && k1 == k) atomicAdd(&row[i1], mass_p);                                         // :148
        }
      }
    }
  }
  if (!NGP) {
    // CIC, fine_cic_mass.f90:17-43 / fine_cic_mass_buffer.f90:25-53 (clipped to 1..nf).  ds_add_f32 costs ~2.6 clocks per
    // lane on this part (tools/ldsbench.hip: 0.2 T updates/s against 1.16 T/s for a read-add-write), so the scatter avoids
    // it: the 64 records of a chunk come from ONE sorted cell row, their cells ascend, equal cells are neighbours.  The
    // first lane of every run of equal cells updates the row buffer with a plain read-add-write -- no two of them touch the
    // same word -- and the (rare) other lanes of a run follow with atomics; the lower and the upper cell of the records are
    // two such rounds, and the LDS operations of a wavefront complete in order.  A chunk whose cells do not ascend (xv +
    // offset rounded a record into the next cell ahead of a neighbour of its own cell) takes the atomics for everybody.
    auto scatter = [&](const float4 &p, bool live) {   // a wavefront operation: every lane calls
      const float x = p.x + offx, y = p.y + offy, z = p.z + offz;                                 // :139
      const int i1 = live ? (int)floorf(x) : 0x3fffffff, j1 = (int)floorf(y), k1 = (int)floorf(z);   // 0-based (:143 minus 1)
      const float dx1 = (float)(i1 + 1) - x, dy1 = (float)(j1 + 1) - y, dz1 = (float)(k1 + 1) - z;
      const float dx2 = 1.f - dx1, dy2 = 1.f - dy1, dz2 = 1.f - dz1;
      float wy = 0.f, wz = 0.f; bool use = live;
      if (j1 == j) wy = dy1; else if (j1 + 1 == j) wy = dy2; else use = false;
      if (k1 == k) wz = dz1; else if (k1 + 1 == k) wz = dz2; else use = false;
      const float mx1 = mass_p * dx1, mx2 = mass_p * dx2;                                          // :23-24
      const float w1 = mx1 * wy * wz, w2 = mx2 * wy * wz;
      const int prev = __builtin_amdgcn_update_dpp(i1, i1, 0x138, 0xf, 0xf, false);   // wave_shr:1 (not the LDS round trip of __shfl_up; lane 0 is a first lane anyway)
      const bool first = threadIdx.x == 0 || prev != i1;          // first lane of a run of equal cells
      const bool ascending = __ballot(threadIdx.x != 0 && prev > i1) == 0ull;
      const bool plain = ascending && first;
      if (use && plain) { const float v = row[i1]; row[i1] = v + w1; }
      if (use && !plain) atomicAdd(&row[i1], w1);
      if (use && i1 + 1 < nf) {
        if (plain) { const float v = row[i1 + 1]; row[i1 + 1] = v + w2; } else atomicAdd(&row[i1 + 1], w2);
      }
    };
    // The four source cell rows {k-1,k} x {j-1,j} in the order of the reference's sums; their extents first, then the first 128
    // records of each (a row holds ~70 at the reference's density), all in flight together: one workgroup used to wait for
    // eight to twelve dependent round trips (extent, chunk, chunk per source row) and the launch was bound by that chain
    int p0[4], p1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ks = k - 1 + (q >> 1), js = j - 1 + (q & 1);
      p0[q] = 0; p1[q] = 0;
      if (ks >= wlo && ks < whi && js >= wlo && js < whi) {
        const int64_t rb = ((int64_t)(tz * pt + ks) * E + (ty * pt + js)) * E + tx * pt;
        p0[q] = cs[rb + wlo]; p1[q] = cs[rb + whi];
      }
    }
    float4 rec[4][2];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int s = p0[q] + u * 64 + (int)threadIdx.x;
        rec[q][u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < p1[q]) rec[q][u] = spos[s];
      }
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int u = 0; u < 2; u++)
        if (p0[q] + u * 64 < p1[q]) scatter(rec[q][u], p0[q] + u * 64 + (int)threadIdx.x < p1[q]);   // uniform
      for (int s0 = p0[q] + 128; s0 < p1[q]; s0 += 64) {   // uniform trip count
        const int s = s0 + (int)threadIdx.x;
        const bool live = s < p1[q];
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) p = spos[s];
        scatter(p, live);
      }
    }
  }
  __syncthreads();
  float *out = rho + ((int64_t)tl * nf * nf + (int64_t)k * nf + j) * G.rp;
  float part = 0.f;
  const bool interior_row = (j >= nb && j < nf - nb && k >= nb && k < nf - nb);
  for (int i = 4 * threadIdx.x; i < G.rp; i += 256) {   // (rp is a multiple of four: 16-byte stores)
    const float4 v = *reinterpret_cast<const float4 *>(row + i);
    *reinterpret_cast<float4 *>(out + i) = v;
    if (interior_row) {                                                                                  // :167-173
      if (i >= nb && i < nf - nb) part += v.x;
      if (i + 1 >= nb && i + 1 < nf - nb) part += v.y;
      if (i + 2 >= nb && i + 2 < nf - nb) part += v.z;
      if (i + 3 >= nb && i + 3 < nf - nb) part += v.w;
    }
  }
  if (sum_interior) {
    part = wave_scan_incl_f(part);   // lane 63: the row's sum (DPP)
    if (threadIdx.x == 63 && interior_row && part != 0.f) atomicAdd(sum_interior + p3m_slot() * 8, (double)part);
  }
}

// ------------------------------------------------------------------ NGP deposit from the cell counts
// rho(cell) = mass_p added count(cell) times (:148), count = cs[c+1]-cs[c]: a streaming pass over the
// sorted cell offsets, no particle reads, no atomics.  Cells outside the NGP chain window [4,nf-4)
// (:120-121) are zero.  The handful of records whose xv+offset rounds into the next cell are moved
// afterwards by k_ngp_fixup.
__global__ __launch_bounds__(256) void k_ngp_counts(const int *__restrict__ cs, float *__restrict__ rho, int tile0, int ntile, TileGeo G,
                                                    float mass_p, double *__restrict__ sum_interior) {
  // grid: (row groups of RG rows, nf planes, tiles); lanes run along x: coalesced 4-byte loads and stores
  constexpr int RG = 8;
  __shared__ float sh[4];
  const int nf = G.nf, E = G.E, pt = G.pt, nb = G.nb, pitch = G.rp;
  const int k = blockIdx.y, tl = blockIdx.z;
  int tx, ty, tz; tile_xyz(tile0 + tl, G.T, tx, ty, tz);
  float part = 0.f;
  for (int jr = 0; jr < RG; jr++) {
    const int j = blockIdx.x * RG + jr;
    if (j >= nf) break;
    float *out = rho + (((int64_t)tl * nf + k) * nf + j) * pitch;
    const bool row_in = (j >= 4 && j < nf - 4 && k >= 4 && k < nf - 4);
    const bool row_int = (j >= nb && j < nf - nb && k >= nb && k < nf - nb);
    const int *row = cs + ((int64_t)(tz * pt + k) * E + (ty * pt + j)) * E + tx * pt;
    for (int i = threadIdx.x; i < pitch; i += 256) {
      float r = 0.f;
      if (row_in && i >= 4 && i < nf - 4) {
        const int cnt = row[i + 1] - row[i];
        for (int q = 0; q < cnt; q++) r = r + mass_p;                      // :148, same partial sums
        if (row_int && i >= nb && i < nf - nb) part += r;                    // :167-173
      }
      out[i] = r;
    }
  }
  if (sum_interior) {
    const float s = block_sum_f(part, sh);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(sum_interior + p3m_slot() * 8, (double)s);
  }
}
// candidates: sorted indices of records within 2^-10 below a cell face in some coordinate, in P3M_CAND_SLOTS lists (k_row_sort;
// blockIdx.y = list).  A list overflowed (cand_cnt[16 * slots] != 0): every sorted record is looked at instead, by the same launch
// The fused inverse-x + kick pass (kick_fused.hip) kicks a record from the box rows of ITS batch of `fuse_nr` rows: a physical record of
// this tile whose reference cell lies in a row of another batch than the row it is sorted into gets that row flagged here -- the pass
// then stores the row to the force box as well and k_kick_fix kicks the record from there.
struct FuseFlag { unsigned char *rowflag; int nr, Nn, ms; bool u8; float *cmax; };   // u8: the density is one byte per cell (RowDep::rho8); cmax: where a saturated byte is reported
__device__ __forceinline__ bool owner_is(const float4 &p, const TileGeo &G, int Nn, int ms, int tx, int ty, int tz) {
  const float fNn = (float)Nn;
  if (!(p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn)) return false;   // chains of hoc(1..ncn) only (:234-236)
  const int nct = G.pt / ms;
  return G.T == 1 || (((int)floorf(p.x / (float)ms)) / nct == tx && ((int)floorf(p.y / (float)ms)) / nct == ty && ((int)floorf(p.z / (float)ms)) / nct == tz);
}
__device__ __forceinline__ void ngp_fixup_record(const float4 &p, float *__restrict__ rho, int tile0, int tl, const TileGeo &G, float mass_p, double *__restrict__ sum_interior,
                                                 const FuseFlag &ff) {
  const int nf = G.nf, pt = G.pt, nb = G.nb;
  int t3[3]; tile_xyz(tile0 + tl, G.T, t3[0], t3[1], t3[2]);
  const float xs[3] = {p.x, p.y, p.z};
  int gl[3], rr[3]; bool member = true, moved = false;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    gl[d] = (int)floorf(xs[d]) + nb - t3[d] * pt;                       // cell the count-based deposit used
    member = member && gl[d] >= 4 && gl[d] < nf - 4;                     // chain window (:120-121)
    rr[d] = (int)floorf(xs[d] + (float)(-t3[d] * pt + nb));             // the reference's cell (:134,:139,:143)
    moved = moved || (rr[d] != gl[d]);
  }
  if (!member || !moved) return;
  if (ff.rowflag && (rr[1] != gl[1] || rr[2] != gl[2]) && owner_is(p, G, ff.Nn, ff.ms, t3[0], t3[1], t3[2])) {
    const int lo = nb - 2, tile = tile0 + tl;
    const int bs = (tile * G.fb + (gl[2] - lo)) * G.fb + (gl[1] - lo), br = (tile * G.fb + (rr[2] - lo)) * G.fb + (rr[1] - lo);
    if (bs / ff.nr != br / ff.nr) ff.rowflag[br] = 1;
  }
  if (ff.u8) {
    // one byte per cell (RowDep::rho8): -1 / +1 on the cell's byte inside its word.  The -1 never borrows (the record itself was counted
    // there); a +1 onto 255 would carry into the neighbour: it is left out and reported like a saturated count (the step then fails)
    unsigned *w32 = reinterpret_cast<unsigned *>(rho) + (int64_t)tl * nf * nf * (G.rp / 4);
    const int64_t ig8 = ((int64_t)gl[2] * nf + gl[1]) * G.rp + gl[0], ir8 = ((int64_t)rr[2] * nf + rr[1]) * G.rp + rr[0];
    atomicSub(&w32[ig8 >> 2], 1u << (8 * (ig8 & 3)));
    const unsigned old = atomicAdd(&w32[ir8 >> 2], 1u << (8 * (ir8 & 3)));
    if (((old >> (8 * (ir8 & 3))) & 255u) == 255u) { atomicSub(&w32[ir8 >> 2], 1u << (8 * (ir8 & 3))); p3m_atomic_max_nonneg(ff.cmax, 1.0e9f); }
  } else {
  float *base = rho + (int64_t)tl * nf * nf * G.rp;
  atomicAdd(&base[((int64_t)gl[2] * nf + gl[1]) * G.rp + gl[0]], -mass_p);
  atomicAdd(&base[((int64_t)rr[2] * nf + rr[1]) * G.rp + rr[0]], mass_p);
  }
  if (sum_interior) {
    const bool ig = gl[0] >= nb && gl[0] < nf - nb && gl[1] >= nb && gl[1] < nf - nb && gl[2] >= nb && gl[2] < nf - nb;
    const bool ir = rr[0] >= nb && rr[0] < nf - nb && rr[1] >= nb && rr[1] < nf - nb && rr[2] >= nb && rr[2] < nf - nb;
    if (ig != ir) atomicAdd(sum_interior, ir ? (double)mass_p : -(double)mass_p);
  }
}
__global__ __launch_bounds__(256) void k_ngp_fixup(const float4 *__restrict__ spos, int nrec, const int *__restrict__ cand, const int *__restrict__ cand_cnt, int cand_seg,
                                                   float *__restrict__ rho, int tile0, int ntile, TileGeo G, float mass_p, double *__restrict__ sum_interior, FuseFlag ff) {
  // the candidate counts stay on the device (written by k_row_sort of this step): no host round trip between sort and deposit
  if (cand_cnt[16 * P3M_CAND_SLOTS] != 0) {   // a list overflowed: every sorted record is looked at instead, by the same grid (one launch either way)
    const float thr = 1.0f - 0.0009765625f;
    const int64_t nthr = (int64_t)gridDim.x * gridDim.y * 256;
    for (int64_t id = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; id < (int64_t)nrec * ntile; id += nthr) {
      const int tl = (int)(id / nrec); const int s = (int)(id - (int64_t)tl * nrec);
      const float4 p = spos[s];
      if ((p.x - floorf(p.x) >= thr) || (p.y - floorf(p.y) >= thr) || (p.z - floorf(p.z) >= thr)) ngp_fixup_record(p, rho, tile0, tl, G, mass_p, sum_interior, ff);
    }
    return;
  }
  const int slot = blockIdx.y, ncand = min(cand_cnt[slot * 16], cand_seg);
  const int *list = cand + (int64_t)slot * cand_seg;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < (int64_t)ncand * ntile; id += (int64_t)gridDim.x * 256) {
    const int tl = (int)(id / ncand); const int ci = (int)(id - (int64_t)tl * ncand);
    ngp_fixup_record(spos[list[ci]], rho, tile0, tl, G, mass_p, sum_interior, ff);
  }
}

int fine_deposit(p3m_ctx *c, int tile0, int ntile, float mass_p, bool fuse) {   // fuse: the kick will run inside the inverse x pass (flag rows for it)
  const Geometry &g = c->g;
  TileGeo G{g.T, g.nf, g.nb, g.pt, g.E, g.fb, 2 * g.px, g.fbp};
  if (c->p.flags & P3M_FLAG_NGP) {
    const bool have = c->rho_from_sort && tile0 == 0 && ntile == g.ntiles;   // written by k_row_sort of this step
    c->rho_u8_force = have && c->rho_u8;               // ... as one byte per cell: the fix-up and fine_force's forward x pass read bytes
    c->rho_mass = mass_p;
    c->rho_from_sort = false; c->rho_u8 = false;
    if (!have) P3M_TRY(particles_full_cells(c));
    if (!have) hipLaunchKernelGGL(k_ngp_counts, dim3(cdiv(g.nf, 8), g.nf, ntile), dim3(256), 0, c->stream, (const int *)c->cell_end,
                       c->rho, tile0, ntile, G, mass_p, c->d_sums);
    HIP_TRY(hipGetLastError());
    if (c->np_all > 0) {
      // records within 2^-10 below a cell face: ~0.3 % of the records; the grids are sized for that share, the loops cover any count
      const int64_t guess = std::max<int64_t>(1, (int64_t)c->np_all / 256 / P3M_CAND_SLOTS) * ntile;
      const FuseFlag ff{fuse ? c->rowflag : (unsigned char *)nullptr, c->fuse_nr, g.Nn, g.ms, c->rho_u8_force, c->d_red + 3 * P3M_RED_SPAN};
      hipLaunchKernelGGL(k_ngp_fixup, dim3((unsigned)std::min<int64_t>(256, cdiv(guess, 256)), P3M_CAND_SLOTS), dim3(256), 0, c->stream, (const float4 *)c->spos, c->np_all,
                         (const int *)c->cand, (const int *)c->cand_cnt, c->cand_seg, c->rho, tile0, ntile, G, mass_p, c->d_sums, ff);
      HIP_TRY(hipGetLastError());
    }
    return P3M_OK;
  }
  P3M_TRY(particles_full_cells(c));
  const unsigned blocks = (unsigned)((int64_t)ntile * g.nf * g.nf);
  const size_t lds = sizeof(float) * (2 * g.px);
  hipLaunchKernelGGL(k_fine_deposit<false>, dim3(8 * cdiv(blocks, 8)), dim3(64), lds, c->stream, (const float4 *)c->spos, (const int *)c->cell_end, c->rho,
                     tile0, G, mass_p, c->d_sums, (int)blocks);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// ------------------------------------------------------------------ projection.f90
// build_projection (:126-188): the CIC density of a tile (fine_cic_mass.f90, whatever the deposit of the force path is)
// summed along each axis into global nf_physical_dim^2 maps.  One thread owns one map cell and adds the tile's
// interior cells in the reference's order (k, j or i ascending; tiles in launch order), so that the sums differ from the
// reference's only through the deposit's own addition order.  Maps: pxy[y][x], pxz[z][x], pyz[z][y] (the reference's
// column-major (x,y), (x,z), (y,z)).
__global__ __launch_bounds__(256) void k_project(const float *__restrict__ rho, int nf, int rp, int nb, int pt, int Np, int osx, int osy, int osz,
                                                 float *__restrict__ pxy, float *__restrict__ pxz, float *__restrict__ pyz, int do_xy, int do_xz, int do_yz) {
  const int t = blockIdx.x * 256 + threadIdx.x, which = blockIdx.y;
  if (t >= pt * pt) return;
  const int a = t % pt, b = t / pt;
  auto R = [&](int i, int j, int k) { return rho[((int64_t)(nb + k) * nf + (nb + j)) * rp + (nb + i)]; };
  if (which == 0 && do_xy) {          // (i, j) = (a, b), sum over k (:171-172)
    float *o = pxy + (int64_t)(osy + b) * Np + (osx + a); float acc = *o;
    for (int k = 0; k < pt; k++) acc = acc + R(a, b, k);
    *o = acc;
  } else if (which == 1 && do_xz) {   // (i, k) = (a, b), sum over j (:175-176)
    float *o = pxz + (int64_t)(osz + b) * Np + (osx + a); float acc = *o;
    for (int j = 0; j < pt; j++) acc = acc + R(a, j, b);
    *o = acc;
  } else if (which == 2 && do_yz) {   // (j, k) = (a, b), sum over i (:179-180)
    float *o = pyz + (int64_t)(osz + b) * Np + (osy + a); float acc = *o;
    for (int i = 0; i < pt; i++) acc = acc + R(i, a, b);
    *o = acc;
  }
}
__global__ __launch_bounds__(256) void k_project_mass(const float *__restrict__ rho, int nf, int rp, int nb, int pt, double *__restrict__ out) {
  __shared__ double sh[4];
  double part = 0.0;
  const int64_t tot = (int64_t)pt * pt * pt;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += (int64_t)gridDim.x * 256) {
    const int i = (int)(t % pt), j = (int)((t / pt) % pt), k = (int)(t / ((int64_t)pt * pt));
    part += (double)rho[((int64_t)(nb + k) * nf + (nb + j)) * rp + (nb + i)];                      // :183
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + p3m_slot() * 8, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}
// adds this context's tiles to the device maps (Np^2 floats each); the records must be sorted with their ghosts
int fine_projection(p3m_ctx *c, float mass_p, float *d_pxy, float *d_pxz, float *d_pyz) {
  const Geometry &g = c->g;
  TileGeo G{g.T, g.nf, g.nb, g.pt, g.E, g.fb, 2 * g.px, g.fbp};
  const int Np = g.Nn * g.nodes_dim;
  const size_t S = (size_t)(2 * g.px) * g.nf * g.nf;
  c->rho_from_sort = false; c->rho_u8 = false; c->rho_u8_force = false;
  P3M_TRY(particles_full_cells(c));
  for (int t0 = 0; t0 < g.ntiles; t0 += c->tile_batch) {
    const int nt = std::min(c->tile_batch, g.ntiles - t0);
    hipLaunchKernelGGL(k_fine_deposit<false>, dim3(8 * cdiv((int64_t)nt * g.nf * g.nf, 8)), dim3(64), sizeof(float) * (2 * g.px), c->stream,
                       (const float4 *)c->spos, (const int *)c->cell_end, c->rho, t0, G, mass_p, (double *)nullptr, (int)((int64_t)nt * g.nf * g.nf));
    HIP_TRY(hipGetLastError());
    for (int t = 0; t < nt; t++) {   // tile order of :24-32 (x fastest)
      const int tl = t0 + t, tz = tl / (g.T * g.T), ty = (tl / g.T) % g.T, tx = tl % g.T;
      const float *rho = c->rho + (size_t)t * S;
      hipLaunchKernelGGL(k_project, dim3(cdiv(g.pt * g.pt, 256), 3), dim3(256), 0, c->stream, rho, g.nf, 2 * g.px, g.nb, g.pt, Np, tx * g.pt + g.cart[2] * g.Nn,
                         ty * g.pt + g.cart[1] * g.Nn, tz * g.pt + g.cart[0] * g.Nn, d_pxy, d_pxz, d_pyz, g.cart[0] == 0, g.cart[1] == 0, g.cart[2] == 0);
      HIP_TRY(hipGetLastError());
      hipLaunchKernelGGL(k_project_mass, dim3(std::min(1024, cdiv((int64_t)g.pt * g.pt * g.pt, 256))), dim3(256), 0, c->stream, rho, g.nf, 2 * g.px, g.nb, g.pt,
                         c->d_sums + 3 * P3M_SUM_SPAN);
      HIP_TRY(hipGetLastError());
    }
  }
  return P3M_OK;
}

// ------------------------------------------------------------------ :176-204 forward FFT, 3 x (i K_c multiply, inverse FFT, box extract)
int fine_force(p3m_ctx *c, int tile0, int ntile, bool defer_x) {
  const Geometry &g = c->g;
  // forward x and y passes; the forward z pass is the prologue of the inverse z pass (rho-hat never touches HBM)
  P3M_TRY(fft3d_forward_xy(c, c->plan_f, c->rho, c->work, ntile, c->rho_u8_force, c->rho_mass));
  c->rho_u8_force = false;
  const size_t boxsz = (size_t)g.fb * g.fb * g.fbp;
  if (defer_x) {   // whole NGP steps: the inverse x pass runs with the kick (fine_xinv_kick_fused), no force box
    P3M_TRY(fft_inverse3_box_z(c, c->plan_f, c->rho, c->work, c->kern_f, ntile, g.fb, g.nb - 2, true));
    return fft_inverse3_box_y(c, c->plan_f, c->work, ntile, g.fb, g.nb - 2);
  }
  // one fused launch per axis for all three components
  return fft_inverse3_box(c, c->plan_f, c->rho, c->work, c->kern_f, ntile, c->fbox + (size_t)tile0 * boxsz, g.fb, g.nb - 2,
                          (int64_t)g.ntiles * boxsz, true);
}

// ------------------------------------------------------------------ :208-223 max |F|^2 over every tile's force box
__global__ __launch_bounds__(256) void k_force_max(const float *__restrict__ fbox, int64_t n, int64_t comp_stride, float *__restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = fbox[i], b = fbox[i + comp_stride], d = fbox[i + 2 * comp_stride];
    const float f = a * a + b * b + d * d;                                                               // :217-218
    m = fmaxf(m, f);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) p3m_atomic_max_nonneg(out + p3m_slot() * 16, m);    // m >= 0
}
int fine_force_max(p3m_ctx *c) {
  const Geometry &g = c->g;
  const int64_t n = (int64_t)g.ntiles * g.fb * g.fb * g.fbp;   // pad columns hold zeros
  hipLaunchKernelGGL(k_force_max, dim3(std::min<int64_t>(2048, cdiv(n, 256))), dim3(256), 0, c->stream, (const float *)c->fbox, n, n, c->d_red);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// ------------------------------------------------------------------ :208-319 CIC: max |F|^2, gather + kick of the physical particles
// One workgroup per BLOCK of the force box: CK_BK x CK_BJ cell rows, CK_XS cells long.  A record interpolates between the eight
// box points around it, so the block stages (CK_BK + 1) x (CK_BJ + 1) box rows of CK_XS + 4 points, all three components, in LDS
// -- 1.6 times its own volume, 39 KB -- with coalesced 16-byte loads, forms the maximum of |F|^2 over the points that are its
// own on the way (every box point is some block's own exactly once: the pass over the box that k_force_max used to be), and its
// records -- CK_BK * CK_BJ short ranges of the sorted store, ~256 in all, flattened over the threads -- take their 24 values from
// LDS.  (The per-record gather from global memory this replaces re-fetched every box row for four cell rows: 753 us + 313 us for
// the maximum per 560-tile; one wavefront per cell row with its four box rows in LDS was tried in round 3: 25 KB per wavefront,
// six wavefronts per CU, 1.99 ms.)  Blocks go to the XCDs in contiguous eighths (an XCD's L2 then holds the rows its neighbouring
// blocks share).  Same terms, same order of the eight corners as :293-316.  COARSE: the coarse-mesh kick
// (coarse_velocity.f90:137-179, same arithmetic and order as k_coarse_kick) follows the fine kick of a record in registers --
// PM-only whole steps, where nothing else touches the velocities between the two kicks.  A record whose reference cell
// floor(xv + offset_tile) lies outside the staged rows (rounding at a face of the block) reads global memory.
// (Round 4 also tried to start the chain range -> record -> velocity ahead of the barrier -- ranges by one vector load per wavefront and
// v_readlane, the thread's record and velocity requested before / underneath the staging loads: 990-1000 us against 870; with the
// ranges as scalar loads 1360 us: 32 misses of the scalar cache are served one after the other.)
#define CK_BK 4
#define CK_BJ 4
#define CK_XS 124      // cells of an x segment at most: CK_XS + 4 points = 32 groups of four, one per lane of half a wavefront
#define CK_XP (CK_XS + 4)
#define CK_NU (((CK_BK + 1) * (CK_BJ + 1) + 7) / 8)   // staging items (three 16-byte loads each) per thread: thread (q = tid % 32, r0 = tid / 32) takes the four-point
                                                   // group q of the box rows r0, r0 + 8, ...  (Flattened as e = tid + 256 u with e % nx4, e / nx4 % nj, e / nx4 / nj -- divisions
                                                   // by run-time numbers, ~35 vector instructions each -- the index arithmetic was 800 of the kernel's ~1060 vector instructions per
                                                   // wavefront and block: profiles/r05_cic_sq_counters.txt)
// Round 5: PERSISTENT workgroups, software-pipelined.  One block at a time the pass was a chain of four round trips (ranges, box,
// record, velocity) per block with four blocks per CU to hide it: 2.7 TB/s.  Now a workgroup walks a list of blocks, and the box
// points and ranges of block i + 1 are requested into registers (12 x 16 bytes per thread) before the records of block i are
// worked: the staging loads fly under the record -> velocity chain.  x segments are cut evenly (xsl cells: 5 x 104 at fb = 515
// instead of 4 x 128 + 3).  897 -> 873 us per 560-tile: less than hoped.  Ablations of this kernel in the step: without the record phase
// ~510 us (1.64 GB of box at 3.2 TB/s: the k halo plane of a block is re-read 645 blocks later, past the L2), the records add ~360, of
// which the coarse kick 70.  A workgroup that marches along k and keeps the plane it shares with its next block (four planes fetched per
// block instead of five, the re-read from beyond the L2 gone) was built next: 900 us.  Neither the bytes nor the round trips, then.
__device__ __forceinline__ int ck_mj(int nj) { return nj >= 5 ? 52 : nj == 4 ? 64 : nj == 3 ? 86 : nj == 2 ? 128 : 256; }   // ceil(256 / nj)
struct CicGeom { int k0, j0, x0, nk, nj, nx4, tx, ty, tz; const float *f0; int64_t row; };
typedef __attribute__((address_space(3))) float lds_cfloat;
template <bool COARSE>
__global__ __launch_bounds__(256) void k_fine_kick_cic(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, TileGeo G, int Nn, int ms,
                                                       const float *__restrict__ fbox, int64_t comp_stride, float a_mid, float dt, float *__restrict__ fmax_out,
                                                       const float *__restrict__ fc, int ncn, int nxs, int nbj, int nbk, int nblk, int xsl) {
  __shared__ float sb[(CK_BK + 1) * (CK_BJ + 1) * 3 * CK_XP];
  __shared__ int rp0[CK_BK * CK_BJ], rpre[CK_BK * CK_BJ + 1];
  const int tid = threadIdx.x, fb = G.fb, fbp = G.fbp, lo = G.nb - 2;
  // workgroups go to the XCDs in turn: XCD x works the contiguous eighth [x*per, (x+1)*per) of the blocks, its workgroups striding it
  const int per = (nblk + 7) >> 3, xcd = (int)(blockIdx.x & 7u), nsl = (int)(gridDim.x >> 3), lend = min(nblk, (xcd + 1) * per);
  int lb = xcd * per + (int)(blockIdx.x >> 3);
  const fdiv_t d_xs = mk_fdiv(nxs), d_bj = mk_fdiv(nbj), d_bk = mk_fdiv(nbk);
  auto geom = [&](int b) {
    CicGeom g;
    const int t1 = fdiv(b, d_xs), xs = b - t1 * nxs, t2 = fdiv(t1, d_bj), bj = t1 - t2 * nbj, tile = fdiv(t2, d_bk), bk = t2 - tile * nbk;   // (one mulhi each: nblk * max(nxs, nbj, nbk) < 2^32)
    tile_xyz(tile, G.T, g.tx, g.ty, g.tz);
    g.k0 = bk * CK_BK; g.j0 = bj * CK_BJ; g.x0 = xs * xsl;
    g.nk = min(CK_BK + 1, fb - g.k0); g.nj = min(CK_BJ + 1, fb - g.j0); g.nx4 = min(xsl + 4, fbp - g.x0) >> 2;
    g.f0 = fbox + (int64_t)tile * fb * fb * fbp;
    return g;
  };
  float4 ra[CK_NU], rb[CK_NU], rd[CK_NU]; int st = 0, rcnt = 0;
  auto request = [&](const CicGeom &g) {   // the box points and (threads 0..15) the cell rows' record ranges of a block, into registers
    st = 0; rcnt = 0;
    if (tid < CK_BK * CK_BJ) {
      const int kk = g.k0 + tid / CK_BJ, jj = g.j0 + tid % CK_BJ;
      if (kk < fb && jj < fb) {
        const int64_t row = ((int64_t)(g.tz * G.pt + kk + lo) * G.E + (g.ty * G.pt + jj + lo)) * G.E + g.tx * G.pt + lo;   // box column i <-> extended cell tx*pt + lo + i
        st = cs[row + g.x0]; rcnt = cs[row + min(g.x0 + xsl, fb)] - st;
      }
    }
    const int q = tid & 31, mj = ck_mj(g.nj);   // row / nj = row * mj >> 8 for row < 32, nj <= 5
#pragma unroll
    for (int u = 0; u < CK_NU; u++) {
      const int row = (tid >> 5) + 8 * u, rk = (row * mj) >> 8, rj = row - rk * g.nj;
      ra[u] = rb[u] = rd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < g.nx4 && row < g.nk * g.nj) {
        const float *src = g.f0 + ((int64_t)(g.k0 + rk) * fb + (g.j0 + rj)) * fbp + g.x0 + 4 * q;
        ra[u] = *reinterpret_cast<const float4 *>(src); rb[u] = *reinterpret_cast<const float4 *>(src + comp_stride); rd[u] = *reinterpret_cast<const float4 *>(src + 2 * comp_stride);
      }
    }
  };
  float m = 0.f;
  const float fNn = (float)Nn;
  const int nct = G.pt / ms;
  CicGeom g = geom(lb < lend ? lb : 0);
  if (lb < lend) request(g);
  for (; lb < lend; lb += nsl) {
    // ---- the requested block into LDS; the maximum over its own points (pad columns are zero)
    const int q = tid & 31, mj = ck_mj(g.nj);
#pragma unroll
    for (int u = 0; u < CK_NU; u++) {
      const int row = (tid >> 5) + 8 * u, rk = (row * mj) >> 8, rj = row - rk * g.nj;
      if (q < g.nx4 && row < g.nk * g.nj) {
        const float4 a = ra[u], b = rb[u], d = rd[u];
        float *dst = sb + (rk * (CK_BJ + 1) + rj) * 3 * CK_XP + 4 * q;
        *reinterpret_cast<float4 *>(dst) = a; *reinterpret_cast<float4 *>(dst + CK_XP) = b; *reinterpret_cast<float4 *>(dst + 2 * CK_XP) = d;
        if (rk < CK_BK && rj < CK_BJ && 4 * q < xsl)                                                                   // this block's own points
          m = fmaxf(m, fmaxf(fmaxf(a.x * a.x + b.x * b.x + d.x * d.x, a.y * a.y + b.y * b.y + d.y * d.y),              // :217-218
                             fmaxf(a.z * a.z + b.z * b.z + d.z * d.z, a.w * a.w + b.w * b.w + d.w * d.w)));
      }
    }
    if (tid < CK_BK * CK_BJ) rp0[tid] = st;
    if (tid < 64) {   // exclusive prefix of the CK_BK * CK_BJ counts
      const int inc = wave_scan_incl_i(rcnt);   // (DPP; lanes past the sixteen rows hold zero)
      if (tid < CK_BK * CK_BJ) rpre[tid] = inc - rcnt;
      if (tid == CK_BK * CK_BJ - 1) rpre[CK_BK * CK_BJ] = inc;
    }
    __syncthreads();
    const CicGeom c = g;
    if (lb + nsl < lend) { g = geom(lb + nsl); request(g); }   // the next block: in flight under this block's records
    // ---- the records of the block's cell rows
    const int total = rpre[CK_BK * CK_BJ];
    const int nx = 4 * c.nx4;
    const float offx = (float)G.nb - (float)(c.tx * G.pt), offy = (float)G.nb - (float)(c.ty * G.pt), offz = (float)G.nb - (float)(c.tz * G.pt);  // :227
    for (int t = tid; t < total; t += 256) {
      int r = 0;
#pragma unroll
      for (int k = 1; k < CK_BK * CK_BJ; k++) r += (rpre[k] <= t) ? 1 : 0;
      const float4 p = spos[rp0[r] + (t - rpre[r])];
      if (!(p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn)) continue;  // chains of hoc(1..ncn) only (:234-236)
      // owner tile from the coarse cell of the chain the particle sits in: hoc index floor(x/mesh_scale)+1 (link_list.f90:19-21)
      if (G.T > 1 && (((int)floorf(p.x / (float)ms)) / nct != c.tx || ((int)floorf(p.y / (float)ms)) / nct != c.ty || ((int)floorf(p.z / (float)ms)) / nct != c.tz)) continue;
      const int vi = rec_index(p);   // the velocity stays in arrival order (p3m_internal.h)
      float4 v = vel[vi];
      const float x = p.x + offx, y = p.y + offy, z = p.z + offz;                                        // :248
      const int i1 = (int)floorf(x) - lo, j1 = (int)floorf(y) - lo, k1 = (int)floorf(z) - lo;          // index into the force box
      const float dx1 = (float)(i1 + lo + 1) - x, dy1 = (float)(j1 + lo + 1) - y, dz1 = (float)(k1 + lo + 1) - z;  // :290
      const float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1, dz2 = 1.0f - dz1;
      const int li = i1 - c.x0, lj = j1 - c.j0, lk = k1 - c.k0;
      const bool staged = li >= 0 && li + 1 < nx && lj >= 0 && lj + 1 < c.nj && lk >= 0 && lk + 1 < c.nk;
      // the 24 values first, through a pointer that is LDS by its TYPE: with `staged ? LDS : global` inside the corner loop the compiler
      // selected between two generic pointers per lane and read both memories through 24 flat loads per record
      float f8[8][3];
      if (staged) {
        const lds_cfloat *q0 = (const lds_cfloat *)sb + (lk * (CK_BJ + 1) + lj) * 3 * CK_XP + li;
#pragma unroll
        for (int cn = 0; cn < 8; cn++) {
          const lds_cfloat *q = q0 + ((cn >> 2) * (CK_BJ + 1) + ((cn >> 1) & 1)) * 3 * CK_XP + (cn & 1);
          f8[cn][0] = q[0]; f8[cn][1] = q[CK_XP]; f8[cn][2] = q[2 * CK_XP];
        }
      } else {
#pragma unroll
        for (int cn = 0; cn < 8; cn++) {
          const int64_t o = ((int64_t)(k1 + (cn >> 2)) * fb + (j1 + ((cn >> 1) & 1))) * fbp + (i1 + (cn & 1));
          f8[cn][0] = c.f0[o]; f8[cn][1] = c.f0[o + comp_stride]; f8[cn][2] = c.f0[o + 2 * comp_stride];
        }
      }
#pragma unroll
      for (int cz = 0; cz < 2; cz++)
#pragma unroll
        for (int cy = 0; cy < 2; cy++)
#pragma unroll
          for (int cx = 0; cx < 2; cx++) {                                                                // order of :293-316
            const float dVc = a_mid * P3M_G_F * dt * (cx ? dx2 : dx1) * (cy ? dy2 : dy1) * (cz ? dz2 : dz1);
            const int cn = 4 * cz + 2 * cy + cx;
            v.x = v.x + f8[cn][0] * dVc; v.y = v.y + f8[cn][1] * dVc; v.z = v.z + f8[cn][2] * dVc;
          }
      if (COARSE) {
        const float inv = 1.0f / (float)ms;
        const float cx_ = inv * p.x - 0.5f, cy_ = inv * p.y - 0.5f, cz_ = inv * p.z - 0.5f;            // coarse_velocity.f90:143
        const int ci = (int)floorf(cx_) + 1, cj = (int)floorf(cy_) + 1, ck = (int)floorf(cz_) + 1;
        const float ex1 = (float)ci - cx_, ey1 = (float)cj - cy_, ez1 = (float)ck - cz_;
        const float ex2 = 1.0f - ex1, ey2 = 1.0f - ey1, ez2 = 1.0f - ez1;
        const int mm = ncn + 2; const int64_t ccs = (int64_t)mm * mm * mm;
#pragma unroll
        for (int cz = 0; cz < 2; cz++)
#pragma unroll
          for (int cy = 0; cy < 2; cy++)
#pragma unroll
            for (int cx = 0; cx < 2; cx++) {                                                              // :153-168
              const float dV = a_mid * P3M_G_F * dt * (cx ? ex2 : ex1) * (cy ? ey2 : ey1) * (cz ? ez2 : ez1);
              const int64_t o = ((int64_t)(ck + cz) * mm + (cj + cy)) * mm + (ci + cx);
              v.x = v.x + fc[o] * dV; v.y = v.y + fc[o + ccs] * dV; v.z = v.z + fc[o + 2 * ccs] * dV;
            }
      }
      vel[vi] = v;
    }
    __syncthreads();   // the block's LDS is rewritten at the top
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((tid & 63) == 0 && m > 0.f) p3m_atomic_max_nonneg(fmax_out + p3m_slot() * 16, m);
}

// NGP: max |F|^2 (:208-223) and the kick (:244-270) in ONE pass over the force box.  One wavefront per box row
// (tile, kk, jj): the three component rows go through LDS (coalesced 16-byte loads, each box byte is read once),
// the row's maximum goes to the reduction slots, and the physical records of the same cell row -- one contiguous
// range of the sorted store -- that this tile owns take their force from LDS.  A record whose reference cell
// floor(xv + offset_tile) is in another row than its sorted cell (rounding at a face) reads global memory.
// COARSE: the coarse-mesh kick of coarse_velocity.f90:137-179 (k_coarse_kick, same arithmetic, same order of the eight
// corner terms) follows the fine kick of a record in registers -- PM-only runs, where nothing else touches the
// velocities between the two kicks, so the sums are the ones two separate passes would form.
#ifndef P3M_KICK_WPB
#define P3M_KICK_WPB 1      // force-box rows (wavefronts) per workgroup
#endif
template <bool COARSE>
__global__ __launch_bounds__(64 * P3M_KICK_WPB) void k_fine_kick_rows(const float4 *__restrict__ spos, float4 *__restrict__ vel, const int *__restrict__ cs, TileGeo G,
                                                      int Nn, int ms, const float *__restrict__ fbox, int64_t comp_stride, float a_mid, float dt,
                                                      float *__restrict__ fmax_out, const float *__restrict__ fc, int ncn, const int *__restrict__ crow, int crow_w,
                                                      int *__restrict__ cnt256) {   // cnt256: survivors per block of 256 sorted records, counted on the way (or nullptr)
  extern __shared__ float frow_all[];   // [rows of the workgroup][3][fbp]
  const int fb = G.fb, fbp = G.fbp, lo = G.nb - 2, lane = threadIdx.x & 63;
  const int brow = blockIdx.x * P3M_KICK_WPB + (threadIdx.x >> 6);   // the wavefront's row of the force boxes; rows never wait for each other
  if (brow >= G.T * G.T * G.T * fb * fb) return;
  float *frow = frow_all + (threadIdx.x >> 6) * 3 * fbp;
  const int jj = brow % fb, kk = (brow / fb) % fb, tile = brow / (fb * fb);
  int tx, ty, tz; tile_xyz(tile, G.T, tx, ty, tz);
  const float *f0 = fbox + (int64_t)tile * fb * fb * fbp;
  const int64_t ro = ((int64_t)kk * fb + jj) * fbp;
  // tile-local cell l <-> extended cell l + t*pt; the box row (jj,kk) is the local cell row (jj+lo, kk+lo).  The chain row range ->
  // record -> velocity (reached through the arrival index in the record's fourth lane) is three dependent round trips: it is
  // started here, for the first 64 records of the row (a row holds ~70), so that it runs underneath the loads of the box row
  int p0, p1;
  if (crow) { const int *t = crow + ((int64_t)(tz * G.pt + kk + lo) * G.E + (ty * G.pt + jj + lo)) * crow_w + ncn + 2 + 2 * tx; p0 = t[0]; p1 = t[1]; }   // compact table
  else { const int64_t row = ((int64_t)(tz * G.pt + kk + lo) * G.E + (ty * G.pt + jj + lo)) * G.E + tx * G.pt + lo; p0 = cs[row]; p1 = cs[row + fb]; }
  float4 pf = make_float4(-1.f, -1.f, -1.f, 0.f), vf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p0 + lane < p1) { pf = spos[p0 + lane]; vf = vel[rec_index(pf)]; }
  float m = 0.f;
  for (int q = lane; q < (fbp >> 2); q += 64) {
    const float4 a = *reinterpret_cast<const float4 *>(f0 + ro + 4 * q), b = *reinterpret_cast<const float4 *>(f0 + ro + comp_stride + 4 * q),
                 d = *reinterpret_cast<const float4 *>(f0 + ro + 2 * comp_stride + 4 * q);
    *reinterpret_cast<float4 *>(frow + 4 * q) = a; *reinterpret_cast<float4 *>(frow + fbp + 4 * q) = b; *reinterpret_cast<float4 *>(frow + 2 * fbp + 4 * q) = d;
    m = fmaxf(m, fmaxf(fmaxf(a.x * a.x + b.x * b.x + d.x * d.x, a.y * a.y + b.y * b.y + d.y * d.y),                 // :217-218 (pad columns are zero)
                       fmaxf(a.z * a.z + b.z * b.z + d.z * d.z, a.w * a.w + b.w * b.w + d.w * d.w)));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if (lane == 0 && m > 0.f) p3m_atomic_max_nonneg(fmax_out + p3m_slot() * 16, m);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  const float fNn = (float)Nn;
  const int nct = G.pt / ms;
  const float offx = (float)G.nb - (float)(tx * G.pt), offy = (float)G.nb - (float)(ty * G.pt), offz = (float)G.nb - (float)(tz * G.pt);  // :227
  for (int s = p0 + lane; s < p1; s += 64) {
    const bool first = s < p0 + 64;   // uniform over the wavefront
    const float4 p = first ? pf : spos[s];
    if (!(p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn)) continue;  // chains of hoc(1..ncn) only (:234-236)
    // owner tile from the coarse cell of the chain the particle sits in: hoc index floor(x/mesh_scale)+1 (link_list.f90:19-21)
    // (one tile per rank: every physical record is this tile's -- the three float and three integer divisions were 40 % of the loop's instructions)
    if (G.T > 1 && (((int)floorf(p.x / (float)ms)) / nct != tx || ((int)floorf(p.y / (float)ms)) / nct != ty || ((int)floorf(p.z / (float)ms)) / nct != tz)) continue;
    if (cnt256) {
      // every physical record passes here exactly once (in its owner tile): they are the survivors of delete_particles.
      // The lanes hold consecutive sorted indices, i.e. at most two blocks of 256: one atomic per block and wavefront
      const int blk = s >> 8;
      const unsigned long long act = __ballot(1);
      const int first = __builtin_amdgcn_readlane(blk, __ffsll((long long)act) - 1);
      const unsigned long long m1 = __ballot(blk == first);
      if (lane == __ffsll((long long)m1) - 1) atomicAdd(&cnt256[first], __popcll(m1));
      if (blk != first) { const unsigned long long m2 = __ballot(1); if (lane == __ffsll((long long)m2) - 1) atomicAdd(&cnt256[blk], __popcll(m2)); }
    }
    const float x = p.x + offx, y = p.y + offy, z = p.z + offz;                                        // :248
    const int i1 = (int)floorf(x) - lo, j1 = (int)floorf(y) - lo, k1 = (int)floorf(z) - lo;          // index into the force box
    float fx, fy, fz;
    if (j1 == jj && k1 == kk) { const lds_cfloat *fl = (const lds_cfloat *)frow; fx = fl[i1]; fy = fl[fbp + i1]; fz = fl[2 * fbp + i1]; }   // (typed LDS: no flat loads)
    else { const int64_t o = ((int64_t)k1 * fb + j1) * fbp + i1; fx = f0[o]; fy = f0[o + comp_stride]; fz = f0[o + 2 * comp_stride]; }
    const int vi = rec_index(p);   // the velocity stays in arrival order (p3m_internal.h)
    float4 v = first ? vf : vel[vi];
    v.x = v.x + fx * a_mid * P3M_G_F * dt;                                                             // :265-266
    v.y = v.y + fy * a_mid * P3M_G_F * dt;
    v.z = v.z + fz * a_mid * P3M_G_F * dt;
    if (COARSE) {
      const float inv = 1.0f / (float)ms;
      const float cx_ = inv * p.x - 0.5f, cy_ = inv * p.y - 0.5f, cz_ = inv * p.z - 0.5f;            // coarse_velocity.f90:143
      const int ci = (int)floorf(cx_) + 1, cj = (int)floorf(cy_) + 1, ck = (int)floorf(cz_) + 1;
      const float dx1 = (float)ci - cx_, dy1 = (float)cj - cy_, dz1 = (float)ck - cz_;
      const float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1, dz2 = 1.0f - dz1;
      const int m = ncn + 2; const int64_t ccs = (int64_t)m * m * m;
#pragma unroll
      for (int cz = 0; cz < 2; cz++)
#pragma unroll
        for (int cy = 0; cy < 2; cy++)
#pragma unroll
          for (int cx = 0; cx < 2; cx++) {                                                              // :153-168
            const float dV = a_mid * P3M_G_F * dt * (cx ? dx2 : dx1) * (cy ? dy2 : dy1) * (cz ? dz2 : dz1);
            const int64_t o = ((int64_t)(ck + cz) * m + (cj + cy)) * m + (ci + cx);
            v.x = v.x + fc[o] * dV; v.y = v.y + fc[o + ccs] * dV; v.z = v.z + fc[o + 2 * ccs] * dV;
          }
    }
    vel[vi] = v;
  }
}

// ------------------------------------------------------------------ the fused inverse-x + NGP kick pass (kick_fused.hip) and its fix-up
// k_kick_fix: the records the fused pass left out -- reference cell floor(xv + offset_tile) in a box row of another batch than the row
// they are sorted into (flagged by k_ngp_fixup, stored to the box by the pass) -- get their kick (:244-270), their coarse kick
// and their survivor count here, from the box.  Candidates as in k_ngp_fixup (lists of k_row_sort; every record when a list overflowed).
template <bool COARSE>
__device__ __forceinline__ void kick_fix_record(const float4 &p, int s, const TileGeo &G, int Nn, int ms, int nr, unsigned char *__restrict__ rowflag, const float *__restrict__ fbox,
                                                int64_t comp_stride, float a_mid, float dt, float4 *__restrict__ vel, const float *__restrict__ fc, int ncn, int *__restrict__ cnt256) {
  const float fNn = (float)Nn;
  if (!(p.x >= 0.f && p.x < fNn && p.y >= 0.f && p.y < fNn && p.z >= 0.f && p.z < fNn)) return;
  int tx = 0, ty = 0, tz = 0;
  if (G.T > 1) { const int nct = G.pt / ms; tx = ((int)floorf(p.x / (float)ms)) / nct; ty = ((int)floorf(p.y / (float)ms)) / nct; tz = ((int)floorf(p.z / (float)ms)) / nct; }
  const int tile = (tz * G.T + ty) * G.T + tx, lo = G.nb - 2, fb = G.fb;
  const float offx = (float)G.nb - (float)(tx * G.pt), offy = (float)G.nb - (float)(ty * G.pt), offz = (float)G.nb - (float)(tz * G.pt);  // :227
  const int i1 = (int)floorf(p.x + offx) - lo, j1 = (int)floorf(p.y + offy) - lo, k1 = (int)floorf(p.z + offz) - lo;                     // :248
  const int js = (int)floorf(p.y) + G.nb - ty * G.pt - lo, ks = (int)floorf(p.z) + G.nb - tz * G.pt - lo;                                   // the row the record is sorted into
  const int bs = (tile * fb + ks) * fb + js, br = (tile * fb + k1) * fb + j1;
  if (bs / nr == br / nr) return;                                                                                                        // kicked by the fused pass
  const int64_t o = (int64_t)br * G.fbp + i1;
  const float fx = fbox[o], fy = fbox[o + comp_stride], fz = fbox[o + 2 * comp_stride];
  const int vi = rec_index(p);
  float4 v = vel[vi];
  v.x = v.x + fx * a_mid * P3M_G_F * dt;                                                             // :265-266
  v.y = v.y + fy * a_mid * P3M_G_F * dt;
  v.z = v.z + fz * a_mid * P3M_G_F * dt;
  if (COARSE) {
    const float inv = 1.0f / (float)ms;
    const float cx_ = inv * p.x - 0.5f, cy_ = inv * p.y - 0.5f, cz_ = inv * p.z - 0.5f;            // coarse_velocity.f90:143
    const int ci = (int)floorf(cx_) + 1, cj = (int)floorf(cy_) + 1, ck = (int)floorf(cz_) + 1;
    const float dx1 = (float)ci - cx_, dy1 = (float)cj - cy_, dz1 = (float)ck - cz_;
    const float dx2 = 1.0f - dx1, dy2 = 1.0f - dy1, dz2 = 1.0f - dz1;
    const int m = ncn + 2; const int64_t ccs = (int64_t)m * m * m;
#pragma unroll
    for (int cz = 0; cz < 2; cz++)
#pragma unroll
      for (int cy = 0; cy < 2; cy++)
#pragma unroll
        for (int cx = 0; cx < 2; cx++) {                                                              // :153-168
          const float dV = a_mid * P3M_G_F * dt * (cx ? dx2 : dx1) * (cy ? dy2 : dy1) * (cz ? dz2 : dz1);
          const int64_t oc = ((int64_t)(ck + cz) * m + (cj + cy)) * m + (ci + cx);
          v.x = v.x + fc[oc] * dV; v.y = v.y + fc[oc + ccs] * dV; v.z = v.z + fc[oc + 2 * ccs] * dV;
        }
  }
  vel[vi] = v;
  if (cnt256) atomicAdd(&cnt256[s >> 8], 1);
  rowflag[br] = 0;   // every record that needed the row clears it: the flags are all zero again when the step ends
}
template <bool COARSE>
__global__ __launch_bounds__(256) void k_kick_fix(const float4 *__restrict__ spos, int nrec, const int *__restrict__ cand, const int *__restrict__ cand_cnt, int cand_seg, TileGeo G,
                                                  int Nn, int ms, int nr, unsigned char *__restrict__ rowflag, const float *__restrict__ fbox, int64_t comp_stride, float a_mid, float dt,
                                                  float4 *__restrict__ vel, const float *__restrict__ fc, int ncn, int *__restrict__ cnt256) {
  if (cand_cnt[16 * P3M_CAND_SLOTS] != 0) {   // a candidate list overflowed: the same grid looks at every record
    const float thr = 1.0f - 0.0009765625f;
    const int64_t nthr = (int64_t)gridDim.x * gridDim.y * 256;
    for (int64_t s = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; s < nrec; s += nthr) {
      const float4 p = spos[s];
      if ((p.y - floorf(p.y) >= thr) || (p.z - floorf(p.z) >= thr)) kick_fix_record<COARSE>(p, (int)s, G, Nn, ms, nr, rowflag, fbox, comp_stride, a_mid, dt, vel, fc, ncn, cnt256);
    }
    return;
  }
  const int slot = blockIdx.y, ncand = min(cand_cnt[slot * 16], cand_seg);
  const int *list = cand + (int64_t)slot * cand_seg;
  for (int ci = blockIdx.x * 256 + threadIdx.x; ci < ncand; ci += gridDim.x * 256) {
    const int s = list[ci];
    kick_fix_record<COARSE>(spos[s], s, G, Nn, ms, nr, rowflag, fbox, comp_stride, a_mid, dt, vel, fc, ncn, cnt256);
  }
}

// Whole NGP steps run the fused pass when the tile size has a two-register-stage x kernel whose staging buffer holds the box rows, all
// tiles are swept in one batch (the LY rows of every tile must still exist when the kick runs) and the coarse force array can be
// indexed with 32-bit byte offsets.  P3M_KICK_UNFUSED=1 keeps the force box + k_fine_kick_rows pair (A/B measurements, tests).
bool fine_kick_fusable(const p3m_ctx *c) {
  const char *env = getenv("P3M_KICK_UNFUSED");   // (read at every step: a test runs both paths in one process)
  const bool off = env && env[0] == '1';
  const Geometry &g = c->g;
  const int64_t m = g.ncn + 2;
  return !off && (c->p.flags & P3M_FLAG_NGP) && c->tile_batch == g.ntiles && c->fuse_nr > 0 && c->rowflag && fft_x2_box_pass(g.nf, g.nb - 2) &&
         3 * m * m * m < 0x3fffffffLL && (int64_t)g.ntiles * g.fb * g.fb < 0x7fffffffLL;
}
static int fine_xinv_kick_fused(p3m_ctx *c, float a_mid, float dt, int *cnt256, bool dry = false) {
  const Geometry &g = c->g;
  const int64_t cs = (int64_t)g.ntiles * g.fb * g.fb * g.fbp;
  KickFuseArgs a{};
  a.src = reinterpret_cast<const float2 *>(c->work); a.tw_g = c->plan_f.d_tw; a.inv_scale = (float)g.nf * (float)g.nf * (float)g.nf; a.n = g.nf; a.px = g.px;
  a.fb = g.fb; a.fbp = g.fbp; a.lo = g.nb - 2; a.ntile = g.ntiles; a.rows_total = g.ntiles * g.fb * g.fb;
  a.T = g.T; a.pt = g.pt; a.E = g.E; a.nb = g.nb; a.Nn = g.Nn; a.ms = g.ms; a.ncn = g.ncn;
  a.box = c->fbox; a.bcs = cs; a.rowflag = c->rowflag;
  a.spos = c->spos; a.vel = c->vel; a.cs = c->cell_end; a.crow = c->cells_compact ? c->crow : nullptr; a.crow_w = c->crow_w;
  a.a_mid = a_mid; a.dt = dt; a.fmax_out = c->d_red; a.fc = c->coarse_first ? c->force_c : nullptr; a.cnt256 = cnt256; a.dry = dry ? 1 : 0;
  if (dry) a.fmax_out = c->d_red + 7 * P3M_RED_SPAN;   // scratch slot: the step's maximum stays what it is
  P3M_TRY(kick_fused_launch(c, a, c->coarse_first));
  if (c->np_all > 0 && !dry) {
    TileGeo G{g.T, g.nf, g.nb, g.pt, g.E, g.fb, 2 * g.px, g.fbp};
    const int64_t guess = std::max<int64_t>(1, (int64_t)c->np_all / 256 / P3M_CAND_SLOTS);
    const dim3 gl((unsigned)std::min<int64_t>(64, cdiv(guess, 256)), P3M_CAND_SLOTS);
#define P3M_FIX(COv) hipLaunchKernelGGL((k_kick_fix<COv>), gl, dim3(256), 0, c->stream, (const float4 *)c->spos, c->np_all, (const int *)c->cand, (const int *)c->cand_cnt, \
      c->cand_seg, G, g.Nn, g.ms, c->fuse_nr, c->rowflag, (const float *)c->fbox, cs, a_mid, dt, c->vel, (const float *)a.fc, g.ncn, cnt256)
    if (c->coarse_first) P3M_FIX(true); else P3M_FIX(false);
#undef P3M_FIX
    HIP_TRY(hipGetLastError());
  }
  return P3M_OK;
}

// timing hook (p3m_hip_time_fft_pass, which = 7): the fused pass as the step runs it, over whatever LY rows `work` holds, without the
// velocity stores and the fix-up (call after a whole step: sorted records, row tables, coarse force)
int fine_time_fused_kick(p3m_ctx *c) {
  if (!fine_kick_fusable(c) || c->np_all == 0) { p3m_set_error("the fused inverse-x + kick pass is not what this context runs (CIC, tile size, P3M_KICK_UNFUSED) or no sorted records"); return P3M_ESTATE; }
  const bool cf = c->coarse_first;
  c->coarse_first = coarse_kick_rides_on_fine(c);
  const int r = fine_xinv_kick_fused(c, 0.5f, 0.0f, nullptr, true);
  c->coarse_first = cf;
  return r;
}

// CIC fine mesh: the maximum and the kick in one pass over the force box (k_fine_kick_cic)
static int fine_max_and_kick_cic(p3m_ctx *c, float a_mid, float dt, bool count_survivors_reset) {
  const Geometry &g = c->g;
  TileGeo G{g.T, g.nf, g.nb, g.pt, g.E, g.fb, 2 * g.px, g.fbp};
  const int64_t cs = (int64_t)g.ntiles * g.fb * g.fb * g.fbp;
  P3M_TRY(particles_full_cells(c));
  if (count_survivors_reset) c->cnt_from_kick = 0;   // delete_particles counts its survivors itself
  const int nxs = cdiv(g.fb, CK_XS), xsl = (cdiv(g.fb, nxs) + 3) & ~3, nbj = cdiv(g.fb, CK_BJ), nbk = cdiv(g.fb, CK_BK);   // x segments of equal length
  const int64_t nblk = (int64_t)g.ntiles * nbk * nbj * nxs;
  if (nblk * std::max(nxs, std::max(nbj, nbk)) >= 0xffffffffLL) { p3m_set_error("CIC kick: too many force-box blocks"); return P3M_EINVAL; }
  static int occ[2] = {0, 0};
  const int ci = c->coarse_first ? 1 : 0;
  if (occ[ci] == 0) {
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[ci], ci ? reinterpret_cast<const void *>(k_fine_kick_cic<true>) : reinterpret_cast<const void *>(k_fine_kick_cic<false>), 256, 0));
    if (occ[ci] < 1) occ[ci] = 1;
  }
  const unsigned grid = (unsigned)std::min<int64_t>((int64_t)256 * occ[ci], 8 * cdiv(nblk, 8));   // persistent workgroups, a multiple of the eight XCDs
  if (c->coarse_first)
    hipLaunchKernelGGL(k_fine_kick_cic<true>, dim3(grid), dim3(256), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, g.Nn, g.ms,
                       (const float *)c->fbox, cs, a_mid, dt, c->d_red, (const float *)c->force_c, g.ncn, nxs, nbj, nbk, (int)nblk, xsl);
  else
    hipLaunchKernelGGL(k_fine_kick_cic<false>, dim3(grid), dim3(256), 0, c->stream, (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, g.Nn, g.ms,
                       (const float *)c->fbox, cs, a_mid, dt, c->d_red, (const float *)nullptr, g.ncn, nxs, nbj, nbk, (int)nblk, xsl);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}
// :208-319 for every tile: maximum and kick; fused into one pass over the force box for NGP
int fine_max_and_kick(p3m_ctx *c, float a_mid, float dt, bool count_survivors) {
  const Geometry &g = c->g;
  if (!(c->p.flags & P3M_FLAG_NGP)) return fine_max_and_kick_cic(c, a_mid, dt, count_survivors);
  TileGeo G{g.T, g.nf, g.nb, g.pt, g.E, g.fb, 2 * g.px, g.fbp};
  const int64_t cs = (int64_t)g.ntiles * g.fb * g.fb * g.fbp;
  // the kick visits every physical record once: it counts them per block of 256 sorted records for delete_particles
  // (particles_finalize_enqueue), which then needs no pass of its own over the positions -- unless the grid moves back first
  int *cnt256 = nullptr;
  if (count_survivors) c->cnt_from_kick = 0;
  if (count_survivors && !(c->p.flags & P3M_FLAG_MOVE_GRID_BACK) && c->np_all > 0) {
    cnt256 = c->flags;
    if (!c->step_zeroed) HIP_TRY(hipMemsetAsync(cnt256, 0, sizeof(int) * (size_t)(cdiv(c->np_all, 256) + 1), c->stream));   // (whole steps: step_prezero)
    c->cnt_from_kick = c->np_all;
  }
  if (c->xinv_deferred) { c->xinv_deferred = false; return fine_xinv_kick_fused(c, a_mid, dt, cnt256); }   // the force phase stopped after the inverse y pass
  if (c->coarse_first)
    hipLaunchKernelGGL(k_fine_kick_rows<true>, dim3((unsigned)cdiv((int64_t)g.ntiles * g.fb * g.fb, P3M_KICK_WPB)), dim3(64 * P3M_KICK_WPB), sizeof(float) * 3 * g.fbp * P3M_KICK_WPB, c->stream,
                       (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, g.Nn, g.ms, (const float *)c->fbox, cs, a_mid, dt, c->d_red,
                       (const float *)c->force_c, g.ncn, c->cells_compact ? (const int *)c->crow : (const int *)nullptr, c->crow_w, cnt256);
  else
    hipLaunchKernelGGL(k_fine_kick_rows<false>, dim3((unsigned)cdiv((int64_t)g.ntiles * g.fb * g.fb, P3M_KICK_WPB)), dim3(64 * P3M_KICK_WPB), sizeof(float) * 3 * g.fbp * P3M_KICK_WPB, c->stream,
                       (const float4 *)c->spos, c->vel, (const int *)c->cell_end, G, g.Nn, g.ms, (const float *)c->fbox, cs, a_mid, dt, c->d_red,
                       (const float *)nullptr, g.ncn, c->cells_compact ? (const int *)c->crow : (const int *)nullptr, c->crow_w, cnt256);
  HIP_TRY(hipGetLastError());
  return P3M_OK;
}

// ------------------------------------------------------------------ fine_kernel (kernel_initialization.f90:2-267)
// real-space table mirrored into the nf^3 box (odd along the component's own axis, :69-85),
// PP_EXT corner zeroed (:38-54); then r2c and keep the imaginary part (:93-99).
__global__ __launch_bounds__(256) void k_fine_kernel_real(float *__restrict__ rho, const float *__restrict__ table, int nf, int rp, int ncut, int comp,
                                                          int zero_corner) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t tot = (int64_t)nf * nf * rp;
  if (idx >= tot) return;
  const int i = (int)(idx % rp); const int64_t r = idx / rp; const int j = (int)(r % nf), k = (int)(r / nf);
  float v = 0.f;
  if (i < nf) {
    int c3[3] = {i, j, k}, t3[3]; float sgn = 1.f; bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      if (c3[d] < ncut) t3[d] = c3[d];
      else if (c3[d] > nf - ncut) { t3[d] = nf - c3[d]; if (d == comp) sgn = -sgn; }
      else ok = false;
    }
    if (ok) {
      if (!(zero_corner > 0 && t3[0] < zero_corner && t3[1] < zero_corner && t3[2] < zero_corner))
        v = sgn * table[(((int64_t)t3[2] * ncut + t3[1]) * ncut + t3[0]) * 3 + comp];
    }
  }
  rho[idx] = v;
}
__global__ __launch_bounds__(256) void k_take_imag(const float *__restrict__ hat, float *__restrict__ kern, int64_t ncomplex) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < ncomplex) kern[i] = hat[2 * i + 1];
}

// K(c, n - z) := +-K(c, z) for z > n/2 in the bundle layout LZ ([bundle][z][16 columns]): the real-space table is mirrored exactly
// (k_fine_kernel_real), its transform only to rounding; made exact, the fused z pass may read the upper half of a line from the lower
// (LinesArgs::kmirror) and every other reader sees the same numbers.  Component 2 is odd along z: its Nyquist plane is zero.
__global__ __launch_bounds__(256) void k_kf_symmetrise_z(float *__restrict__ kern, int n, int64_t nbundles, int odd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int hz = n / 2;   // planes hz .. n-1 are written (hz itself only for the odd component)
  if (i >= nbundles * (n - hz) * BXC) return;
  const int col = (int)(i % BXC); const int64_t t = i / BXC; const int z = hz + (int)(t % (n - hz)); const int64_t b = t / (n - hz);
  float *line = kern + b * n * BXC + col;
  if (z == hz) { if (odd && 2 * hz == n) line[(int64_t)z * BXC] = 0.f; return; }
  const float v = line[(int64_t)(n - z) * BXC];
  line[(int64_t)z * BXC] = odd ? -v : v;
}
int build_fine_kernel(p3m_ctx *c, const float *table16_host) {
  const Geometry &g = c->g;
  float *d_table = nullptr;
  const size_t tb = sizeof(float) * 3 * g.ncut * g.ncut * g.ncut;
  HIP_TRY(hipMalloc(&d_table, tb));
  HIP_TRY(hipMemcpyAsync(d_table, table16_host, tb, hipMemcpyHostToDevice, c->stream));
  const int64_t tot = (int64_t)g.nf * g.nf * (2 * g.px), ncx = (int64_t)g.nf * g.nf * g.px;
  const int zc = (c->p.flags & P3M_FLAG_PP_EXT) ? g.pp_range + 1 : 0;
  for (int comp = 0; comp < 3; comp++) {
    hipLaunchKernelGGL(k_fine_kernel_real, dim3(cdiv(tot, 256)), dim3(256), 0, c->stream, c->rho, (const float *)d_table, g.nf, 2 * g.px, g.ncut, comp, zc);
    HIP_TRY(hipGetLastError());
    P3M_TRY(fft3d_forward(c, c->plan_f, c->rho, c->work, 1));   // rho-hat in the bundle layout LZ; so is kern_f
    hipLaunchKernelGGL(k_take_imag, dim3(cdiv(ncx, 256)), dim3(256), 0, c->stream, (const float *)c->rho, c->kern_f + comp * ncx, ncx);
    HIP_TRY(hipGetLastError());
    const int64_t nbun = (int64_t)g.nf * (g.px / BXC), nsym = nbun * (g.nf - g.nf / 2) * BXC;
    hipLaunchKernelGGL(k_kf_symmetrise_z, dim3(cdiv(nsym, 256)), dim3(256), 0, c->stream, c->kern_f + comp * ncx, g.nf, nbun, comp == 2 ? 1 : 0);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  (void)hipFree(d_table);
  c->have_kf = true; c->kf_zmirror = true;
  return P3M_OK;
}
