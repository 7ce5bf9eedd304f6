// kick_fused.h -- arguments of k_fft_x_inv2_kick (kick_fused.hip): the inverse x pass of the three force components with the NGP
// kick of particle_mesh_threaded.f90:208-270 (and coarse_velocity.f90:137-179) applied from LDS instead of from a force box
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

struct KickFuseArgs {
  // the x pass (k_fft_x_inv2): LY source of the three components [comp][tile], twiddles, n^3
  const float2 *src; const float2 *tw_g; float inv_scale; int n, px;
  // force-box geometry: rows_total = ntile * fb * fb box rows, each with three components
  int fb, fbp, lo, ntile, rows_total, FP;   // FP: row pitch of the rows in LDS (set by the launcher)
  int T, pt, E, nb, Nn, ms, ncn;
  // flagged rows (k_ngp_fixup) are ALSO stored to the box for k_kick_fix
  float *box; int64_t bcs; const unsigned char *rowflag;
  // records: sorted positions, velocities in arrival order, row ranges from cell_end (cs) or the compact row table (crow)
  const float4 *spos; float4 *vel; const int *cs; const int *crow; int crow_w;
  float a_mid, dt; float *fmax_out; const float *fc; int *cnt256;
  unsigned m_fb;   // division magic of fb (set by the launcher)
  int dry;         // timing hook (p3m_hip_time_fft_pass 7): everything but the velocity stores
};

struct p3m_ctx;
int kick_fused_rows(int n, int fbp, int lo);                       // box rows per batch of the fused pass for line length n, box pitch fbp, box offset lo (0: none)
int kick_fused_launch(p3m_ctx *c, KickFuseArgs &a, bool coarse);   // coarse: the coarse kick rides along (a.fc)
