// timestep.hip -- host time loop of the reference (timestep.f90): HOST code only, no device work.
// The types follow the Fortran source operation by operation (real(4) state, real(8) inside expansion with
// real(4) literals and parameters promoted where Fortran promotes them) so that a multi-step run chooses the
// same steps as the reference does.
#include "p3m_internal.h"
#include <cmath>

// subroutine expansion (timestep.f90:218-293)
static float half_step(const p3m_time_params *P, double a_x, float dt_x) {
  const double omHsq = (double)(4.0f / 9.0f);                                      // :239 (real(4) constant expression)
  const double a3rlm = pow(a_x, (double)(-(3 * P->wde))) * P->omega_l / P->omega_m;  // :241
  const double arkm = a_x * ((1.0f - P->omega_m) - P->omega_l) / P->omega_m;       // :243
  const double adot = sqrt(omHsq * ((a_x * a_x) * a_x) * ((1.0 + arkm) + a3rlm));  // :245
  const double addot = (a_x * a_x) * omHsq * ((1.5 + 2.0 * arkm) + (double)(1.5f * (1.0f - P->wde)) * a3rlm);                          // :247
  const double atdot = a_x * adot * omHsq * ((3.0 + 6.0 * arkm) + (double)((1.5f * (2.0f - 3.0f * P->wde)) * (1.0f - P->wde)) * a3rlm);  // :250
  const float dt2 = dt_x * dt_x, dt3 = dt2 * dt_x;                                 // dt_x**2, dt_x**3 in real(4)
  return (float)((adot * dt_x + (addot * dt2) / 2.0) + (atdot * dt3) / 6.0);       // :253
}
// subroutine Chaplygin (timestep.f90:296-339).  a3rchm, arkm and G_ch are formed ONCE, at a_x = a0 (:310-312): the second half
// step (:322-329) changes a_x only where it stands explicitly in adot / addot / atdot -- the reference's text, kept
static void chaplygin(const p3m_time_params *P, float a0, float dt0, float *da1, float *da2) {
  const float dt_x = dt0 / 2;                                                          // :306
  double a_x = (double)a0;
  const double omHsq = (double)(4.0f / 9.0f);                                          // :308
  const double a3rchm = pow(a_x, -3.0) * P->omega_ch / P->omega_m;                     // :310  a_x**(-3): integer power of a real(8)
  const double arkm = a_x * ((1.0f - P->omega_m) - P->omega_ch) / P->omega_m;          // :311
  const float e1 = -3.0f - 3.0f * P->alpha_ch, e2 = -6.0f - 6.0f * P->alpha_ch;        // real(4) exponents
  const double G_ch = (double)P->A_ch + (double)(1.0f - P->A_ch) * pow(a_x, (double)e1);   // :312
  const float p1 = 1.0f / (1.0f + P->alpha_ch), p2 = -P->alpha_ch / (1.0f + P->alpha_ch), p3 = 1.0f / (1.0f + P->alpha_ch) - 2.0f;
  const float c5 = 5.0f * (P->A_ch * P->A_ch), c3 = (3.0f * P->A_ch) * (1.0f - P->A_ch), ch = 2.0f + P->alpha_ch / 2.0f, c1 = (1.0f - P->A_ch) * (1.0f - P->A_ch);
  auto half = [&](double ax) {
    const double adot = sqrt(omHsq * ((ax * ax) * ax) * ((1.0 + arkm) + a3rchm * pow(G_ch, (double)p1)));                                 // :314
    const double addot = (ax * ax) * omHsq * ((1.5 + 2.0 * arkm) + ((3.0 * a3rchm) * P->A_ch) * pow(G_ch, (double)p2));                   // :315
    const double poly = ((double)c5 + ((double)c3 * pow(ax, (double)e1)) * (double)ch) + (double)c1 * pow(ax, (double)e2);
    const double atdot = ax * adot * omHsq * ((3.0 + 6.0 * arkm) + ((3.0 * a3rchm) * pow(G_ch, (double)p3)) * poly);                      // :316
    const float dt2 = dt_x * dt_x, dt3 = dt2 * dt_x;
    return (float)((adot * dt_x + (addot * dt2) / 2.0) + (atdot * dt3) / 6.0);                                                            // :319
  };
  *da1 = half(a_x);
  a_x = (double)(a0 + *da1);                                                           // :321
  *da2 = half(a_x);
}
extern "C" void p3m_hip_expansion(const p3m_time_params *P, float a0, float dt0, float *da1, float *da2) {
  if (P->chaplygin) { chaplygin(P, a0, dt0, da1, da2); return; }                       // :251-252
  const float dt_x = dt0 / 2;                          // :236
  *da1 = half_step(P, (double)a0, dt_x);               // :237-253
  *da2 = half_step(P, (double)(a0 + *da1), dt_x);      // :255 a_x = a0 + da1 (real(4) sum)
}

static float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }

// subroutine timestep (timestep.f90:2-216), the rank == 0 branch; the mpi_bcast pairs are the caller's business
extern "C" int p3m_hip_timestep(const p3m_time_params *P, uint32_t flags, p3m_time_state *S, float dt_f_acc, float dt_pp_acc,
                                float dt_pp_ext_acc, float dt_c_acc) {
  if (!P || !S) return P3M_EINVAL;
  if (S->cur_checkpoint < 1 || S->cur_checkpoint > P3M_MAX_INPUT || S->cur_projection < 1 || S->cur_projection > P3M_MAX_INPUT ||
      S->cur_halofind < 1 || S->cur_halofind > P3M_MAX_INPUT) { p3m_set_error("timestep: cur_* out of 1..%d", P3M_MAX_INPUT); return P3M_EINVAL; }
  S->nts = S->nts + 1;                                  // :20
  if (S->nts != 1) S->dt_old = S->dt;                   // :21
  const bool ppint = (flags & P3M_FLAG_PPINT) != 0, ppext = ppint && (flags & P3M_FLAG_PP_EXT) != 0;
  if (P->cosmo) {
    float da_1, da_2, ra;
    float dt_e = P->dt_max;                             // :59
    for (int n = 1;; n++) {                             // :63-74 restrict expansion
      p3m_hip_expansion(P, S->a, dt_e, &da_1, &da_2);
      S->da = da_1 + da_2;
      ra = S->da / (S->a + S->da);
      if (ra > P->ra_max) dt_e = dt_e * (P->ra_max / ra); else break;
      if (n > 10) break;
    }
    if (P->restrict_da) {                               // :76-88
      for (int n = 0;;) {
        p3m_hip_expansion(P, S->a, dt_e, &da_1, &da_2);
        S->da = da_1 + da_2;
        if (S->da > P->da_max) dt_e = dt_e * (P->da_max / S->da); else break;
        n++;
        if (n > 10) break;
      }
    }
    float dt = fminf(dt_e, dt_f_acc);                   // :93-115
    if (ppint) dt = fminf(dt, dt_pp_acc);
    if (ppext) dt = fminf(dt, dt_pp_ext_acc);
    dt = fminf(dt, dt_c_acc);
    dt = dt * P->dt_scale;                              // :117
    p3m_hip_expansion(P, S->a, dt, &da_1, &da_2);       // :119
    S->da = da_1 + da_2;
    S->checkpoint_step = S->projection_step = S->halofind_step = 0;   // :125-127
    const float ac = P->a_checkpoint[S->cur_checkpoint - 1], ap = P->a_projection[S->cur_projection - 1], ah = P->a_halofind[S->cur_halofind - 1];
    const float am = min3(ac, ap, ah);                  // :130
    const bool prj_ok = S->cur_projection <= P->num_projections, hf_ok = S->cur_halofind <= P->num_halofinds;
    if (ac == am) {                                     // :135-142
      if (S->a + S->da > ac) {
        S->checkpoint_step = 1;
        dt = dt * (ac - S->a) / S->da;
        p3m_hip_expansion(P, S->a, dt, &da_1, &da_2);
        if (S->cur_checkpoint == P->num_checkpoints) S->final_step = 1;
        if (ap == am && prj_ok) S->projection_step = 1;
        if (ah == am && hf_ok) S->halofind_step = 1;
      }
    } else if (ap == am && prj_ok) {                    // :144-151
      if (S->a + S->da > ap) {
        S->projection_step = 1;
        dt = dt * (ap - S->a) / S->da;
        p3m_hip_expansion(P, S->a, dt, &da_1, &da_2);
        if (ah == am && hf_ok) S->halofind_step = 1;
      }
    } else if (ah == am && hf_ok) {                     // :153-159
      if (S->a + S->da > ah) {
        S->halofind_step = 1;
        dt = dt * (ah - S->a) / S->da;
        p3m_hip_expansion(P, S->a, dt, &da_1, &da_2);
      }
    }
    S->dt = dt;
    S->dt_gas = dt / 4;                                 // :165
    S->da = da_1 + da_2;
    S->a_mid = S->a + (S->da / 2);                      // :168
    S->tau = S->tau + dt; S->t = S->t + dt; S->a = S->a + S->da;   // :193-195
  } else {                                              // :197-216
    S->a = 1.0f; S->a_mid = S->a; S->da = 0.0f;
    float dt;
    if (ppint && P->pair_infall) {                      // :204-206: min(0.05/sqrt(G*mass_p/cur_sep**2), dt_f_acc, dt_pp_acc, dt_c_acc)
      dt = fminf(0.05f / sqrtf(P3M_G_F * P->mass_p / (P->cur_sep * P->cur_sep)), dt_f_acc);
      dt = fminf(dt, dt_pp_acc);
      dt = fminf(dt, dt_c_acc);
    } else {                                            // :208-214
      dt = fminf(1.0f, dt_f_acc);
      if (ppint) dt = fminf(dt, dt_pp_acc);
      if (ppext) dt = fminf(dt, dt_pp_ext_acc);
      dt = fminf(dt, dt_c_acc);
    }
    if (P->pairwise_ic) dt = 1.0f;                      // :210
    if (P->shake_test_ic) dt = 1.0f;                    // :211
    S->dt = dt;
    S->t = S->t + dt;                                   // :212
  }
  return P3M_OK;
}
