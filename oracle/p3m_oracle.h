/*
 * p3m_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32 like the reference's real(4)) of cubep3m's
 * `particle_mesh` path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported baseline.
 * The product (cubep3m_amd/libp3m_hip.so) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - FFT-free stages (update_position, link_list, particle_pass, fine NGP/CIC deposit,
 *     coarse CIC deposit, coarse_force_buffer, coarse_max_dt, coarse_velocity,
 *     delete_particles, move_grid_back) are checked against the reference's own object
 *     code built where it lies by oracle/Makefile -> oracle/_ref/ (tests/test_oracle_vs_ref.py)
 *     and against the golden fixtures generated from it (tests/golden/).
 *   - The FFT is FFTW 2.1.5 single precision (Makefile_gnu_sfftw2:5), a third-party
 *     library absent from /root/reference and from this image; it is restated here as the
 *     textbook unnormalised DFT (r2c sign -1, c2r sign +1, half-complex on the fastest
 *     axis, fftw2.f90:19-22) and pinned against numpy.fft and closed forms.
 *   - Stages inlined in particle_mesh_threaded.f90 between FFT calls cannot be executed
 *     from the reference without FFTW; they are pinned by the known-answer values the
 *     survey recorded from a run of the reference and by analytic
 *     tests (Newtonian pair force inside the PP range, momentum conservation).
 *
 * Multi-rank: all nodes_dim^3 ranks of the reference are simulated inside ONE process;
 * MPI calls become memcpy between the per-rank states.
 */
#ifndef P3M_ORACLE_H
#define P3M_ORACLE_H
#include "../include/p3m_hip.h" /* p3m_params, p3m_step_out, P3M_FLAG_* (struct layouts only) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

orc_ctx *orc_create(const p3m_params *p);
void orc_destroy(orc_ctx *c);
int64_t orc_derived(const orc_ctx *c, int what); /* same selector as p3m_hip_derived */

/* kernel_initialization.f90: fine_kernel (:2-267), coarse_kernel (:272-732) */
void orc_fine_kernel(orc_ctx *c, const float *table16);
void orc_coarse_kernel(orc_ctx *c, const float *table4);
const float *orc_kern_f(orc_ctx *c);            /* (3, nf/2+1, nf, nf)             */
const float *orc_kern_c(orc_ctx *c);            /* (3, nc/2+1, nc, nc) all slabs   */

/* particle store of one simulated rank */
int orc_set_particles(orc_ctx *c, int rank, const float *xv6, const int64_t *pid, int np);
int orc_get_np(orc_ctx *c, int rank);
void orc_get_particles(orc_ctx *c, int rank, float *xv6, int64_t *pid);

/* phases, each acting on ALL simulated ranks, in the order particle_mesh calls them */
void orc_update_position(orc_ctx *c, float dt, float dt_old, const float *offset);
void orc_link_list(orc_ctx *c);
int orc_particle_pass(orc_ctx *c);
void orc_fine_mesh(orc_ctx *c, float a_mid, float dt, float mass_p);
void orc_coarse_mesh(orc_ctx *c, float a_mid, float dt, float mass_p);
void orc_move_grid_back(orc_ctx *c, const float *shake_offset);
void orc_delete_particles(orc_ctx *c);
void orc_step_out(orc_ctx *c, float a_mid, p3m_step_out *out);

/* subroutine particle_mesh (particle_mesh_threaded.f90:2-726) */
int orc_particle_mesh(orc_ctx *c, float a_mid, float dt, float dt_old, float mass_p,
                      const float *offset, const float *move_back, p3m_step_out *out);

/* probes */
/* projection.f90: the three density projections (global nf_physical_dim^2 each, reference memory order) and the projected mass; needs link_list + particle_pass */
void orc_projection(orc_ctx *c, float mass_p, float *pxy, float *pxz, float *pyz, double *rho_tot);
void orc_tile_density(orc_ctx *c, int rank, int tx, int ty, int tz, float mass_p, float *rho_f);
void orc_tile_force(orc_ctx *c, const float *rho_f, float *force_f, float *force_max2);
/* gather + kick + intra-cell PP of one tile on a caller-supplied force box (3,(pt+3)^3 component fastest);
   out[0] = max |F|^2 over the box (:208-223), out[1] = pp_force_max (:355-358) */
void orc_tile_velocity(orc_ctx *c, int rank, int tx, int ty, int tz, const float *force_f, float a_mid, float dt, float mass_p, float *out);
void orc_coarse_density(orc_ctx *c, float mass_p);           /* coarse_mass on all ranks */
const float *orc_rho_c(orc_ctx *c, int rank);                /* (ncn,ncn,ncn)            */
const float *orc_force_c(orc_ctx *c, int rank);              /* (3,0:ncn+1,0:ncn+1,0:ncn+1) */
void orc_coarse_force(orc_ctx *c);
/* coarse_power.f90: ps (2, nc_dim) = (k, Delta^2(k)) rows of <z>ps.dat; needs coarse_density */
void orc_coarse_power(orc_ctx *c, float mass_p, float box, float *ps);                           /* coarse_force + buffer     */
void orc_distribute_force(orc_ctx *c, const float *force_global); /* coarse_force_buffer.f90 */
void orc_coarse_max_dt_and_velocity(orc_ctx *c, float a_mid, float dt); /* coarse_max_dt + coarse_velocity */
void orc_fft3d(float *data, int n, int dir);                 /* fftw2.f90 semantics       */
void orc_fft3d_rect(float *data, int nx, int ny, int nz, int dir);

/* host time loop (timestep.f90) */
void orc_expansion(const p3m_time_params *P, float a0, float dt0, float *da1, float *da2);
void orc_timestep(const p3m_time_params *P, unsigned flags, p3m_time_state *S, float dt_f_acc, float dt_pp_acc, float dt_pp_ext_acc,
                  float dt_c_acc);

#ifdef __cplusplus
}
#endif
#endif
