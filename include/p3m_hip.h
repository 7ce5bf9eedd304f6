/*
 * p3m_hip.h -- C ABI of the MI355X-native P3M gravity step that drops in behind
 * cubep3m's `subroutine particle_mesh`
 * (reference: source_threads/particle_mesh_threaded.f90:2, callers cubepm.f90:143,
 * report_force.f90:41,100).
 *
 * The reference has no plugin API: `particle_mesh` takes no arguments and exchanges
 * all state through COMMON blocks (source_threads/cubep3m.fh:147-171).  Its only FFI
 * precedent is the F77-style `pp_force_c_` (source_threads/nbody-ueli.cu:368, called at
 * particle_mesh_cuda.f90:578): lower-case name + trailing underscore, every argument by
 * reference, caller-owned arrays.  This header therefore offers
 *   (1) a context API with plain pointers and sizes (what an ISO_C_BINDING adapter or
 *       ctypes binds), and
 *   (2) `particle_mesh_hip_` -- an F77-ABI one-call wrapper in the style of
 *       `pp_force_c_` for hosts that want the smallest possible change.
 * Every entry point cites the reference routine it replaces.
 *
 * Conventions shared with the reference:
 *   - all reals are fp32 (`real(4)`), indices int32, particle ids int64 (cubep3m.fh:75-79)
 *   - xv is AoS (6,np): x,y,z,vx,vy,vz; positions are in fine-cell units local to the
 *     rank, physical range [0, nf_physical_node_dim)                (cubep3m.fh:75)
 *   - kern_f(3, nf_tile/2+1, nf_tile, nf_tile), kern_c(3, nc_dim/2+1, nc_dim, nc_slab)
 *     with the component index fastest                             (cubep3m.fh:35,56)
 * Errors: the reference calls mpi_abort/stop (particle_pass.f90:96-99,136-139;
 * particle_mesh_threaded.f90:280-283).  This library never aborts the host: every
 * function returns 0 on success or a negative P3M_E* code; p3m_hip_last_error() returns
 * the message.
 */
#ifndef P3M_HIP_H
#define P3M_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compile-time switches of the reference, as run-time flags -------------------- */
#define P3M_FLAG_NGP        (1u << 0) /* -DNGP : fine NGP deposit+gather; else CIC
                                         (particle_mesh_threaded.f90:119-161,260-317)   */
#define P3M_FLAG_PPINT      (1u << 1) /* -DPPINT: intra-fine-cell PP (requires NGP)
                                         (particle_mesh_threaded.f90:274-285,324-361)   */
#define P3M_FLAG_PP_EXT     (1u << 2) /* -DPP_EXT: extended PP over pp_range cells
                                         (particle_mesh_threaded.f90:378-624) and the
                                         zeroed kernel corner (kernel_initialization.f90:38-54) */
#define P3M_FLAG_LRCKCORR   (1u << 3) /* -DLRCKCORR: long-range coarse kernel correction
                                         (kernel_initialization.f90:465-687)             */
#define P3M_FLAG_MOVE_GRID_BACK (1u << 4) /* -DMOVE_GRID_BACK (move_grid_back.f90)       */
#define P3M_FLAG_COARSE_NGP (1u << 6) /* -DCOARSE_NGP: the coarse deposit and the coarse gather put the whole weight on
                                         cell i2 = floor(x/mesh_scale - 0.5) + 2 (coarse_cic_mass.f90:21-24,
                                         coarse_cic_mass_buffer.f90:26-29, coarse_velocity.f90:146-149)              */
#define P3M_FLAG_PENCIL     (1u << 5) /* coarse FFT decomposed in 2-D pencils instead of slabs: the build that links
                                         p3dfft_coarse.f90 (:8-66, pack_pencils :69-127, unpack_pencils :129-183) with
                                         dim_y = nodes_dim, dim_z = nodes_dim^2 (cubepm.par:210-215).  Needs
                                         nc_node_dim % nodes_dim == 0 instead of nc_dim % nodes_dim^3 == 0; groups only */

#define P3M_FLAG_COARSE_ONLY (1u << 7) /* groups only: every logical rank holds its coarse mesh and nothing else (no particle
                                         store, no fine mesh): the stand-alone distributed coarse transform, e.g. a literal
                                         1024^3 slab FFT (fftw3ds.f90:103-183 at nc_dim = 1024, cubepm.par:196-197).  Only
                                         the p3m_hip_group_*coarse* entry points and the kernel tables work on such a group */

/* ---- error codes -------------------------------------------------------------------- */
#define P3M_OK            0
#define P3M_EINVAL       -1  /* bad parameter / unsupported size                         */
#define P3M_ENOMEM       -2  /* device allocation failed                                 */
#define P3M_ECAPACITY    -3  /* particle/ghost capacity exceeded ("exceeded max_np in pass",
                                particle_pass.f90:136-139)                               */
#define P3M_EDEVICE      -4  /* HIP runtime error                                        */
#define P3M_ESTATE       -5  /* call sequence error (e.g. step before kernels are set)   */
#define P3M_ECOMM        -6  /* transport error                                          */

/* Compile-time parameters of the reference (parameters.example, cubepm.par) as a struct. */
typedef struct p3m_params {
  int32_t nodes_dim;       /* ranks per dimension, nodes = nodes_dim^3  (parameters.example:14) */
  int32_t tiles_node_dim;  /* fine tiles per rank per dimension          (parameters.example:17) */
  int32_t nf_tile;         /* fine tile size incl. buffers               (parameters.example:32) */
  int32_t nf_cutoff;       /* fine force cut-off, 16                     (parameters.example:50) */
  int32_t nf_buf;          /* tile buffer, nf_cutoff+8 = 24              (parameters.example:53) */
  int32_t mesh_scale;      /* coarse/fine ratio, 4                       (cubepm.par:157)        */
  int32_t pp_range;        /* extended-PP reach in fine cells, 2         (cubepm.par:92)         */
  int32_t cores;           /* OpenMP threads of the reference; only reproduces the per-thread
                              `pp_ext_force_max` overwrite (particle_mesh_threaded.f90:617)      */
  uint32_t flags;          /* P3M_FLAG_*                                                         */
  float rsoft;             /* PP hard cut, 0.1                           (cubepm.par:76)         */
  float pp_bias;           /* 1.0                                        (cubepm.par:80)         */
  float dt_pp_scale;       /* 0.05                                       (cubepm.par:78)         */
  float density_buffer;    /* capacity factor for max_np                 (parameters.example:24) */
  int32_t rank;            /* this process' rank, 0..nodes-1; x <-> rank%nodes_dim
                              (mpi_initialization.f90:42-76, kernel_initialization.f90:293-298)  */
  int32_t device;          /* HIP device ordinal to use (-1: current device)                     */
} p3m_params;

/* What `particle_mesh` leaves in COMMON for the next `timestep` (cubep3m.fh:20-21) + DIAG sums. */
typedef struct p3m_step_out {
  float dt_f_acc;          /* particle_mesh_threaded.f90:643-656 */
  float dt_pp_acc;         /* :662-673 (1000 if PPINT off, variable_initialization.f90:22-29)   */
  float dt_pp_ext_acc;     /* :685-696                                                          */
  float dt_c_acc;          /* coarse_max_dt.f90:36                                              */
  double sum_rho_f;        /* DIAG "sum of rho_f", particle_mesh_threaded.f90:166-174,702-706   */
  double sum_rho_c;        /* DIAG "sum of rho_c", coarse_mesh.f90:31-43                        */
  int64_t np_total;        /* DIAG "total number of particles", delete_particles.f90:62-64      */
  int32_t np_local;        /* particles left on this rank after delete_particles                 */
  int32_t np_ghost;        /* ghosts this rank held after particle_pass                          */
  int32_t np_deleted;      /* out-of-range particles dropped by link_list (link_list.f90:26-53)  */
  float f_force_max;       /* sqrt(max |F_fine|^2), :643                                         */
  float pp_force_max;      /* :662                                                               */
  float pp_ext_force_max;  /* :685                                                               */
  float c_force_max;       /* coarse_max_dt.f90:24-31                                            */
} p3m_step_out;

/*
 * Host transport for multi-process groups (replaces the MPI calls of particle_pass.f90,
 * fftw3ds.f90:24-39,84-99, coarse_force_buffer.f90:25-63 and the mpi_reduce/bcast pairs) when the
 * processes are NOT connected by RCCL: an MPI Fortran host, or torch.distributed/gloo in the Python
 * host.  All buffers handed to the callbacks are HOST pointers (pinned staging owned by the
 * library).  With RCCL (p3m_hip_group_comm_init_rccl) the callbacks are not used and device
 * buffers go straight over xGMI.  Exchanges between logical ranks of one process never leave the GPU.
 */
typedef struct p3m_transport {
  void *user;
  /* One neighbourhood exchange: for i < npeers send sbytes[i] bytes from sbuf[i] to process peer[i] and
     receive exactly rbytes[i] bytes from it into rbuf[i] (either may be 0).  Post every receive and send,
     then wait for all (MPI_Irecv / MPI_Isend / MPI_Waitall).  All processes call this the same number of
     times in the same order and the sizes agree by construction.  Return 0 on success. */
  int (*exchange)(void *user, int32_t npeers, const int32_t *peer, const void *const *sbuf, const int64_t *sbytes,
                  void *const *rbuf, const int64_t *rbytes);
  /* in-place max / sum over all processes (the mpi_reduce + mpi_bcast pairs) */
  int (*allreduce_max_f32)(void *user, float *v, int32_t n);
  int (*allreduce_sum_f64)(void *user, double *v, int32_t n);
} p3m_transport;

typedef struct p3m_ctx p3m_ctx;

/* -- lifecycle ------------------------------------------------------------------------- */
/* Replaces the static COMMON storage of cubep3m.fh: allocates all device buffers once.   */
int p3m_hip_create(const p3m_params *params, p3m_ctx **out);
void p3m_hip_destroy(p3m_ctx *ctx);
const char *p3m_hip_last_error(void);
int32_t p3m_hip_device_count(void);   /* GPUs visible to this process (an MPI host maps its local rank onto them) */
/* Derived sizes exactly as cubepm.par:170-208 computes them. what: 0 max_np, 1 nc_dim,
   2 nc_node_dim, 3 nf_physical_node_dim, 4 nc_slab, 5 nf_physical_tile_dim */
int64_t p3m_hip_derived(const p3m_ctx *ctx, int32_t what);

/* RCCL over xGMI: unique_id is the 128-byte ncclUniqueId the host broadcast from rank 0
   (p3m_hip_rccl_unique_id fills it on rank 0); handed to p3m_hip_group_comm_init_rccl. */
int p3m_hip_rccl_unique_id(void *unique_id_128);

/* -- Green's functions (kernel_initialization.f90) -------------------------------------- */
/* fine_table: the 16^3 rows of kernels/wfxyzf.3.ascii as float[16][16][16][3] with the file's
   row order (i fastest, then j, then k; 3 components per row) -- fine_kernel :25-36.
   coarse_table: kernels/wfxyzc.2.ascii as float[4][4][4][3] -- coarse_kernel :349-358.
   Builds kern_f / kern_c on the device with the library's own FFT. */
int p3m_hip_set_kernel_tables(p3m_ctx *ctx, const float *fine_table, const float *coarse_table);
/* Alternatively hand over the arrays the reference already computed (COMMON /rvar/ kern_f,
   kern_c) in the reference's own layout. Single-rank contexts only for kern_c. */
int p3m_hip_set_kernels_raw(p3m_ctx *ctx, const float *kern_f, const float *kern_c);
/* Read the device kernels back in the reference layout (tests, kernel_checkpoint.f90). */
int p3m_hip_get_kernels(p3m_ctx *ctx, float *kern_f, float *kern_c);

/* -- particle store (xv, PID, np_local of cubep3m.fh:75-79) ----------------------------- */
int p3m_hip_upload_particles(p3m_ctx *ctx, const float *xv6, const int64_t *pid, int32_t np_local);
/* Order differs from the reference's swap-with-last compaction (delete_particles.f90:17-47):
   match by PID. xv6/pid may be NULL to query np_local only. */
int p3m_hip_download_particles(p3m_ctx *ctx, float *xv6, int64_t *pid, int32_t *np_local);

/* -- the step: subroutine particle_mesh (particle_mesh_threaded.f90:2-726) --------------- */
/* offset: the DISP_MESH shift added in update_position (update_position.f90:56-58,71), or NULL.
   The host keeps the RNG (random_number on rank 0). move_back: with P3M_FLAG_MOVE_GRID_BACK,
   the accumulated shake_offset subtracted after the kick (move_grid_back.f90:17-24), or NULL. */
int p3m_hip_particle_mesh(p3m_ctx *ctx, float a_mid, float dt, float dt_old, float mass_p,
                          const float *offset, const float *move_back, p3m_step_out *out);

/* -- individual phases (same order as particle_mesh calls them); used by tests, the
      benchmark's per-kernel timing and hosts that interleave their own work ------------- */
int p3m_hip_update_position(p3m_ctx *ctx, float dt, float dt_old, const float *offset); /* update_position.f90 */
int p3m_hip_link_list_and_pass(p3m_ctx *ctx);   /* link_list.f90 + particle_pass.f90 (sort replaces hoc/ll) */
int p3m_hip_fine_mesh(p3m_ctx *ctx, float a_mid, float dt, float mass_p);   /* :72-628 */
int p3m_hip_coarse_mesh(p3m_ctx *ctx, float a_mid, float dt, float mass_p); /* coarse_mesh.f90 */
int p3m_hip_delete_particles(p3m_ctx *ctx, const float *move_back);        /* delete_particles.f90 */
int p3m_hip_get_step_out(p3m_ctx *ctx, float a_mid, p3m_step_out *out);
/* projection.f90:2-188 (density projections).  Call between p3m_hip_link_list_and_pass and
 * p3m_hip_delete_particles, as cubepm.f90:193-228 does.  pxy / pxz / pyz: host arrays of nf_physical_dim^2 floats each
 * (nf_physical_dim = nodes_dim * nf_physical_node_dim), in the reference's memory order rho_pxy(x,y), rho_pxz(x,z),
 * rho_pyz(y,z) with the first index fastest; they receive THIS rank's contribution (:170-181: only ranks at coordinate 0
 * of the projected axis add anything) -- the sum over ranks is the host's mpi_reduce (:41-54).  rho_node: the
 * rank's projected mass (:183).  The deposit is the CIC of fine_cic_mass.f90 whatever P3M_FLAG_NGP says. */
int p3m_hip_projection(p3m_ctx *ctx, float mass_p, float *pxy, float *pxz, float *pyz, double *rho_node);

/* -- mesh-level probes for parity tests and the roofline benchmark ----------------------- */
/* One tile's density after deposit, reference layout rho_f(nf_tile+2, nf_tile, nf_tile)
   (particle_mesh_threaded.f90:100-164). Requires p3m_hip_link_list_and_pass first. */
int p3m_hip_probe_tile_density(p3m_ctx *ctx, int32_t tile_x, int32_t tile_y, int32_t tile_z,
                               float mass_p, float *rho_f);
/* One tile's force_f(3, nf_buf-1:nf_tile-nf_buf+1, ...) from a given density (:176-204). */
int p3m_hip_probe_tile_force(p3m_ctx *ctx, const float *rho_f, float *force_f, float *force_max2);
/* Local coarse density rho_c(nc_node_dim^3) (coarse_mass.f90) and force_c(3,0:ncn+1,...) */
int p3m_hip_probe_coarse(p3m_ctx *ctx, float mass_p, float *rho_c, float *force_c);
/* In-place 3-D r2c / c2r of a host array in the reference layout (nf+2, nf, nf) through the
   library's FFT: dir=+1 forward (fftw2.f90:19), dir=-1 inverse incl. 1/n^3 (fftw2.f90:21-22). */
int p3m_hip_fft3d(p3m_ctx *ctx, float *data, int32_t n, int32_t dir);
/* Fine-mesh sweep over all tiles `reps` times with device-resident inputs, timed with HIP
   events on the library's stream; returns average milliseconds per sweep (benchmark leg). */
/* coarse_power.f90 for a single-rank context (see p3m_hip_group_coarse_power) */
int p3m_hip_coarse_power(p3m_ctx *ctx, float mass_p, float box, float *ps);
int p3m_hip_time_fine_sweep(p3m_ctx *ctx, float mass_p, int32_t reps, float *ms_per_sweep);
/* The gather half of the fine mesh (particle_mesh_threaded.f90:208-319: max |F|^2 + NGP or CIC interpolation + kick) over the force
   boxes the last sweep left, `reps` times with dt = 0 (the velocities keep their values), HIP events on the library's stream;
   returns the average milliseconds per pass.  Call after a step or after p3m_hip_time_fine_sweep (sorted records, force boxes). */
int p3m_hip_time_fine_gather(p3m_ctx *ctx, int32_t reps, float *ms_per_pass);
/* One pass kernel of the fine-mesh FFT over the whole tile batch, launched `reps` times between
   HIP events on the library's stream; returns the average milliseconds per launch.
   which: 0 x-forward (r2c rows), 1 y-forward lines, 2 z-forward lines, 3 z-inverse lines fused with
   the i*K multiply, 4 y-inverse lines, 5 x-inverse (c2r rows) + force-box extraction, 6 the multiply +
   inverse z pass on its own (the second launch of the un-fused z pair that tiles longer than 608 cells and
   P3M_Z_UNFUSED=1 run instead of 3: 2 then 6), 7 the inverse x pass with the NGP kick inside (kick_fused.hip: what NGP whole steps
   run instead of 5 and the kick; call after a whole step; the velocities are not stored).  *batch returns the number of tiles one
   launch processed. */
int p3m_hip_time_fft_pass(p3m_ctx *ctx, int32_t which, int32_t reps, float *ms_per_launch, int32_t *batch);
/* Benchmark hook for the two PP kernels (particle_mesh_threaded.f90:324-361 and :378-624) on the sorted records with ghosts
   (call p3m_hip_link_list_and_pass first; velocities are kicked `reps`+1 times): average ms per launch and the number of
   pair evaluations one launch performs (a pair of two kicked records counts twice: each member sums over its partners). */
int p3m_hip_time_pp(p3m_ctx *ctx, float a_mid, float dt, float mass_p, int32_t reps, float *ms_intra, float *ms_ext, int64_t *evals_intra,
                    int64_t *evals_ext);
/* Per-phase GPU times of the last whole step -- what a reference host built with -DMPI_TIME prints (timers.f90:68-77; call sites
   link_list.f90:138-143, particle_pass.f90:762-767, coarse_mesh.f90; test.log:80-94).  Off by default (two timing events per span);
   `on` != 0 enables it for the following steps.  ms12, in milliseconds of GPU time between the phase's first and last kernel on its stream:
     0 update_position (+ the copy of the previous step's delete_particles, which rides on it)   1 link_list (the cell sort)
     2 particle_pass   3 fine mass assignment   4 fine FFT + multiply + inverse (NGP whole steps: up to the inverse y pass)
     5 fine force maximum + kick (NGP whole steps: with the inverse x pass inside)   6 PP intra-cell   7 PP extended
     8 coarse_mass   9 coarse_force (+ buffer, max; on the second stream, underneath 3-4)   10 coarse_velocity (0 when it rides on 5)
     11 delete_particles (count + scan; the copy is deferred into the next step's phase 0)
   A group reports the spans of ITS ranks (the host reduces max / avg / min over processes as timers.f90 does). */
#define P3M_NPHASES 12
int p3m_hip_phase_timing(p3m_ctx *ctx, int32_t on);
int p3m_hip_last_phase_ms(p3m_ctx *ctx, float *ms12);
/* HIP stream the kernels are launched on (for hipEvent timing by the host). */
void *p3m_hip_stream(p3m_ctx *ctx);

/* -- host time loop: timestep.f90 ---------------------------------
 * Pure host arithmetic (no device work): the step-size choice and the scale-factor integration the
 * reference does between two particle_mesh calls, so that a host without cubep3m's Fortran can run
 * multi-step simulations.  A Fortran host keeps its own timestep.f90 and never calls these. */
#define P3M_MAX_INPUT 100   /* cubepm.par:183 max_input */
typedef struct p3m_time_params {
  int32_t cosmo;        /* cubepm.par:15                                                   */
  int32_t restrict_da;  /* -DRESTRICT_DA (timestep.f90:75-88)                              */
  float omega_m, omega_l, wde;              /* parameters file; cubepm.par:19              */
  float dt_scale, dt_max, ra_max, da_max;   /* cubepm.par:27-30                            */
  int32_t num_checkpoints, num_projections, num_halofinds;
  /* scale factors of the output steps; timestep.f90 reads entry cur_* even past num_* (the
     reference leaves zeros there): callers pad with a value > 1                                    */
  float a_checkpoint[P3M_MAX_INPUT], a_projection[P3M_MAX_INPUT], a_halofind[P3M_MAX_INPUT];
  /* the non-cosmological test runs (cosmo = 0, timestep.f90:197-216; cubepm.par:61-68): pairwise_ic / shake_test_ic force
     dt = 1 (:210-211); pair_infall (with P3M_FLAG_PPINT) limits dt by 0.05 / sqrt(G mass_p / cur_sep^2) and leaves the
     extended-PP limit out (:204-206); cur_sep is what report_pair.f90:49 measured last, the host keeps it current        */
  int32_t pairwise_ic, pair_infall, shake_test_ic;
  float cur_sep, mass_p;
  /* -DChaplygin (timestep.f90:251-252, subroutine Chaplygin :296-339): expansion() integrates the Friedmann equation of a Chaplygin
     gas instead; omega_ch from the parameters file, A_ch and alpha_ch from cubepm.par:21-22 (there: 1 and 0)                      */
  int32_t chaplygin;
  float omega_ch, A_ch, alpha_ch;
} p3m_time_params;
typedef struct p3m_time_state {   /* the COMMON variables timestep.f90 reads and writes (cubepm.fh:19-29) */
  int32_t nts;
  float a, a_mid, da, dt, dt_old, dt_gas, tau, t;
  int32_t cur_checkpoint, cur_projection, cur_halofind;   /* 1-based; advanced by the HOST after an output step,
                                                             as checkpoint.f90 / projection.f90 / halofind.f90 do */
  int32_t checkpoint_step, projection_step, halofind_step, final_step;
} p3m_time_state;
/* subroutine expansion (timestep.f90:218-293): third-order integration of the Friedmann equation over two half steps */
void p3m_hip_expansion(const p3m_time_params *par, float a0, float dt0, float *da1, float *da2);
/* subroutine timestep (timestep.f90:2-216) on rank 0's values; flags: P3M_FLAG_PPINT / P3M_FLAG_PP_EXT select the limits
   that enter the minimum (:93-115).  The four dt_*_acc are what p3m_step_out returned for the previous step. */
int p3m_hip_timestep(const p3m_time_params *par, uint32_t flags, p3m_time_state *st, float dt_f_acc, float dt_pp_acc,
                     float dt_pp_ext_acc, float dt_c_acc);

/* -- particle files: checkpoint.f90, particle_initialization.f90 ------
 * Host code.  `binary` != 0: form='binary' (-DBINARY: a byte stream); 0: form='unformatted' (every Fortran
 * WRITE is a record framed by two 4-byte length markers -- one record PER PARTICLE in the checkpoint and in
 * the unformatted IC file).  `ppint` != 0: the header carries dt_pp_acc (-DPPINT, checkpoint.f90:55-61). */
typedef struct p3m_ckpt_header {   /* checkpoint.f90:55-61 */
  int32_t np_local; float a, t, tau; int32_t nts; float dt_f_acc, dt_pp_acc, dt_c_acc;
  int32_t cur_checkpoint, cur_projection, cur_halofind; float mass_p;
} p3m_ckpt_header;
/* <z>xv<rank>.dat: header, then xv(1:3,j) - shake_offset, xv(4:6,j) per particle (checkpoint.f90:63-83) */
int p3m_hip_write_checkpoint(const char *path, const p3m_ckpt_header *h, const float *xv6, const float *shake_offset3, int32_t binary, int32_t ppint);
/* header only (xv6 == NULL) or header + particles (particle_initialization.f90:114-145); cap = room in xv6 (particles) */
int p3m_hip_read_checkpoint(const char *path, p3m_ckpt_header *h, float *xv6, int64_t cap, int32_t binary, int32_t ppint);
/* <z>PID<rank>.dat: the same header, then PID(j) per particle (checkpoint.f90:86-124) */
int p3m_hip_write_pid_checkpoint(const char *path, const p3m_ckpt_header *h, const int64_t *pid, int32_t binary, int32_t ppint);
int p3m_hip_read_pid_checkpoint(const char *path, p3m_ckpt_header *h, int64_t *pid, int64_t cap, int32_t binary, int32_t ppint);
/* xv<rank>.ic: np_local, then xv(:,i) per particle (unformatted) or as one block (binary) (particle_initialization.f90:296-332) */
/* projection.f90:62-113: one projection file = the scalar a, then the nf_physical_dim^2 map; `binary` as for the checkpoints */
int p3m_hip_write_projection(const char *path, float a, const float *map, int32_t nf_physical_dim, int32_t binary);
/* coarse_power.f90:121-133: <z>ps.dat, formatted, one '(2f20.10)' line per bin of the spectrum p3m_hip_coarse_power returns */
int p3m_hip_write_power(const char *path, const float *ps, int32_t nc_dim);
int p3m_hip_read_projection(const char *path, float *a, float *map, int32_t nf_physical_dim, int32_t binary);
int p3m_hip_write_ic(const char *path, const float *xv6, int32_t np_local, int32_t binary);
int p3m_hip_read_ic(const char *path, float *xv6, int64_t cap, int32_t *np_local, int32_t binary);

/* -- multi-rank: a group of logical ranks (the reference's nodes_dim^3 MPI ranks) ---------
 * One process drives one GPU and owns nodes_dim^3 / nprocs consecutive logical ranks; ranks on the
 * same GPU exchange by device copies, ranks on different GPUs by RCCL send/recv over xGMI
 * (replaces the MPI calls of particle_pass.f90, fftw3ds.f90, coarse_force_buffer.f90 and the
 * mpi_reduce/mpi_bcast pairs of particle_mesh_threaded.f90:646-696, coarse_max_dt.f90:34-37).
 * params->rank is ignored (set per logical rank), params->device selects the GPU. */
typedef struct p3m_group p3m_group;
int p3m_hip_group_create(const p3m_params *params, int32_t proc, int32_t nprocs, p3m_group **out);
void p3m_hip_group_destroy(p3m_group *g);
/* nprocs > 1: every process calls this with the id rank 0 obtained from p3m_hip_rccl_unique_id and the
   host broadcast.  force_for_local_peers != 0 routes even same-GPU exchanges through RCCL (test mode). */
int p3m_hip_group_comm_init_rccl(p3m_group *g, const void *unique_id_128, int32_t force_for_local_peers);
/* host-callback transport instead of RCCL (the struct is copied; `user` must outlive the group) */
int p3m_hip_group_set_transport(p3m_group *g, const p3m_transport *t);
int32_t p3m_hip_group_nlocal(const p3m_group *g);                 /* logical ranks owned by this process */
int32_t p3m_hip_group_local_rank(const p3m_group *g, int32_t i);  /* logical rank id of the i-th local one */
p3m_ctx *p3m_hip_group_ctx(p3m_group *g, int32_t i);              /* its context (probes, derived sizes) */
int p3m_hip_group_set_kernel_tables(p3m_group *g, const float *fine_table, const float *coarse_table);
int p3m_hip_group_upload_particles(p3m_group *g, int32_t i, const float *xv6, const int64_t *pid, int32_t np_local);
int p3m_hip_group_download_particles(p3m_group *g, int32_t i, float *xv6, int64_t *pid, int32_t *np_local);
/* `particle_mesh` on all ranks; `out` holds the global dt limits and DIAG sums (identical on every process) */
/* update_position.f90 on every local rank (the output-step half drift of cubepm.f90:196-198 uses it with dt_old = 0) */
int p3m_hip_group_update_position(p3m_group *g, float dt, float dt_old, const float *offset);
/* p3m_hip_phase_timing / p3m_hip_last_phase_ms for the ranks of this process */
int p3m_hip_group_phase_timing(p3m_group *g, int32_t on);
int p3m_hip_group_last_phase_ms(p3m_group *g, float *ms12);
int p3m_hip_group_particle_mesh(p3m_group *g, float a_mid, float dt, float dt_old, float mass_p,
                                const float *offset, const float *move_back, p3m_step_out *out);
int p3m_hip_group_probe_coarse(p3m_group *g, float mass_p, int32_t i, float *rho_c, float *force_c);
/* Green's functions a multi-rank host already holds (kernel_checkpoint.f90 / a restart): kern_f as for p3m_hip_set_kernels_raw
 * (identical on every rank), and kern_c_slabs[i] = the z-slab kern_c(3, nc_dim/2+1, nc_dim, nc_slab) of local rank i in the
 * reference's layout (cubep3m.fh:56; kz = rank*nc_slab + local index).  The library keeps k space in transposed order (every rank
 * owns a ky slab), so the slabs are redistributed once through the group's transport.  Slab decomposition only. */
int p3m_hip_group_set_kernels_raw(p3m_group *g, const float *kern_f, const float *const *kern_c_slabs);
/* The distributed coarse transform on its own (coarse_force.f90:18-90 through fftw3ds.f90:4-183 / p3dfft_coarse.f90), on any
 * group and in particular on a P3M_FLAG_COARSE_ONLY one (the literal 1024^3 slab FFT of BASELINE config 4):
 *   set_coarse_density  rho_c of local rank i from the host, float[ncn][ncn][ncn] (what coarse_mass leaves in rho_c)
 *   coarse_transform    what = 0: the forward transform (cube -> x-lines, x and y passes, all-to-all, z pass: rho-hat of the
 *                       own ky slab); what = 1: coarse_force -- forward transform, i K_c multiply, three inverse transforms,
 *                       slab -> cube, force halo (coarse_force_buffer.f90), maximum.  Runs once untimed, then `reps` times
 *                       between HIP events on the group's stream: *ms = average milliseconds per run (reps = 0: once, no timing)
 *   get_coarse_hat      rho-hat of local rank i as the device holds it: float2[s][ncl][nc][16] = (local ky, kx chunk, kz, kx
 *                       within the chunk); slabs: ky = rank*nc_slab + local ky, ncl = all chunks of (nc/2+1 rounded up to 16)/16
 *   get_coarse_force    force_c(3, 0:ncn+1, 0:ncn+1, 0:ncn+1) of local rank i, component fastest (as p3m_hip_group_probe_coarse)
 *   coarse_exchange_bytes  bytes one rank sends to ONE peer in the y<->z transpose of one transform (SURVEY 8d "per-link bytes") */
/* Who this process is in the group's RCCL communicator and which device it drives: comm_count / comm_rank from
 * ncclCommCount / ncclCommUserRank (both -1 when no communicator was set up), the HIP device ordinal, and the device's UUID
 * as 32 hex digits + NUL (uuid_hex33).  bench.py prints these per rank so that a multi-GPU line can be audited. */
int p3m_hip_group_comm_info(p3m_group *g, int32_t *comm_count, int32_t *comm_rank, int32_t *device, char *uuid_hex33);
int p3m_hip_group_set_coarse_density(p3m_group *g, int32_t i, const float *rho_c);
int p3m_hip_group_coarse_transform(p3m_group *g, int32_t what, int32_t reps, float *ms);
int p3m_hip_group_get_coarse_hat(p3m_group *g, int32_t i, float *hat, int64_t nfloats);
int p3m_hip_group_get_coarse_force(p3m_group *g, int32_t i, float *force_c);
int64_t p3m_hip_group_coarse_exchange_bytes(const p3m_group *g);
/* projection.f90 for every logical rank of this process: ghost pass, sort, CIC projection, ghost removal (the sequence of
 * cubepm.f90:193-228); the maps receive the sum over THIS process's ranks, rho_tot the sum over all ranks (:34-35); a host
 * running several processes adds the maps up (the reference's mpi_reduce, :41-54). */
int p3m_hip_group_projection(p3m_group *g, float mass_p, float *pxy, float *pxz, float *pyz, double *rho_tot);
/* coarse_power.f90:2-139 (called from coarse_force.f90:108 when coarse_ps is set): the mass power spectrum of the coarse
 * density of the LAST particle_mesh step (its rho-hat is still on the device).  ps: float[nc_dim][2] = the rows of <z>ps.dat,
 * (k = 2 pi (bin-1) / box, Delta^2(k) = 4 pi (bin-1)^3 <P>), bins without modes (count, 0).  `box` is the parameter of the
 * reference's `parameters` file.  Every process of a group receives the full spectrum. */
int p3m_hip_group_coarse_power(p3m_group *g, float mass_p, float box, float *ps);
/* Host-only (no device needed): the exchange schedule of the distributed coarse transform, for hosts that route the
 * messages themselves and for tests.  which = 0: cube <-> x-lines (pack_slab, fftw3ds.f90:24-52; with P3M_FLAG_PENCIL
 * pack_pencils, p3dfft_coarse.f90:69-127); 1: x <-> y transpose (pencils only); 2: y <-> z transpose.  Block j of `rank`
 * goes to *peer, where it is block *index.  Returns the number of blocks (peers) of that exchange, 0 if it does not
 * exist in this decomposition, or a negative P3M_E* code. */
int32_t p3m_hip_coarse_fft_schedule(int32_t nodes_dim, uint32_t flags, int32_t rank, int32_t which, int32_t j,
                                    int32_t *peer, int32_t *index);

/* -- F77-ABI one-call wrapper in the style of pp_force_c_ (nbody-ueli.cu:368) ------------ */
/* Single-rank hosts: uploads xv/PID, runs the step, downloads, returns the four dt limits.
   handle: integer(8) holding the context (0 on first call: created from params and the
   kernel tables; pass it back on later calls). */
void particle_mesh_hip_(int64_t *handle, const p3m_params *params,
                        const float *fine_table, const float *coarse_table,
                        float *xv, int64_t *pid, int32_t *np_local,
                        const float *a_mid, const float *dt, const float *dt_old,
                        const float *mass_p, const float *offset, const float *move_back,
                        float *dt_f_acc, float *dt_pp_acc, float *dt_pp_ext_acc,
                        float *dt_c_acc, int32_t *ierr);

#ifdef __cplusplus
}
#endif
#endif /* P3M_HIP_H */
