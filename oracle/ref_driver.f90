! TEST INFRASTRUCTURE -- driver for oracle/_ref: exposes the reference's COMMON state
! (source_threads/cubep3m.fh:147-171) and its FFT-FREE hot-path subroutines to ctypes.
! Everything arithmetic that runs here is the reference's own object code, compiled from
! /root/reference/source_threads/*.f90 where they lie (oracle/build_ref.sh).  This file only
! moves data in and out of COMMON and repeats the loop headers of particle_mesh_threaded.f90
! that select which coarse-cell chains a tile deposits (:118-130,154-160).
! The FFT-dependent parts of particle_mesh (FFTW 2.1.5, absent) are NOT built or emulated.

subroutine ref_init() bind(C, name="ref_init")
  implicit none
  include 'mpif.h'
  include 'cubepm.fh'
  logical :: flag
  call mpi_initialized(flag, ierr)
  if (.not. flag) call mpi_initialize            ! mpi_initialization.f90:2
  np_local = 0
  shake_offset = 0.0
  nts = 1
end subroutine ref_init

subroutine ref_sizes(out) bind(C, name="ref_sizes")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int) :: out(16)
  out(1) = nodes_dim; out(2) = tiles_node_dim; out(3) = nf_tile; out(4) = nf_buf
  out(5) = nc_node_dim; out(6) = nc_dim; out(7) = max_np; out(8) = hoc_nc_l; out(9) = hoc_nc_h
  out(10) = nf_physical_node_dim; out(11) = rank; out(12) = cores; out(13) = max_buf
  out(14) = cart_coords(1); out(15) = cart_coords(2); out(16) = cart_coords(3)
end subroutine ref_sizes

subroutine ref_neighbors(out) bind(C, name="ref_neighbors")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int) :: out(6)
  out(1:6) = cart_neighbor(1:6)
end subroutine ref_neighbors

subroutine ref_set_scalars(a_mid_in, dt_in, dt_old_in, mass_p_in) bind(C, name="ref_set_scalars")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float), value :: a_mid_in, dt_in, dt_old_in, mass_p_in
  a_mid = a_mid_in; dt = dt_in; dt_old = dt_old_in; mass_p = mass_p_in
end subroutine ref_set_scalars

subroutine ref_set_particles(xv_in, pid_in, n) bind(C, name="ref_set_particles")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int), value :: n
  real(c_float) :: xv_in(6, n)
  integer(c_int64_t) :: pid_in(n)
  xv(:, 1:n) = xv_in(:, 1:n)
  PID(1:n) = pid_in(1:n)
  np_local = n
end subroutine ref_set_particles

function ref_np_local() bind(C, name="ref_np_local") result(n)
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int) :: n
  n = np_local
end function ref_np_local

subroutine ref_get_particles(xv_out, pid_out) bind(C, name="ref_get_particles")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: xv_out(6, *)
  integer(c_int64_t) :: pid_out(*)
  xv_out(:, 1:np_local) = xv(:, 1:np_local)
  pid_out(1:np_local) = PID(1:np_local)
end subroutine ref_get_particles

subroutine ref_get_lists(hoc_out, ll_out) bind(C, name="ref_get_lists")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int) :: hoc_out(*), ll_out(*)
  integer :: i, j, k, m
  m = 0
  do k = hoc_nc_l, hoc_nc_h
    do j = hoc_nc_l, hoc_nc_h
      do i = hoc_nc_l, hoc_nc_h
        m = m + 1
        hoc_out(m) = hoc(i, j, k)
      enddo
    enddo
  enddo
  ll_out(1:np_local) = ll(1:np_local)
end subroutine ref_get_lists

! --- the reference's own subroutines, called as particle_mesh calls them -------------------
subroutine ref_update_position() bind(C, name="ref_update_position")
  call update_position                             ! update_position.f90:2
end subroutine
subroutine ref_link_list() bind(C, name="ref_link_list")
  call link_list                                   ! link_list.f90:3
end subroutine
subroutine ref_particle_pass() bind(C, name="ref_particle_pass")
  call particle_pass                               ! particle_pass.f90:2
end subroutine
subroutine ref_delete_particles() bind(C, name="ref_delete_particles")
  call delete_particles                            ! delete_particles.f90:2
end subroutine

! fine deposit of one tile into rho_f(:,:,:,1): loop headers of particle_mesh_threaded.f90:118-130
! (NGP bounds) / :123-124,154-160 (CIC bounds and boundary selection); the deposits themselves are the
! reference's fine_ngp_mass.f90 / fine_cic_mass.f90 / fine_cic_mass_buffer.f90.
subroutine ref_fine_deposit(tile, use_ngp, rho_out) bind(C, name="ref_fine_deposit")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int) :: tile(3)
  integer(c_int), value :: use_ngp
  real(c_float) :: rho_out(nf_tile+2, nf_tile, nf_tile)
  integer(4) :: i, j, k, pp, cic_l(3), cic_h(3), thread
  thread = 1
  rho_f(:, :, :, thread) = 0.0
  if (use_ngp /= 0) then
    cic_l(:) = nc_tile_dim * tile(:) + 2 - nc_buf
    cic_h(:) = nc_tile_dim * (tile(:) + 1) + nc_buf - 1
  else
    cic_l(:) = nc_tile_dim * tile(:) + 1 - nc_buf
    cic_h(:) = nc_tile_dim * (tile(:) + 1) + nc_buf
  endif
  do k = cic_l(3), cic_h(3)
    do j = cic_l(2), cic_h(2)
      do i = cic_l(1), cic_h(1)
        pp = hoc(i, j, k)
        if (use_ngp /= 0) then
          call fine_ngp_mass(pp, tile, thread)
        else
          if (i == cic_l(1) .or. i == cic_h(1) .or. j == cic_l(2) .or. j == cic_h(2) .or. &
              k == cic_l(3) .or. k == cic_h(3)) then
            call fine_cic_mass_boundry(pp, tile, thread)
          else
            call fine_cic_mass(pp, tile, thread)
          endif
        endif
      enddo
    enddo
  enddo
  rho_out(:, :, :) = rho_f(:, :, :, thread)
end subroutine ref_fine_deposit

! fine_velocity.f90:6-237 (the non-inlined twin of particle_mesh_threaded.f90:208-368: force maximum, NGP/CIC gather +
! kick, intra-cell PP) on a caller-supplied force_f, which stands where :202 would have filled it from the FFT.
! out = f_force_max(1) (fine_velocity.f90:39-50: the maximum of |F|, not of |F|^2), pp_force_max(1)
subroutine ref_fine_velocity(tile, f_in, out) bind(C, name="ref_fine_velocity")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  integer(c_int) :: tile(3)
  real(c_float) :: f_in(3, nf_buf-1:nf_tile-nf_buf+1, nf_buf-1:nf_tile-nf_buf+1, nf_buf-1:nf_tile-nf_buf+1)
  real(c_float) :: out(2)
  integer(4) :: thread
  thread = 1
  force_f(:, :, :, :, thread) = f_in
  f_force_max(thread) = 0.0
  pp_force_max(thread) = 0.0
  call fine_velocity(tile, thread)
  out(1) = f_force_max(thread)
  out(2) = pp_force_max(thread)
end subroutine ref_fine_velocity

subroutine ref_coarse_mass(rho_out) bind(C, name="ref_coarse_mass")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: rho_out(nc_node_dim, nc_node_dim, nc_node_dim)
  call coarse_mass                                 ! coarse_mass.f90:2
  rho_out = rho_c
end subroutine ref_coarse_mass

! interior of force_c <- caller (stands where coarse_force.f90:52,71,90 would have filled it from the
! FFT); then the reference's halo exchange, max-dt and kick run on it.
subroutine ref_set_force_c(f_in) bind(C, name="ref_set_force_c")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: f_in(3, nc_node_dim, nc_node_dim, nc_node_dim)
  force_c = 0.0
  force_c(:, 1:nc_node_dim, 1:nc_node_dim, 1:nc_node_dim) = f_in
end subroutine ref_set_force_c

subroutine ref_get_force_c(f_out) bind(C, name="ref_get_force_c")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: f_out(3, 0:nc_node_dim+1, 0:nc_node_dim+1, 0:nc_node_dim+1)
  f_out = force_c
end subroutine ref_get_force_c

subroutine ref_coarse_force_buffer() bind(C, name="ref_coarse_force_buffer")
  call coarse_force_buffer                         ! coarse_force_buffer.f90:2
end subroutine
function ref_coarse_max_dt() bind(C, name="ref_coarse_max_dt") result(v)
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: v
  call coarse_max_dt                               ! coarse_max_dt.f90:2
  v = dt_c_acc
end function
subroutine ref_coarse_velocity() bind(C, name="ref_coarse_velocity")
  call coarse_velocity                             ! coarse_velocity.f90:7
end subroutine

! ranks started by mpiexec must leave together (mpi_finalize is what cubepm.f90:254 does)
subroutine ref_finalize() bind(C, name="ref_finalize")
  implicit none
  include 'mpif.h'
  integer :: ierr2
  call mpi_barrier(mpi_comm_world, ierr2)
  call mpi_finalize(ierr2)
end subroutine ref_finalize

! --- timestep.f90 (host time loop, SURVEY section 8f rank 1) --------------------------------
! rv_in : a, tau, t, dt, dt_old, dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc ; iv_in : nts, cur_checkpoint, cur_projection, cur_halofind
subroutine ref_time_set(rv_in, iv_in, a_chk, n_chk, a_prj, n_prj, a_hf, n_hf) bind(C, name="ref_time_set")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: rv_in(9)
  integer(c_int) :: iv_in(4)
  integer(c_int), value :: n_chk, n_prj, n_hf
  real(c_float) :: a_chk(n_chk), a_prj(n_prj), a_hf(n_hf)
  a = rv_in(1); tau = rv_in(2); t = rv_in(3); dt = rv_in(4); dt_old = rv_in(5)
  dt_f_acc = rv_in(6); dt_pp_acc = rv_in(7); dt_pp_ext_acc = rv_in(8); dt_c_acc = rv_in(9)
  nts = iv_in(1); cur_checkpoint = iv_in(2); cur_projection = iv_in(3); cur_halofind = iv_in(4)
  a_checkpoint = 100.0; a_projection = 100.0; a_halofind = 100.0
  num_checkpoints = n_chk; num_projections = n_prj; num_halofinds = n_hf
  a_checkpoint(1:n_chk) = a_chk(1:n_chk); a_projection(1:n_prj) = a_prj(1:n_prj); a_halofind(1:n_hf) = a_hf(1:n_hf)
  final_step = .false.
end subroutine ref_time_set

subroutine ref_timestep() bind(C, name="ref_timestep")
  call timestep
end subroutine

! rv_out : a, a_mid, da, dt, dt_old, dt_gas, tau, t ; iv_out : nts, checkpoint_step, projection_step, halofind_step, final_step
subroutine ref_time_get(rv_out, iv_out) bind(C, name="ref_time_get")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: rv_out(8)
  integer(c_int) :: iv_out(5)
  rv_out(1) = a; rv_out(2) = a_mid; rv_out(3) = da; rv_out(4) = dt; rv_out(5) = dt_old; rv_out(6) = dt_gas; rv_out(7) = tau; rv_out(8) = t
  iv_out(1) = nts
  iv_out(2) = merge(1, 0, checkpoint_step); iv_out(3) = merge(1, 0, projection_step)
  iv_out(4) = merge(1, 0, halofind_step); iv_out(5) = merge(1, 0, final_step)
end subroutine ref_time_get

subroutine ref_expansion(a0, dt0, da1, da2) bind(C, name="ref_expansion")
  use iso_c_binding
  implicit none
  real(c_float), value :: a0, dt0
  real(c_float) :: da1, da2
  real(4) :: a0_, dt0_
  a0_ = a0; dt0_ = dt0
  call expansion(a0_, dt0_, da1, da2)
end subroutine ref_expansion

! cosmo, dt_scale, dt_max, ra_max, da_max, wde, omega_m, omega_l of this build (cubepm.par, parameters)
subroutine ref_time_params(out) bind(C, name="ref_time_params")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: out(8)
  out(1) = merge(1.0, 0.0, cosmo); out(2) = dt_scale; out(3) = dt_max; out(4) = ra_max; out(5) = da_max
  out(6) = wde; out(7) = omega_m; out(8) = omega_l
end subroutine ref_time_params

! --- checkpoint.f90 / particle_initialization.f90 (particle files, SURVEY section 8f rank 2) ------
! rv_in : a, t, tau, dt_f_acc, dt_pp_acc, dt_c_acc, mass_p, z_write, shake_offset(3) ; iv_in : nts, cur_checkpoint, cur_projection, cur_halofind
subroutine ref_checkpoint(rv_in, iv_in) bind(C, name="ref_checkpoint")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: rv_in(11)
  integer(c_int) :: iv_in(4)
  a = rv_in(1); t = rv_in(2); tau = rv_in(3); dt_f_acc = rv_in(4); dt_pp_acc = rv_in(5); dt_c_acc = rv_in(6); mass_p = rv_in(7)
  shake_offset = rv_in(9:11)
  nts = iv_in(1); cur_checkpoint = iv_in(2); cur_projection = iv_in(3); cur_halofind = iv_in(4)
  z_checkpoint(cur_checkpoint) = rv_in(8)
  call checkpoint
end subroutine ref_checkpoint

subroutine ref_particle_initialize() bind(C, name="ref_particle_initialize")
  call particle_initialize
end subroutine

! --- projection.f90 (density projections, SURVEY section 8f rank 3) ------------------------------
! needs link_list + particle_pass; writes <z>proj_{xy,xz,yz}.dat into output_path and leaves the maps in rho_pxy/pxz/pyz
! rv_in : a, mass_p, z_projection
subroutine ref_projection(rv_in, pxy, pxz, pyz) bind(C, name="ref_projection")
  use iso_c_binding
  implicit none
  include 'cubepm.fh'
  real(c_float) :: rv_in(3)
  real(c_float) :: pxy(nf_physical_dim, nf_physical_dim), pxz(nf_physical_dim, nf_physical_dim), pyz(nf_physical_dim, nf_physical_dim)
  a = rv_in(1); mass_p = rv_in(2)
  cur_projection = 1
  z_projection(1) = rv_in(3)
  call projection
  pxy = rho_pxy; pxz = rho_pxz; pyz = rho_pyz
end subroutine ref_projection
