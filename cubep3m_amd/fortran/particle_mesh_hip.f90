!! particle_mesh_hip.f90 -- drop-in replacement for the reference's `particle_mesh`
!! (source_threads/particle_mesh_threaded.f90:2): same name, no arguments, same COMMON state
!! (cubepm.fh), so that swapping ONE object in the link line
!!     OBJS = ... particle_mesh_threaded.o ...   ->   ... particle_mesh_hip.o ...   (+ -lp3m_hip)
!! (source_threads/Makefile_gnu_sfftw2:8) moves the whole gravity step onto an MI355X.
!! The host keeps cubep3m's decomposition, time-step loop, RNG, I/O; this file is a thin
!! ISO_C_BINDING shim over include/p3m_hip.h.  Single-rank builds (nodes_dim = 1); multi-rank hosts
!! additionally hand the library a transport (include/p3m_hip.h, p3m_transport).
!! RESIDENT PARTICLES (opt-in, P3M_HIP_RESIDENT=1): between output steps the main loop of cubepm.f90 (:103-236) never reads xv
!! except through particle_mesh, and on checkpoint / projection / halofind steps through update_position, checkpoint,
!! link_list, ...; with the switch on, the particles stay on the device: they are uploaded when the host's copy is the newer one
!! (first call, and the call after an output step, where cubepm.f90:175-176 drifted xv on the host) and downloaded only when
!! the flags in COMMON /lvar/ say the host reads them next (checkpoint_step .or. projection_step .or. halofind_step .or.
!! final_step; or the loop's other exits, nts == max_nts and a > 1, cubepm.f90:235).  np_local follows the device every step.
!! The DEFAULT is the copy-in / copy-out of every step, because not every reader of xv announces itself in COMMON:
!! cubepm_kill.f90 decides `kill_step` AFTER particle_mesh returns (a local of its main program) and then writes xv through
!! checkpoint_kill; a -DMHD host reads xv in its gas coupling.  Such hosts must leave the switch off.
!! Compile with the reference's own flags, e.g.
!!   flang -cpp -ffree-form -I<source_threads> -DNGP -DPPINT -DPP_EXT -DDISP_MESH -c particle_mesh_hip.f90
subroutine particle_mesh
  use iso_c_binding
  implicit none
  include 'cubepm.fh'

  type, bind(C) :: p3m_params        ! struct p3m_params, include/p3m_hip.h
    integer(c_int32_t) :: nodes_dim, tiles_node_dim, nf_tile, nf_cutoff, nf_buf, mesh_scale, pp_range, cores
    integer(c_int32_t) :: flags
    real(c_float)      :: rsoft, pp_bias, dt_pp_scale, density_buffer
    integer(c_int32_t) :: rank, device
  end type
  type, bind(C) :: p3m_step_out      ! struct p3m_step_out
    real(c_float)      :: dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc
    real(c_double)     :: sum_rho_f, sum_rho_c
    integer(c_int64_t) :: np_total
    integer(c_int32_t) :: np_local, np_ghost, np_deleted
    real(c_float)      :: f_force_max, pp_force_max, pp_ext_force_max, c_force_max
  end type

  interface
    integer(c_int) function p3m_hip_create(params, ctx) bind(C, name="p3m_hip_create")
      import :: c_int, c_ptr, p3m_params
      type(p3m_params), intent(in) :: params
      type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function p3m_hip_set_kernel_tables(ctx, fine, coarse) bind(C, name="p3m_hip_set_kernel_tables")
      import :: c_int, c_ptr, c_float
      type(c_ptr), value :: ctx
      real(c_float), intent(in) :: fine(*), coarse(*)
    end function
    integer(c_int) function p3m_hip_upload_particles(ctx, xv6, pid, n) bind(C, name="p3m_hip_upload_particles")
      import :: c_int, c_ptr, c_float, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx
      real(c_float), intent(in) :: xv6(6, *)
      integer(c_int64_t), intent(in) :: pid(*)
      integer(c_int32_t), value :: n
    end function
    integer(c_int) function p3m_hip_download_particles(ctx, xv6, pid, n) bind(C, name="p3m_hip_download_particles")
      import :: c_int, c_ptr, c_float, c_int64_t, c_int32_t
      type(c_ptr), value :: ctx
      real(c_float), intent(out) :: xv6(6, *)
      integer(c_int64_t), intent(out) :: pid(*)
      integer(c_int32_t), intent(out) :: n
    end function
    integer(c_int) function p3m_hip_particle_mesh(ctx, a_mid, dt, dt_old, mass_p, offset, move_back, sout) &
        bind(C, name="p3m_hip_particle_mesh")
      import :: c_int, c_ptr, c_float, p3m_step_out
      type(c_ptr), value :: ctx
      real(c_float), value :: a_mid, dt, dt_old, mass_p
      real(c_float), intent(in) :: offset(3), move_back(3)
      type(p3m_step_out), intent(out) :: sout
    end function
    integer(c_int) function p3m_hip_phase_timing(ctx, on) bind(C, name="p3m_hip_phase_timing")
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int32_t), value :: on
    end function
    integer(c_int) function p3m_hip_last_phase_ms(ctx, ms12) bind(C, name="p3m_hip_last_phase_ms")
      import :: c_int, c_float, c_ptr
      type(c_ptr), value :: ctx
      real(c_float), intent(out) :: ms12(12)
    end function
    function p3m_hip_last_error() bind(C, name="p3m_hip_last_error") result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function
  end interface

  type(c_ptr), save :: ctx = c_null_ptr
  logical, save :: device_current = .false.   ! the device holds the particles the host's xv describes
  logical, save :: resident = .false.   ! opt-in: P3M_HIP_RESIDENT=1
  logical :: host_reads
  character(len=8) :: envv
  integer :: envl
  type(p3m_params) :: par
  type(p3m_step_out) :: sout
  real(c_float) :: offset(3), fine_tab(3, 16, 16, 16), coarse_tab(3, 4, 4, 4), rt(3)
  integer(c_int32_t) :: np_c
  integer :: ierr_c, i, j, k, temp(3), fstat
#ifdef MPI_TIME
  real(c_float) :: phase_ms(12)
  ! the reference's tags (timers.f90:68-77) where a phase has one, in the order of p3m_hip_last_phase_ms
  character(len=8), parameter :: phase_tag(12) = (/ 'pos updt', 'linklist', 'par pass', 'fm  mass', 'fm   fft', 'fm  kick', 'pp intra', &
                                                    'pp   ext', 'cm  mass', 'cm force', 'cm   vel', 'del part' /)
#endif

  if (.not. c_associated(ctx)) then
    if (nodes_dim /= 1) stop 'particle_mesh_hip.f90: single-rank adapter (pass a p3m_transport for nodes_dim > 1)'
    par%nodes_dim = nodes_dim; par%tiles_node_dim = tiles_node_dim; par%nf_tile = nf_tile
    par%nf_cutoff = nf_cutoff; par%nf_buf = nf_buf; par%mesh_scale = mesh_scale; par%pp_range = pp_range
    par%cores = cores; par%flags = 0
#ifdef NGP
    par%flags = ior(par%flags, 1)
#endif
#ifdef PPINT
    par%flags = ior(par%flags, 2)
#endif
#ifdef PP_EXT
    par%flags = ior(par%flags, 4)
#endif
#ifdef LRCKCORR
    par%flags = ior(par%flags, 8)
#endif
#ifdef MOVE_GRID_BACK
    par%flags = ior(par%flags, 16)
#endif
#ifdef COARSE_NGP
    par%flags = ior(par%flags, 64)
#endif
    par%rsoft = rsoft; par%pp_bias = pp_bias; par%dt_pp_scale = dt_pp_scale; par%density_buffer = density_buffer
    par%rank = rank; par%device = -1
    ierr_c = p3m_hip_create(par, ctx)
    if (ierr_c /= 0) stop 'p3m_hip_create failed'
    call get_environment_variable('P3M_HIP_RESIDENT', envv, envl)
    if (envl > 0) resident = (envv(1:1) == '1')
    ! the same tables fine_kernel / coarse_kernel read (kernel_initialization.f90:15,344)
    open(unit=18, file=kernel_path//'wfxyzf.3.ascii', status='old', iostat=fstat)
    if (fstat /= 0) stop 'error opening fine mesh kernel'
    do k = 1, 16
      do j = 1, 16
        do i = 1, 16
          read(18, '(3i4,3e16.8)') temp(1), temp(2), temp(3), rt(1), rt(2), rt(3)
          fine_tab(:, i, j, k) = rt
        enddo
      enddo
    enddo
    close(18)
    open(unit=11, file=kernel_path//'wfxyzc.2.ascii', status='old', iostat=fstat)
    if (fstat /= 0) stop 'error opening coarse mesh kernel'
    do k = 1, 4
      do j = 1, 4
        do i = 1, 4
          read(11, '(3i4,3e16.8)') temp(:), coarse_tab(:, i, j, k)
        enddo
      enddo
    enddo
    close(11)
    ierr_c = p3m_hip_set_kernel_tables(ctx, fine_tab, coarse_tab)
    if (ierr_c /= 0) stop 'p3m_hip_set_kernel_tables failed'
#ifdef MPI_TIME
    ierr_c = p3m_hip_phase_timing(ctx, 1_c_int32_t)       ! per-phase GPU times of every step (printed below)
#endif
  endif

  offset = 0.0
#ifdef DISP_MESH
  ! the host keeps the RNG: update_position.f90:56-58
  call random_number(offset)
  offset = (offset - 0.5) * mesh_scale * 4.0 - shake_offset
  shake_offset = shake_offset + offset
#endif

  np_c = np_local
  ierr_c = 0
  if (.not. device_current) ierr_c = p3m_hip_upload_particles(ctx, xv, PID, np_c)
  if (ierr_c == 0) ierr_c = p3m_hip_particle_mesh(ctx, a_mid, dt, dt_old, mass_p, offset, shake_offset, sout)
  np_c = sout%np_local
  ! who reads xv next?  the host, on output steps and when the loop ends (cubepm.f90:171-235); otherwise the next particle_mesh
  host_reads = checkpoint_step .or. projection_step .or. halofind_step .or. final_step .or. nts == max_nts .or. a > 1.0 &
               .or. .not. resident
  if (ierr_c == 0 .and. host_reads) ierr_c = p3m_hip_download_particles(ctx, xv, PID, np_c)
  if (ierr_c /= 0) then
    write(*,*) 'particle_mesh (HIP) failed with code', ierr_c
    stop
  endif
  device_current = .not. host_reads
  np_local = np_c
  dt_f_acc = sout%dt_f_acc; dt_c_acc = sout%dt_c_acc
#ifdef PPINT
  dt_pp_acc = sout%dt_pp_acc
#endif
#ifdef PP_EXT
  dt_pp_ext_acc = sout%dt_pp_ext_acc
#endif
#ifdef MOVE_GRID_BACK
  shake_offset = 0.0
#endif
#ifdef MPI_TIME
  ! what the reference prints phase by phase under -DMPI_TIME, in seconds (timers.f90:68-77; one rank: max = avg = min)
  if (p3m_hip_last_phase_ms(ctx, phase_ms) == 0) then
    do i = 1, 12
      call mpi_time_analyze(phase_tag(i), real(phase_ms(i)) * 1.0e-3, rank, nodes)
    enddo
  endif
#endif
#ifdef DIAG
  if (rank == 0) write(*,*) 'sum of rho_f=', sout%sum_rho_f
  if (rank == 0) write(*,*) 'sum of rho_c=', sout%sum_rho_c
  if (rank == 0) write(*,*) 'total number of particles =', sout%np_total
#endif
end subroutine particle_mesh
