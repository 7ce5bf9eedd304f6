!! particle_mesh_hip_mpi.f90 -- the same drop-in `subroutine particle_mesh` for MPI builds of the reference
!! (nodes_dim > 1): ONE MPI rank = one cubic sub-volume = one logical rank of a p3m_group = one GPU.
!! The library's exchanges (ghost particles, slab transposes, force halo, dt reductions; what particle_pass.f90,
!! fftw3ds.f90, coarse_force_buffer.f90 and the mpi_reduce/mpi_bcast pairs do in the reference) go through the three
!! host-transport callbacks of include/p3m_hip.h, implemented below with the MPI calls the host already has.
!! (A host that links RCCL instead calls p3m_hip_group_comm_init_rccl with an id broadcast by MPI_Bcast and needs none
!! of the callbacks.)  Particles stay resident on the device between output steps exactly as in particle_mesh_hip.f90 (see
!! there: opt-in, P3M_HIP_RESIDENT=1; the default copies the particles in and out every step).  Link instead of particle_mesh_threaded.o, with -lp3m_hip.  Compile like the reference:
!!   mpif90 -cpp -ffree-form -I<source_threads> -DNGP -DPPINT -DPP_EXT -DDISP_MESH -c particle_mesh_hip_mpi.f90
module p3m_mpi_transport
  use iso_c_binding
  implicit none
  include 'mpif.h'
contains
  !! exchange(user, npeers, peer[], sbuf[], sbytes[], rbuf[], rbytes[]): post every receive and send, wait for all
  integer(c_int) function p3m_exchange(user, npeers, peer, sbuf, sbytes, rbuf, rbytes) bind(C)
    type(c_ptr), value :: user
    integer(c_int32_t), value :: npeers
    integer(c_int32_t), intent(in) :: peer(npeers)
    type(c_ptr), intent(in) :: sbuf(npeers), rbuf(npeers)
    integer(c_int64_t), intent(in) :: sbytes(npeers), rbytes(npeers)
    integer :: req(2 * npeers), nreq, i, ierr
    integer(1), pointer :: sp(:), rp(:)
    nreq = 0
    do i = 1, npeers
      if (rbytes(i) > 0) then
        call c_f_pointer(rbuf(i), rp, [rbytes(i)])
        nreq = nreq + 1
        call mpi_irecv(rp, int(rbytes(i)), mpi_byte, int(peer(i)), 77, mpi_comm_world, req(nreq), ierr)
      endif
    enddo
    do i = 1, npeers
      if (sbytes(i) > 0) then
        call c_f_pointer(sbuf(i), sp, [sbytes(i)])
        nreq = nreq + 1
        call mpi_isend(sp, int(sbytes(i)), mpi_byte, int(peer(i)), 77, mpi_comm_world, req(nreq), ierr)
      endif
    enddo
    call mpi_waitall(nreq, req, mpi_statuses_ignore, ierr)
    p3m_exchange = ierr
  end function
  integer(c_int) function p3m_allreduce_max_f32(user, v, n) bind(C)
    type(c_ptr), value :: user
    integer(c_int32_t), value :: n
    real(c_float) :: v(n)
    integer :: ierr
    call mpi_allreduce(mpi_in_place, v, int(n), mpi_real, mpi_max, mpi_comm_world, ierr)
    p3m_allreduce_max_f32 = ierr
  end function
  integer(c_int) function p3m_allreduce_sum_f64(user, v, n) bind(C)
    type(c_ptr), value :: user
    integer(c_int32_t), value :: n
    real(c_double) :: v(n)
    integer :: ierr
    call mpi_allreduce(mpi_in_place, v, int(n), mpi_double_precision, mpi_sum, mpi_comm_world, ierr)
    p3m_allreduce_sum_f64 = ierr
  end function
end module p3m_mpi_transport

subroutine particle_mesh
  use iso_c_binding
  use p3m_mpi_transport
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
  type, bind(C) :: p3m_transport     ! struct p3m_transport
    type(c_ptr)    :: user
    type(c_funptr) :: exchange, allreduce_max_f32, allreduce_sum_f64
  end type

  interface
    integer(c_int32_t) function p3m_hip_device_count() bind(C, name="p3m_hip_device_count")
      import :: c_int32_t
    end function
    integer(c_int) function p3m_hip_group_create(params, proc, nprocs, g) bind(C, name="p3m_hip_group_create")
      import :: c_int, c_int32_t, c_ptr, p3m_params
      type(p3m_params), intent(in) :: params
      integer(c_int32_t), value :: proc, nprocs
      type(c_ptr), intent(out) :: g
    end function
    integer(c_int) function p3m_hip_group_set_transport(g, t) bind(C, name="p3m_hip_group_set_transport")
      import :: c_int, c_ptr, p3m_transport
      type(c_ptr), value :: g
      type(p3m_transport), intent(in) :: t
    end function
    integer(c_int) function p3m_hip_group_set_kernel_tables(g, fine, coarse) bind(C, name="p3m_hip_group_set_kernel_tables")
      import :: c_int, c_ptr, c_float
      type(c_ptr), value :: g
      real(c_float), intent(in) :: fine(*), coarse(*)
    end function
    integer(c_int) function p3m_hip_group_upload_particles(g, i, xv6, pid, n) bind(C, name="p3m_hip_group_upload_particles")
      import :: c_int, c_ptr, c_float, c_int64_t, c_int32_t
      type(c_ptr), value :: g
      integer(c_int32_t), value :: i, n
      real(c_float), intent(in) :: xv6(6, *)
      integer(c_int64_t), intent(in) :: pid(*)
    end function
    integer(c_int) function p3m_hip_group_download_particles(g, i, xv6, pid, n) bind(C, name="p3m_hip_group_download_particles")
      import :: c_int, c_ptr, c_float, c_int64_t, c_int32_t
      type(c_ptr), value :: g
      integer(c_int32_t), value :: i
      real(c_float), intent(out) :: xv6(6, *)
      integer(c_int64_t), intent(out) :: pid(*)
      integer(c_int32_t), intent(out) :: n
    end function
    integer(c_int) function p3m_hip_group_particle_mesh(g, a_mid, dt, dt_old, mass_p, offset, move_back, sout) &
        bind(C, name="p3m_hip_group_particle_mesh")
      import :: c_int, c_ptr, c_float, p3m_step_out
      type(c_ptr), value :: g
      real(c_float), value :: a_mid, dt, dt_old, mass_p
      real(c_float), intent(in) :: offset(3), move_back(3)
      type(p3m_step_out), intent(out) :: sout
    end function
  end interface

  type(c_ptr), save :: grp = c_null_ptr
  logical, save :: device_current = .false.   ! the device holds the particles the host's xv describes
  logical, save :: resident = .false.   ! opt-in: P3M_HIP_RESIDENT=1
  logical :: host_reads
  character(len=8) :: envv
  integer :: envl
  type(p3m_params) :: par
  type(p3m_transport) :: tr
  type(p3m_step_out) :: sout
  real(c_float) :: offset(3), fine_tab(3, 16, 16, 16), coarse_tab(3, 4, 4, 4), rt(3)
  integer(c_int32_t) :: np_c
  integer :: ierr_c, i, j, k, temp(3), fstat

  if (.not. c_associated(grp)) then
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
#ifdef PENCIL
    ! the build that links p3dfft_coarse.f90 instead of fftw3ds.f90 (Makefile_p3dfft_nested): no macro of its own in the reference
    par%flags = ior(par%flags, 32)
#endif
    par%rsoft = rsoft; par%pp_bias = pp_bias; par%dt_pp_scale = dt_pp_scale; par%density_buffer = density_buffer
    par%rank = 0                       ! the group assigns the logical ranks
    par%device = mod(rank, max(1, p3m_hip_device_count()))   ! one GPU per MPI rank of the node (ranks share GPUs if there are fewer)
    ierr_c = p3m_hip_group_create(par, int(rank, c_int32_t), int(nodes, c_int32_t), grp)   ! process `rank` of `nodes`: one logical rank each
    if (ierr_c /= 0) stop 'p3m_hip_group_create failed'
    call get_environment_variable('P3M_HIP_RESIDENT', envv, envl)
    if (envl > 0) resident = (envv(1:1) == '1')
    tr%user = c_null_ptr
    tr%exchange = c_funloc(p3m_exchange)
    tr%allreduce_max_f32 = c_funloc(p3m_allreduce_max_f32)
    tr%allreduce_sum_f64 = c_funloc(p3m_allreduce_sum_f64)
    ierr_c = p3m_hip_group_set_transport(grp, tr)
    if (ierr_c /= 0) stop 'p3m_hip_group_set_transport failed'
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
    ierr_c = p3m_hip_group_set_kernel_tables(grp, fine_tab, coarse_tab)      ! collective: builds the distributed coarse kernel
    if (ierr_c /= 0) stop 'p3m_hip_group_set_kernel_tables failed'
  endif

  offset = 0.0
#ifdef DISP_MESH
  ! the host keeps the RNG: update_position.f90:56-61 (rank 0 draws, everybody gets the same offset)
  if (rank == 0) then
    call random_number(offset)
    offset = (offset - 0.5) * mesh_scale * 4.0 - shake_offset
    shake_offset = shake_offset + offset
  endif
  call mpi_bcast(offset, 3, mpi_real, 0, mpi_comm_world, ierr)
  call mpi_bcast(shake_offset, 3, mpi_real, 0, mpi_comm_world, ierr)
#endif

  np_c = np_local
  ierr_c = 0
  if (.not. device_current) ierr_c = p3m_hip_group_upload_particles(grp, 0_c_int32_t, xv, PID, np_c)
  if (ierr_c == 0) ierr_c = p3m_hip_group_particle_mesh(grp, a_mid, dt, dt_old, mass_p, offset, shake_offset, sout)   ! collective
  np_c = sout%np_local
  ! the same flags on every rank (timestep.f90:228-235 broadcasts them): all ranks download, or none
  host_reads = checkpoint_step .or. projection_step .or. halofind_step .or. final_step .or. nts == max_nts .or. a > 1.0 &
               .or. .not. resident
  if (ierr_c == 0 .and. host_reads) ierr_c = p3m_hip_group_download_particles(grp, 0_c_int32_t, xv, PID, np_c)
  if (ierr_c /= 0) then
    write(*,*) 'particle_mesh (HIP) failed with code', ierr_c, ' on rank', rank
    call mpi_abort(mpi_comm_world, ierr, ierr)
  endif
  device_current = .not. host_reads
  np_local = np_c
  dt_f_acc = sout%dt_f_acc; dt_c_acc = sout%dt_c_acc       ! already reduced over all ranks
#ifdef PPINT
  dt_pp_acc = sout%dt_pp_acc
#endif
#ifdef PP_EXT
  dt_pp_ext_acc = sout%dt_pp_ext_acc
#endif
#ifdef MOVE_GRID_BACK
  shake_offset = 0.0
#endif
#ifdef DIAG
  if (rank == 0) write(*,*) 'sum of rho_f=', sout%sum_rho_f
  if (rank == 0) write(*,*) 'sum of rho_c=', sout%sum_rho_c
  if (rank == 0) write(*,*) 'total number of particles =', sout%np_total
#endif
end subroutine particle_mesh
