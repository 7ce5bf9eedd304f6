!! particle_mesh_hip_mpi.f90 -- the same drop-in `subroutine particle_mesh` for MPI builds of the reference
!! (nodes_dim > 1): ONE MPI rank = one cubic sub-volume = one logical rank of a p3m_group = one GPU.
!! The library's exchanges (ghost particles, slab transposes, force halo, dt reductions; what particle_pass.f90:69-722,
!! fftw3ds.f90:24-39,84-99, coarse_force_buffer.f90:25-63 and the mpi_reduce/mpi_bcast pairs do in the reference) travel
!!   * over RCCL / xGMI, device buffer to device buffer, when every MPI rank of a node drives a GPU of its own: rank 0 obtains
!!     the 128-byte id (p3m_hip_rccl_unique_id), MPI broadcasts it, every rank calls p3m_hip_group_comm_init_rccl -- MPI is then
!!     used for nothing else on this path;
!!   * through the three host-transport callbacks of include/p3m_hip.h, implemented below with the MPI calls the host already
!!     has (pinned host staging), when ranks share a GPU (more ranks on a node than GPUs: RCCL refuses two ranks of one
!!     communicator on one device) or when P3M_HIP_TRANSPORT=mpi asks for it.  P3M_HIP_TRANSPORT=rccl insists on RCCL.
!! Particles stay resident on the device between output steps exactly as in particle_mesh_hip.f90 (see
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
    integer(c_int) function p3m_hip_rccl_unique_id(id) bind(C, name="p3m_hip_rccl_unique_id")
      import :: c_int, c_int8_t
      integer(c_int8_t), intent(out) :: id(128)
    end function
    integer(c_int) function p3m_hip_group_comm_init_rccl(g, id, force_local) bind(C, name="p3m_hip_group_comm_init_rccl")
      import :: c_int, c_int8_t, c_int32_t, c_ptr
      type(c_ptr), value :: g
      integer(c_int8_t), intent(in) :: id(128)
      integer(c_int32_t), value :: force_local
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
    integer(c_int) function p3m_hip_group_phase_timing(g, on) bind(C, name="p3m_hip_group_phase_timing")
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: g
      integer(c_int32_t), value :: on
    end function
    integer(c_int) function p3m_hip_group_last_phase_ms(g, ms12) bind(C, name="p3m_hip_group_last_phase_ms")
      import :: c_int, c_float, c_ptr
      type(c_ptr), value :: g
      real(c_float), intent(out) :: ms12(12)
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
  integer :: node_comm, node_rank, node_size, ndev, want_rccl, all_rccl, failed, any_failed
  integer(c_int8_t) :: nccl_id(128)
  character(len=8) :: trv
  integer :: trl
#ifdef MPI_TIME
  real(c_float) :: phase_ms(12)
  ! the reference's tags (timers.f90:68-77; link_list.f90:143, particle_pass.f90:767, coarse_mesh.f90, delete_particles.f90) where a phase has
  ! one, in the order of p3m_hip_group_last_phase_ms
  character(len=8), parameter :: phase_tag(12) = (/ 'pos updt', 'linklist', 'par pass', 'fm  mass', 'fm   fft', 'fm  kick', 'pp intra', &
                                                    'pp   ext', 'cm  mass', 'cm force', 'cm   vel', 'del part' /)
#endif

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
    ! one GPU per MPI rank of the node (mpi_initialization.f90:55-76 places the ranks; which of them share a node is MPI's
    ! knowledge): the ranks of a node count themselves and take the node's GPUs in turn (they share GPUs if there are fewer)
    call mpi_comm_split_type(mpi_comm_world, mpi_comm_type_shared, 0, mpi_info_null, node_comm, ierr)
    call mpi_comm_rank(node_comm, node_rank, ierr)
    call mpi_comm_size(node_comm, node_size, ierr)
    call mpi_comm_free(node_comm, ierr)
    ndev = max(1, p3m_hip_device_count())
    par%device = mod(node_rank, ndev)
    ierr_c = p3m_hip_group_create(par, int(rank, c_int32_t), int(nodes, c_int32_t), grp)   ! process `rank` of `nodes`: one logical rank each
    if (ierr_c /= 0) stop 'p3m_hip_group_create failed'
    call get_environment_variable('P3M_HIP_RESIDENT', envv, envl)
    if (envl > 0) resident = (envv(1:1) == '1')
    ! the transport: RCCL when EVERY rank has a GPU of its own (all ranks take the same decision), else the MPI callbacks
    want_rccl = 0
    if (node_size <= ndev) want_rccl = 1
    call get_environment_variable('P3M_HIP_TRANSPORT', trv, trl)
    if (trl >= 3) then
      if (trv(1:3) == 'mpi') want_rccl = 0
      if (trl >= 4) then
        if (trv(1:4) == 'rccl') want_rccl = 1
      endif
    endif
    call mpi_allreduce(want_rccl, all_rccl, 1, mpi_integer, mpi_min, mpi_comm_world, ierr)
    if (all_rccl == 1) then
      nccl_id = 0
      failed = 0
      if (rank == 0) then
        if (p3m_hip_rccl_unique_id(nccl_id) /= 0) failed = 1
      endif
      ! everybody learns whether rank 0 got an id BEFORE anybody enters ncclCommInitRank (an all-zero id would show up as a hang there)
      call mpi_bcast(failed, 1, mpi_integer, 0, mpi_comm_world, ierr)
      if (failed /= 0) then
        if (rank == 0) write(*,*) 'particle_mesh (HIP): no RCCL id (P3M_HIP_TRANSPORT=mpi selects the MPI transport)'
        call mpi_abort(mpi_comm_world, 1, ierr)
      endif
      call mpi_bcast(nccl_id, 128, mpi_byte, 0, mpi_comm_world, ierr)
      if (p3m_hip_group_comm_init_rccl(grp, nccl_id, 0_c_int32_t) /= 0) failed = 1
      call mpi_allreduce(failed, any_failed, 1, mpi_integer, mpi_max, mpi_comm_world, ierr)
      if (any_failed /= 0) then
        if (rank == 0) write(*,*) 'particle_mesh (HIP): the RCCL communicator could not be set up (P3M_HIP_TRANSPORT=mpi selects the MPI transport)'
        call mpi_abort(mpi_comm_world, 1, ierr)
      endif
      if (rank == 0) write(*,*) 'particle_mesh (HIP): exchanges over RCCL,', nodes, 'ranks,', ndev, 'GPUs per node'
    else
      tr%user = c_null_ptr
      tr%exchange = c_funloc(p3m_exchange)
      tr%allreduce_max_f32 = c_funloc(p3m_allreduce_max_f32)
      tr%allreduce_sum_f64 = c_funloc(p3m_allreduce_sum_f64)
      ierr_c = p3m_hip_group_set_transport(grp, tr)
      if (ierr_c /= 0) stop 'p3m_hip_group_set_transport failed'
      if (rank == 0) write(*,*) 'particle_mesh (HIP): exchanges through MPI (host staging),', node_size, 'ranks on', ndev, 'GPUs per node'
    endif
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
#ifdef MPI_TIME
    ierr_c = p3m_hip_group_phase_timing(grp, 1_c_int32_t)                    ! per-phase GPU times of every step (printed below)
#endif
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
    call mpi_abort(mpi_comm_world, 1, ierr)
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
#ifdef MPI_TIME
  ! what the reference prints phase by phase under -DMPI_TIME: "tag : max avg min" over the ranks, in seconds (timers.f90:68-77)
  ! (mpi_time_analyze is collective: every rank calls it twelve times whatever its own return code was -- zeros on failure)
  if (p3m_hip_group_last_phase_ms(grp, phase_ms) /= 0) phase_ms = 0.0
  do i = 1, 12
    call mpi_time_analyze(phase_tag(i), real(phase_ms(i)) * 1.0e-3, rank, nodes)
  enddo
#endif
#ifdef DIAG
  if (rank == 0) write(*,*) 'sum of rho_f=', sout%sum_rho_f
  if (rank == 0) write(*,*) 'sum of rho_c=', sout%sum_rho_c
  if (rank == 0) write(*,*) 'total number of particles =', sout%np_total
#endif
end subroutine particle_mesh
