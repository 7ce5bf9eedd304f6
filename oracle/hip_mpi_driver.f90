! TEST INFRASTRUCTURE -- an MPI host in the reference's own language that steps the HIP library through the drop-in
! adapter cubep3m_amd/fortran/particle_mesh_hip_mpi.f90: the reference's COMMON blocks (cubepm.fh), the reference's own
! mpi_initialize (mpi_initialization.o, compiled where it lies), `call particle_mesh` exactly as cubepm.f90:143 does.
! Built by oracle/build_ref.sh into oracle/_ref/<cfg>/hip_mpi_driver; run by tests/test_gpu_group.py as
!   mpiexec -n <nodes> hip_mpi_driver <dir>
! <dir>/in<rank>.bin : int32 n, int32 nsteps, real32 a_mid, dt, dt_old, mass_p, real32 xv(6,n), int64 PID(n)   (stream)
! <dir>/out<rank>.bin: int32 n, real32 dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc, real32 xv(6,n), int64 PID(n)
program hip_mpi_driver
  implicit none
  include 'mpif.h'
  include 'cubepm.fh'
  character(len=512) :: dir, fn
  character(len=8) :: rs
  integer(4) :: n, nsteps, s
  integer(8) :: c0, c1, crate
  real(4) :: sc(4)
  real(8) :: tstep, tsum

  call mpi_initialize                     ! mpi_initialization.f90:2 (cartesian ranks, neighbours, slab ranks)
  call get_command_argument(1, dir)
  write(rs, '(i0)') rank
  fn = trim(dir)//'/in'//trim(rs)//'.bin'
  open(unit=31, file=fn, access='stream', form='unformatted', status='old')
  read(31) n, nsteps, sc
  read(31) xv(:, 1:n)
  read(31) PID(1:n)
  close(31)
  np_local = n
  a_mid = sc(1); dt = sc(2); dt_old = sc(3); mass_p = sc(4)
  shake_offset = 0.0
  dt_pp_acc = 1000.0; dt_pp_ext_acc = 1000.0
  ! the flags timestep.f90 would have set (COMMON /lvar/, /ivar/): no output step until the last one, where the host
  ! reads xv back (cubepm.f90:171-235); a < 1 and nts < max_nts throughout
  checkpoint_step = .false.; projection_step = .false.; halofind_step = .false.; final_step = .false.
  a = 0.5; nts = 0
  tsum = 0.d0
  do s = 1, nsteps
    nts = nts + 1
    if (s == nsteps) final_step = .true.
    call mpi_barrier(mpi_comm_world, ierr)
    call system_clock(c0, crate)
    call particle_mesh                    ! cubep3m_amd/fortran/particle_mesh_hip_mpi.f90
    call mpi_barrier(mpi_comm_world, ierr)
    call system_clock(c1)
    tstep = dble(c1 - c0) / dble(crate)
    if (rank == 0) write(*,'(a,i4,a,f10.3,a,i10)') ' hip_mpi_driver step', s, ': ', 1.d3 * tstep, ' ms  np_local =', np_local
    dt_old = dt
  enddo
  fn = trim(dir)//'/out'//trim(rs)//'.bin'
  open(unit=32, file=fn, access='stream', form='unformatted', status='replace')
  write(32) np_local, dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc
  write(32) xv(:, 1:np_local)
  write(32) PID(1:np_local)
  close(32)
  call mpi_finalize(ierr)
end program hip_mpi_driver
