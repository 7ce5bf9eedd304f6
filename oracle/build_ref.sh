#!/bin/bash
# TEST INFRASTRUCTURE.  Builds oracle/_ref/<cfg>/libref.so from the reference's own FFT-free
# hot-path sources, compiled WHERE THEY LIE under /root/reference/source_threads (nothing is
# copied; outputs only under oracle/_ref/).  Needs /root/reference, flang (ROCm) and the image's
# MPICH (/opt/conda).  FFTW 2.1.5 is absent, so no file that calls into it is built or emulated
# (fftw2.f90, fftw3ds.f90, coarse_force.f90, kernel_initialization.f90, particle_mesh*.f90).
#
# The reference's sizes are compile-time parameters read from `../parameters` (cubepm.par:3);
# each configuration therefore gets its own directory with a generated `parameters` file (the
# reference's user-config mechanism, template parameters.example) and its own libref.so.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/source_threads
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
MPI_INC=/opt/conda/include
MPI_LIB=/opt/conda/lib
[ -d "$REF" ] || { echo "no /root/reference: keeping prebuilt oracle/_ref"; exit 0; }

SRCS="update_position link_list particle_pass delete_particles move_grid_back mpi_initialization \
      fine_ngp_mass fine_cic_mass fine_cic_mass_buffer coarse_mass coarse_cic_mass \
      coarse_cic_mass_buffer coarse_force_buffer coarse_max_dt coarse_velocity fine_velocity timestep checkpoint particle_initialization projection"

build_cfg () {  # name nodes_dim tiles nf_tile cores density_buffer "cpp flags"
  local name=$1 nd=$2 T=$3 nf=$4 cores=$5 dens=$6 flags=$7
  local D="$HERE/_ref/$name"
  mkdir -p "$D/inc"
  cat > "$D/parameters" <<PAR
character(*), parameter :: ic_path        = '$D/'
character(*), parameter :: scratch_path = '$D/'
character(*), parameter :: output_path    = '$D/'
character(*), parameter :: cubepm_root    = '$D/'
integer(4),   parameter :: nodes_dim      = $nd
integer(4),   parameter :: tiles_node_dim = $T
integer(4),   parameter :: cores = $cores
integer(4),   parameter :: nested_threads = 1
real(4), parameter :: density_buffer = $dens
integer(4),   parameter :: nf_tile        = $nf
real(4),      parameter :: box            = 200.0
real(4),      parameter :: z_i            = 200.0
real(4),      parameter :: omega_l        = 0.76
real(4),      parameter :: omega_m        = 1.0 - omega_l
real(4),      parameter :: omega_b        = 0.04
real(4),      parameter :: omega_ch       = 0.7
real(4),      parameter :: bias           = 1
real(4),      parameter :: power_index    = 2.0
integer(4),   parameter :: nf_cutoff      = 16
integer(4),   parameter :: nf_buf         = nf_cutoff + 8
integer(4),   parameter :: nc             = (nf_tile-2*nf_buf)*tiles_node_dim*nodes_dim
PAR
  local FFLAGS="-O2 -fPIC -fopenmp -cpp -ffree-form -I$REF -I$D/inc -I$MPI_INC -DDIAG $flags"
  local OBJS=""
  for s in $SRCS; do
    $FC $FFLAGS -c "$REF/$s.f90" -o "$D/$s.o" 2> "$D/$s.log" || { cat "$D/$s.log"; exit 1; }
    OBJS="$OBJS $D/$s.o"
  done
  $FC $FFLAGS -c "$HERE/ref_driver.f90" -o "$D/ref_driver.o"
  $FC -shared -fopenmp -o "$D/libref.so" $OBJS "$D/ref_driver.o" -L$MPI_LIB -Wl,-rpath,$MPI_LIB -lmpifort -lmpi
  echo "built $D/libref.so"
  # an MPI host in the reference's language around the HIP library: the drop-in adapter + the reference's own
  # mpi_initialization.o + COMMON blocks (multi-rank builds only; needs libp3m_hip.so at run time)
  if [ "$nd" -gt 1 ] && [ -f "$HERE/../cubep3m_amd/libp3m_hip.so" ]; then
    # the adapter with -DMPI_TIME (the reference's per-phase report, timers.f90:68-77: mpi_time_analyze comes from the reference's own
    # timers.f90, compiled where it lies; the reference's objects themselves stay without the flag)
    $FC $FFLAGS -DPID_FLAG -DMPI_TIME -J "$D" -c "$HERE/../cubep3m_amd/fortran/particle_mesh_hip_mpi.f90" -o "$D/particle_mesh_hip_mpi.o" 2> "$D/adapter.log" || { cat "$D/adapter.log"; exit 1; }
    $FC $FFLAGS -c "$REF/timers.f90" -o "$D/timers.o" 2> "$D/timers.log" || { cat "$D/timers.log"; exit 1; }
    $FC $FFLAGS -c "$HERE/hip_mpi_driver.f90" -o "$D/hip_mpi_driver.o"
    $FC -fopenmp -o "$D/hip_mpi_driver" "$D/hip_mpi_driver.o" "$D/particle_mesh_hip_mpi.o" "$D/mpi_initialization.o" "$D/timers.o" \
        -L"$HERE/../cubep3m_amd" -lp3m_hip -Wl,-rpath,'$ORIGIN/../../../cubep3m_amd' -L$MPI_LIB -Wl,-rpath,$MPI_LIB -lmpifort -lmpi
    echo "built $D/hip_mpi_driver"
  fi
  # the same host around the single-rank adapter (cubep3m_amd/fortran/particle_mesh_hip.f90)
  if [ "$nd" -eq 1 ] && [ -f "$HERE/../cubep3m_amd/libp3m_hip.so" ]; then
    $FC $FFLAGS -c "$HERE/../cubep3m_amd/fortran/particle_mesh_hip.f90" -o "$D/particle_mesh_hip.o" 2> "$D/adapter.log" || { cat "$D/adapter.log"; exit 1; }
    $FC $FFLAGS -c "$HERE/hip_mpi_driver.f90" -o "$D/hip_mpi_driver.o"
    $FC -fopenmp -o "$D/hip_mpi_driver" "$D/hip_mpi_driver.o" "$D/particle_mesh_hip.o" "$D/mpi_initialization.o" \
        -L"$HERE/../cubep3m_amd" -lp3m_hip -Wl,-rpath,'$ORIGIN/../../../cubep3m_amd' -L$MPI_LIB -Wl,-rpath,$MPI_LIB -lmpifort -lmpi
    echo "built $D/hip_mpi_driver (single-rank adapter)"
  fi
}

#          name        nd T nf  cores dens flags
# -DPID_FLAG only on the 1-rank build: with it the reference's "pass -z" posts the PID isend/irecv
# WITHOUT first waiting for the xv isend/irecv (particle_pass.f90:662-671 vs :577-594), so between
# different processes the unpack loop can read a stale recv_buf (observed here with mpiexec -n 8).
# The CI makefile (Makefile_gnu_sfftw2:6) leaves PID_FLAG off.
build_cfg  cfg1_1rank  1  2 80  2     2.0  "-DNGP -DPID_FLAG"
build_cfg  cfg1_8rank  2  2 80  2     2.0  "-DNGP"
# the PP switches change which limits timestep.f90 takes the minimum of (:93-115)
build_cfg  cfg1_pp     1  2 80  2     2.0  "-DNGP -DPPINT -DPP_EXT -DPID_FLAG"
# -DChaplygin: expansion() calls subroutine Chaplygin (timestep.f90:251-252, :296-339); only timestep.o differs.  A plain -DChaplygin
# cannot compile: cpp replaces the subroutine's own name by 1 (call 1(a0,...)); defined as itself the macro only switches the #ifdef
build_cfg  cfg1_chap   1  2 80  2     2.0  "-DNGP -DChaplygin=Chaplygin -DPID_FLAG"
# fine CIC build (no -DNGP): the #else branch of fine_velocity.f90:175-203 (CIC gather + kick)
build_cfg  cfg1_cic    1  2 80  2     2.0  "-DPID_FLAG"
# -DCOARSE_NGP: the three #ifdef branches of coarse_cic_mass.f90:21, coarse_cic_mass_buffer.f90:26, coarse_velocity.f90:146
build_cfg  cfg1_cngp   1  2 80  2     2.0  "-DNGP -DCOARSE_NGP -DPID_FLAG"
# the MPI host of tests/test_gpu_group.py (hip_mpi_driver): 8 ranks, PP switches on, DISP_MESH off
build_cfg  cfg1_8rank_pp 2 2 80 2     2.0  "-DNGP -DPPINT -DPP_EXT"
# the same host with the adapter's -DPENCIL (the reference's Makefile_p3dfft_nested build has no macro of its own; its
# mpi_initialization_p3dfft.f90 needs cubepm_p3dfft_nested.fh in place of cubepm.fh, so the slab file's object is linked)
build_cfg  cfg1_8rank_pencil 2 2 80 2 2.0  "-DNGP -DLRCKCORR -DPENCIL"
# the host of tests/dropin_bench.py: 8 MPI ranks of 256^3 cells / 128^3 particles each (512^3 fine mesh, 256^3 particles),
# the drop-in adapter timed with and without resident particles
build_cfg  cfg2_8rank  2  2 176 1     1.5  "-DNGP"
