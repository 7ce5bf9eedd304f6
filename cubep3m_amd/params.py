"""ctypes mirrors of include/p3m_hip.h and the derived sizes of cubepm.par.

Reference: /root/reference/parameters.example (user parameters) and
source_threads/cubepm.par:170-208 (derived sizes).  Names follow the reference.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

FLAG_NGP = 1 << 0
FLAG_PPINT = 1 << 1
FLAG_PP_EXT = 1 << 2
FLAG_LRCKCORR = 1 << 3
FLAG_MOVE_GRID_BACK = 1 << 4
FLAG_PENCIL = 1 << 5
FLAG_COARSE_NGP = 1 << 6
FLAG_COARSE_ONLY = 1 << 7

P3M_OK, P3M_EINVAL, P3M_ENOMEM, P3M_ECAPACITY, P3M_EDEVICE, P3M_ESTATE, P3M_ECOMM = 0, -1, -2, -3, -4, -5, -6


class P3MParams(C.Structure):
    """struct p3m_params (include/p3m_hip.h)."""

    _fields_ = [
        ("nodes_dim", C.c_int32),
        ("tiles_node_dim", C.c_int32),
        ("nf_tile", C.c_int32),
        ("nf_cutoff", C.c_int32),
        ("nf_buf", C.c_int32),
        ("mesh_scale", C.c_int32),
        ("pp_range", C.c_int32),
        ("cores", C.c_int32),
        ("flags", C.c_uint32),
        ("rsoft", C.c_float),
        ("pp_bias", C.c_float),
        ("dt_pp_scale", C.c_float),
        ("density_buffer", C.c_float),
        ("rank", C.c_int32),
        ("device", C.c_int32),
    ]


class P3MStepOut(C.Structure):
    """struct p3m_step_out (include/p3m_hip.h)."""

    _fields_ = [
        ("dt_f_acc", C.c_float),
        ("dt_pp_acc", C.c_float),
        ("dt_pp_ext_acc", C.c_float),
        ("dt_c_acc", C.c_float),
        ("sum_rho_f", C.c_double),
        ("sum_rho_c", C.c_double),
        ("np_total", C.c_int64),
        ("np_local", C.c_int32),
        ("np_ghost", C.c_int32),
        ("np_deleted", C.c_int32),
        ("f_force_max", C.c_float),
        ("pp_force_max", C.c_float),
        ("pp_ext_force_max", C.c_float),
        ("c_force_max", C.c_float),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


@dataclass
class Params:
    """User parameters of the reference's `parameters` file + cubepm.par switches."""

    nodes_dim: int = 1
    tiles_node_dim: int = 2
    nf_tile: int = 80
    nf_cutoff: int = 16
    nf_buf: int = 24
    mesh_scale: int = 4
    pp_range: int = 2
    cores: int = 1
    ngp: bool = True
    ppint: bool = False
    pp_ext: bool = False
    lrckcorr: bool = False
    move_grid_back: bool = False
    coarse_ngp: bool = False   # -DCOARSE_NGP: whole-weight coarse deposit and gather on cell i2 (coarse_cic_mass.f90:21-24)
    coarse_only: bool = False   # groups: every rank holds its coarse mesh only (stand-alone distributed coarse transform)
    pencil: bool = False   # coarse FFT in 2-D pencils (p3dfft_coarse.f90) instead of slabs (fftw3ds.f90); multi-rank groups only
    rsoft: float = 0.1
    pp_bias: float = 1.0
    dt_pp_scale: float = 0.05
    density_buffer: float = 2.0
    rank: int = 0
    device: int = -1

    # ---- derived, cubepm.par:186-208 -------------------------------------------------
    @property
    def nodes(self):
        return self.nodes_dim ** 3

    @property
    def nf_physical_tile_dim(self):
        return self.nf_tile - 2 * self.nf_buf

    @property
    def nf_physical_node_dim(self):
        return self.nf_physical_tile_dim * self.tiles_node_dim

    @property
    def nf_physical_dim(self):
        return self.nf_physical_node_dim * self.nodes_dim

    @property
    def nc_buf(self):
        return self.nf_buf // self.mesh_scale

    @property
    def nc_tile_dim(self):
        return self.nf_physical_tile_dim // self.mesh_scale

    @property
    def nc_node_dim(self):
        return self.nc_tile_dim * self.tiles_node_dim

    @property
    def nc_dim(self):
        return self.nc_node_dim * self.nodes_dim

    @property
    def nc_slab(self):
        return self.nc_dim // self.nodes

    @property
    def flags(self):
        return (
            (FLAG_NGP if self.ngp else 0)
            | (FLAG_PPINT if self.ppint else 0)
            | (FLAG_PP_EXT if self.pp_ext else 0)
            | (FLAG_LRCKCORR if self.lrckcorr else 0)
            | (FLAG_MOVE_GRID_BACK if self.move_grid_back else 0)
            | (FLAG_PENCIL if self.pencil else 0)
            | (FLAG_COARSE_NGP if self.coarse_ngp else 0)
            | (FLAG_COARSE_ONLY if self.coarse_only else 0)
        )

    def validate(self):
        """Constraints the reference states: parameters.example:25-31, mpi_initialization.f90:26."""
        if self.nf_physical_tile_dim <= 0 or self.nf_physical_tile_dim % self.mesh_scale:
            raise ValueError("nf_tile - 2*nf_buf must be a positive multiple of mesh_scale")
        if self.nf_buf % self.mesh_scale:
            raise ValueError("nf_buf must be a multiple of mesh_scale")
        if self.pencil:
            if self.nc_node_dim % self.nodes_dim:
                raise ValueError("cannot evenly decompose mesh into pencils (nc_pen = nc_node_dim / nodes_dim, cubepm.par:212)")
        elif self.nc_dim % self.nodes:
            raise ValueError("cannot evenly decompose mesh into slabs")
        if self.ppint and not self.ngp:
            raise ValueError("PPINT is only compiled inside the NGP branch (particle_mesh_threaded.f90:260-287)")

    def to_c(self) -> P3MParams:
        self.validate()
        return P3MParams(
            self.nodes_dim, self.tiles_node_dim, self.nf_tile, self.nf_cutoff, self.nf_buf, self.mesh_scale,
            self.pp_range, self.cores, self.flags, self.rsoft, self.pp_bias, self.dt_pp_scale,
            self.density_buffer, self.rank, self.device,
        )
