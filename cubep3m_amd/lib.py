"""ctypes binding of cubep3m_amd/libp3m_hip.so (include/p3m_hip.h).

There is no CPU fallback: if the HIP library is missing this module raises at first use.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .params import P3MParams, P3MStepOut

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("P3M_HIP_LIB") or os.path.join(_HERE, "libp3m_hip.so")  # override: timing-only ablation builds
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")

# every entry point include/p3m_hip.h declares
EXPORTS = [
    "p3m_hip_create", "p3m_hip_destroy", "p3m_hip_last_error", "p3m_hip_device_count", "p3m_hip_derived",
    "p3m_hip_rccl_unique_id", "p3m_hip_set_kernel_tables", "p3m_hip_set_kernels_raw",
    "p3m_hip_get_kernels", "p3m_hip_upload_particles", "p3m_hip_download_particles", "p3m_hip_particle_mesh",
    "p3m_hip_update_position", "p3m_hip_link_list_and_pass", "p3m_hip_fine_mesh", "p3m_hip_coarse_mesh",
    "p3m_hip_delete_particles", "p3m_hip_get_step_out", "p3m_hip_probe_tile_density", "p3m_hip_probe_tile_force",
    "p3m_hip_probe_coarse", "p3m_hip_fft3d", "p3m_hip_time_fine_sweep", "p3m_hip_time_fine_gather", "p3m_hip_time_fft_pass", "p3m_hip_time_pp", "p3m_hip_stream", "particle_mesh_hip_",
    "p3m_hip_group_create", "p3m_hip_group_destroy", "p3m_hip_group_comm_init_rccl", "p3m_hip_group_set_transport", "p3m_hip_expansion", "p3m_hip_timestep", "p3m_hip_write_checkpoint", "p3m_hip_read_checkpoint", "p3m_hip_write_pid_checkpoint",
    "p3m_hip_read_pid_checkpoint", "p3m_hip_write_ic", "p3m_hip_read_ic", "p3m_hip_group_nlocal", "p3m_hip_group_local_rank",
    "p3m_hip_group_ctx", "p3m_hip_group_set_kernel_tables", "p3m_hip_group_upload_particles", "p3m_hip_group_download_particles",
    "p3m_hip_group_particle_mesh", "p3m_hip_group_update_position", "p3m_hip_phase_timing", "p3m_hip_last_phase_ms", "p3m_hip_group_phase_timing", "p3m_hip_group_last_phase_ms", "p3m_hip_group_probe_coarse",
    "p3m_hip_group_set_coarse_density", "p3m_hip_group_coarse_transform", "p3m_hip_group_get_coarse_hat", "p3m_hip_group_get_coarse_force",
    "p3m_hip_group_coarse_exchange_bytes", "p3m_hip_group_comm_info", "p3m_hip_group_set_kernels_raw",
    "p3m_hip_projection", "p3m_hip_group_projection", "p3m_hip_coarse_power", "p3m_hip_group_coarse_power", "p3m_hip_coarse_fft_schedule", "p3m_hip_write_power", "p3m_hip_write_projection", "p3m_hip_read_projection",
]


class P3MError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"p3m_hip error {code}: {msg}")
        self.code = code


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"] + ([] if verbose else ["-s"]))
    return SO_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: build it with `make -C cubep3m_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(SO_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.p3m_hip_create.argtypes = [C.POINTER(P3MParams), C.POINTER(vp)]
    L.p3m_hip_destroy.argtypes = [vp]
    L.p3m_hip_destroy.restype = None
    L.p3m_hip_last_error.restype = C.c_char_p
    L.p3m_hip_derived.argtypes = [vp, i32]
    L.p3m_hip_derived.restype = C.c_int64
    L.p3m_hip_set_kernel_tables.argtypes = [vp, f32p, f32p]
    L.p3m_hip_set_kernels_raw.argtypes = [vp, f32p, f32p]
    L.p3m_hip_get_kernels.argtypes = [vp, vp, vp]
    L.p3m_hip_upload_particles.argtypes = [vp, vp, vp, i32]
    L.p3m_hip_download_particles.argtypes = [vp, vp, vp, C.POINTER(i32)]
    L.p3m_hip_particle_mesh.argtypes = [vp, f32, f32, f32, f32, vp, vp, C.POINTER(P3MStepOut)]
    L.p3m_hip_update_position.argtypes = [vp, f32, f32, vp]
    L.p3m_hip_link_list_and_pass.argtypes = [vp]
    L.p3m_hip_fine_mesh.argtypes = [vp, f32, f32, f32]
    L.p3m_hip_coarse_mesh.argtypes = [vp, f32, f32, f32]
    L.p3m_hip_delete_particles.argtypes = [vp, vp]
    L.p3m_hip_get_step_out.argtypes = [vp, f32, C.POINTER(P3MStepOut)]
    L.p3m_hip_probe_tile_density.argtypes = [vp, i32, i32, i32, f32, f32p]
    L.p3m_hip_probe_tile_force.argtypes = [vp, f32p, f32p, C.POINTER(f32)]
    L.p3m_hip_probe_coarse.argtypes = [vp, f32, vp, vp]
    L.p3m_hip_fft3d.argtypes = [vp, f32p, i32, i32]
    L.p3m_hip_time_fine_sweep.argtypes = [vp, f32, i32, C.POINTER(f32)]
    L.p3m_hip_time_fine_gather.argtypes = [vp, i32, C.POINTER(f32)]
    L.p3m_hip_time_fft_pass.argtypes = [vp, i32, i32, C.POINTER(f32), C.POINTER(i32)]
    L.p3m_hip_coarse_power.argtypes = [vp, f32, f32, f32p]
    L.p3m_hip_group_coarse_power.argtypes = [vp, f32, f32, f32p]
    L.p3m_hip_coarse_fft_schedule.argtypes = [i32, C.c_uint32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.p3m_hip_coarse_fft_schedule.restype = i32
    L.p3m_hip_write_power.argtypes = [C.c_char_p, f32p, i32]
    L.p3m_hip_time_pp.argtypes = [vp, f32, f32, f32, i32, C.POINTER(f32), C.POINTER(f32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.p3m_hip_rccl_unique_id.argtypes = [vp]
    L.p3m_hip_group_create.argtypes = [C.POINTER(P3MParams), i32, i32, C.POINTER(vp)]
    L.p3m_hip_group_destroy.argtypes = [vp]
    L.p3m_hip_group_destroy.restype = None
    L.p3m_hip_group_comm_init_rccl.argtypes = [vp, vp, i32]
    L.p3m_hip_group_set_transport.argtypes = [vp, vp]
    L.p3m_hip_expansion.argtypes = [vp, f32, f32, C.POINTER(f32), C.POINTER(f32)]
    L.p3m_hip_expansion.restype = None
    L.p3m_hip_timestep.argtypes = [vp, C.c_uint32, vp, f32, f32, f32, f32]
    L.p3m_hip_write_checkpoint.argtypes = [C.c_char_p, vp, vp, vp, i32, i32]
    L.p3m_hip_read_checkpoint.argtypes = [C.c_char_p, vp, vp, C.c_int64, i32, i32]
    L.p3m_hip_write_pid_checkpoint.argtypes = [C.c_char_p, vp, vp, i32, i32]
    L.p3m_hip_read_pid_checkpoint.argtypes = [C.c_char_p, vp, vp, C.c_int64, i32, i32]
    L.p3m_hip_write_ic.argtypes = [C.c_char_p, vp, i32, i32]
    L.p3m_hip_projection.argtypes = [vp, f32, vp, vp, vp, C.POINTER(C.c_double)]
    L.p3m_hip_group_projection.argtypes = [vp, f32, vp, vp, vp, C.POINTER(C.c_double)]
    L.p3m_hip_write_projection.argtypes = [C.c_char_p, f32, vp, i32, i32]
    L.p3m_hip_read_projection.argtypes = [C.c_char_p, C.POINTER(f32), vp, i32, i32]
    L.p3m_hip_read_ic.argtypes = [C.c_char_p, vp, C.c_int64, C.POINTER(i32), i32]
    L.p3m_hip_group_nlocal.argtypes = [vp]
    L.p3m_hip_group_local_rank.argtypes = [vp, i32]
    L.p3m_hip_group_ctx.argtypes = [vp, i32]
    L.p3m_hip_group_ctx.restype = vp
    L.p3m_hip_group_set_kernel_tables.argtypes = [vp, f32p, f32p]
    L.p3m_hip_group_upload_particles.argtypes = [vp, i32, vp, vp, i32]
    L.p3m_hip_group_download_particles.argtypes = [vp, i32, vp, vp, C.POINTER(i32)]
    L.p3m_hip_group_particle_mesh.argtypes = [vp, f32, f32, f32, f32, vp, vp, C.POINTER(P3MStepOut)]
    L.p3m_hip_group_update_position.argtypes = [vp, f32, f32, vp]
    L.p3m_hip_group_probe_coarse.argtypes = [vp, f32, i32, vp, vp]
    L.p3m_hip_group_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_char_p]
    L.p3m_hip_group_set_kernels_raw.argtypes = [vp, f32p, C.POINTER(vp)]
    L.p3m_hip_group_set_coarse_density.argtypes = [vp, i32, f32p]
    L.p3m_hip_group_coarse_transform.argtypes = [vp, i32, i32, C.POINTER(f32)]
    L.p3m_hip_group_get_coarse_hat.argtypes = [vp, i32, vp, C.c_int64]
    L.p3m_hip_group_get_coarse_force.argtypes = [vp, i32, vp]
    L.p3m_hip_group_coarse_exchange_bytes.argtypes = [vp]
    L.p3m_hip_group_coarse_exchange_bytes.restype = C.c_int64
    L.p3m_hip_phase_timing.argtypes = [vp, i32]
    L.p3m_hip_last_phase_ms.argtypes = [vp, f32p]
    L.p3m_hip_group_phase_timing.argtypes = [vp, i32]
    L.p3m_hip_group_last_phase_ms.argtypes = [vp, f32p]
    L.p3m_hip_stream.argtypes = [vp]
    L.p3m_hip_stream.restype = vp
    _lib = L
    return L


def check(code):
    if code != 0:
        raise P3MError(code, load().p3m_hip_last_error().decode(errors="replace"))
