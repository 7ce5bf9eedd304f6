"""Deterministic inputs of the host-time-loop parity tests (timestep.f90): shared by the reference runner, the
oracle tests and the C-ABI tests."""
import numpy as np

A_I = np.float32(1.0) / np.float32(201.0)      # z_i = 200 (oracle/build_ref.sh `parameters`)
SCENARIOS = {
    # PM-only build (-DNGP): dt = min(dt_e, dt_f_acc, dt_c_acc) * dt_scale
    "pm": dict(a0=float(A_I), tau0=float(np.float32(-3.0) / np.sqrt(A_I)), steps=400, flags=dict(ppint=False, pp_ext=False),
               a_checkpoint=[0.01, 0.02, 0.05, 0.2, 1.0], a_projection=[0.02, 0.2], a_halofind=[0.05, 1.0]),
    # -DPPINT -DPP_EXT: all four limits
    "pp": dict(a0=0.02, tau0=-21.0, steps=300, flags=dict(ppint=True, pp_ext=True),
               a_checkpoint=[0.025, 0.1, 0.5], a_projection=[0.025, 0.5], a_halofind=[0.1]),
}


def limits_of_step(name, k):
    """dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc of step k (what particle_mesh would have left in COMMON)."""
    x = np.float32(k)
    f = np.float32(0.6) + np.float32(0.5) * np.sin(np.float32(0.37) * x, dtype=np.float32)
    pp = np.float32(0.4) + np.float32(0.35) * np.cos(np.float32(0.91) * x, dtype=np.float32)
    ext = np.float32(0.5) + np.float32(0.45) * np.sin(np.float32(1.7) * x + np.float32(1.0), dtype=np.float32)
    c = np.float32(2.0) + np.float32(1.5) * np.cos(np.float32(0.13) * x, dtype=np.float32)
    return [float(np.float32(v)) for v in (f, pp, ext, c)]


def expansion_grid():
    g = []
    for a0 in (0.004975124, 0.01, 0.05, 0.2, 0.5, 0.9, 1.0):
        for dt0 in (1e-3, 0.01, 0.1, 0.5, 1.0):
            g.append((float(np.float32(a0)), float(np.float32(dt0))))
    return g
