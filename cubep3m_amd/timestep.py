"""Host time loop of the reference (timestep.f90) through the C ABI (include/p3m_hip.h, p3m_hip_timestep /
p3m_hip_expansion) and the main loop of cubepm.f90:103-236 around `particle_mesh`.

    tp = TimeParams(omega_m=0.24, omega_l=0.76, a_checkpoint=[0.05, 1.0])
    ts = TimeState(a=1/201, tau=-3/(1/201)**0.5)
    sim = Simulation(pm, tp, ts, mass_p=8.0)      # pm: ParticleMesh or ParticleMeshGroup
    while sim.step(): pass
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import lib as _lib
from .params import FLAG_PP_EXT as P3M_FLAG_PP_EXT, FLAG_PPINT as P3M_FLAG_PPINT

MAX_INPUT = 100


class P3MTimeParams(C.Structure):
    _fields_ = [("cosmo", C.c_int32), ("restrict_da", C.c_int32), ("omega_m", C.c_float), ("omega_l", C.c_float), ("wde", C.c_float),
                ("dt_scale", C.c_float), ("dt_max", C.c_float), ("ra_max", C.c_float), ("da_max", C.c_float),
                ("num_checkpoints", C.c_int32), ("num_projections", C.c_int32), ("num_halofinds", C.c_int32),
                ("a_checkpoint", C.c_float * MAX_INPUT), ("a_projection", C.c_float * MAX_INPUT), ("a_halofind", C.c_float * MAX_INPUT),
                ("pairwise_ic", C.c_int32), ("pair_infall", C.c_int32), ("shake_test_ic", C.c_int32), ("cur_sep", C.c_float), ("mass_p", C.c_float),
                ("chaplygin", C.c_int32), ("omega_ch", C.c_float), ("A_ch", C.c_float), ("alpha_ch", C.c_float)]


class P3MTimeState(C.Structure):
    _fields_ = [("nts", C.c_int32), ("a", C.c_float), ("a_mid", C.c_float), ("da", C.c_float), ("dt", C.c_float), ("dt_old", C.c_float),
                ("dt_gas", C.c_float), ("tau", C.c_float), ("t", C.c_float), ("cur_checkpoint", C.c_int32), ("cur_projection", C.c_int32),
                ("cur_halofind", C.c_int32), ("checkpoint_step", C.c_int32), ("projection_step", C.c_int32), ("halofind_step", C.c_int32),
                ("final_step", C.c_int32)]


@dataclass
class TimeParams:
    """cubepm.par:15-30 + the `parameters` file + the checkpoint / projection / halofind lists (scale factors)."""
    cosmo: bool = True
    restrict_da: bool = False
    omega_m: float = 0.24
    omega_l: float = 0.76
    wde: float = -1.0
    dt_scale: float = 1.0
    dt_max: float = 1.0
    ra_max: float = 0.01
    da_max: float = 0.01
    a_checkpoint: list = field(default_factory=lambda: [1.0])
    a_projection: list = field(default_factory=list)
    a_halofind: list = field(default_factory=list)
    pad: float = 100.0       # value of the list entries past the end (see include/p3m_hip.h)
    # the non-cosmological test runs of timestep.f90:197-216 (cubepm.par:61-68)
    pairwise_ic: bool = False
    pair_infall: bool = False
    shake_test_ic: bool = False
    cur_sep: float = 1.0
    mass_p: float = 1.0
    # -DChaplygin (timestep.f90:296-339): omega_ch from the parameters file, A_ch / alpha_ch from cubepm.par:21-22
    chaplygin: bool = False
    omega_ch: float = 0.7
    A_ch: float = 1.0
    alpha_ch: float = 0.0

    def to_c(self) -> P3MTimeParams:
        c = P3MTimeParams(int(self.cosmo), int(self.restrict_da), self.omega_m, self.omega_l, self.wde, self.dt_scale, self.dt_max, self.ra_max,
                          self.da_max, len(self.a_checkpoint), len(self.a_projection), len(self.a_halofind))
        for name in ("a_checkpoint", "a_projection", "a_halofind"):
            lst = list(getattr(self, name))
            if len(lst) > MAX_INPUT:
                raise ValueError("%s: more than max_input = %d entries" % (name, MAX_INPUT))
            arr = getattr(c, name)
            for i in range(MAX_INPUT):
                arr[i] = lst[i] if i < len(lst) else self.pad
        c.pairwise_ic, c.pair_infall, c.shake_test_ic = int(self.pairwise_ic), int(self.pair_infall), int(self.shake_test_ic)
        c.cur_sep, c.mass_p = self.cur_sep, self.mass_p
        c.chaplygin, c.omega_ch, c.A_ch, c.alpha_ch = int(self.chaplygin), self.omega_ch, self.A_ch, self.alpha_ch
        return c


def expansion(tp: TimeParams, a0, dt0):
    d1, d2 = C.c_float(), C.c_float()
    c = tp.to_c()
    _lib.load().p3m_hip_expansion(C.byref(c), C.c_float(a0), C.c_float(dt0), C.byref(d1), C.byref(d2))
    return d1.value, d2.value


def new_state(a=1.0 / 201.0, tau=None) -> P3MTimeState:
    s = P3MTimeState()
    s.a = a
    s.tau = float(np.float32(-3.0) / np.sqrt(np.float32(a))) if tau is None else tau
    s.cur_checkpoint = s.cur_projection = s.cur_halofind = 1
    return s


def timestep(tp_c: P3MTimeParams, flags: int, st: P3MTimeState, dt_f_acc, dt_pp_acc, dt_pp_ext_acc, dt_c_acc):
    """subroutine timestep: advances `st` in place."""
    _lib.check(_lib.load().p3m_hip_timestep(C.byref(tp_c), flags, C.byref(st), C.c_float(dt_f_acc), C.c_float(dt_pp_acc),
                                            C.c_float(dt_pp_ext_acc), C.c_float(dt_c_acc)))
    return st


class Simulation:
    """The main loop of cubepm.f90:103-236 without its file output: timestep -> particle_mesh -> on output steps the
    half drift with dt_old = 0 (:196-198) and dt = 0 afterwards (:231); the cur_* counters advance as checkpoint.f90 /
    projection.f90 / halofind.f90 advance them.  `on_output(sim)` is called where the reference writes its files."""

    def __init__(self, pm, tp: TimeParams, st: P3MTimeState, mass_p, max_nts=4000, offset_fn=None, on_output=None):
        self.pm, self.tp, self.tp_c, self.st, self.mass_p = pm, tp, tp.to_c(), st, float(mass_p)
        p = pm.params
        self.flags = (P3M_FLAG_PPINT if p.ppint else 0) | (P3M_FLAG_PP_EXT if p.pp_ext else 0)
        self.limits = (1000.0, 1000.0, 1000.0, 1000.0)   # cubepm.f90 variable_initialize: no limit before the first force
        self.max_nts, self.offset_fn, self.on_output = max_nts, offset_fn, on_output
        self.shake_offset = np.zeros(3, np.float32)
        self.last = None

    def step(self) -> bool:
        """One pass of the loop; False when the reference would exit (:233)."""
        st = self.st
        timestep(self.tp_c, self.flags, st, *self.limits)
        off = None
        if self.offset_fn is not None:          # update_position.f90:56-58 (the host keeps the RNG)
            off = np.asarray(self.offset_fn(), np.float32)
            self.shake_offset = self.shake_offset + off
        out = self.pm.particle_mesh(st.a_mid, st.dt, st.dt_old, self.mass_p, offset=off, move_back=self.shake_offset if off is not None else None)
        self.last = out
        self.limits = (out.dt_f_acc, out.dt_pp_acc, out.dt_pp_ext_acc, out.dt_c_acc)
        if st.checkpoint_step or st.projection_step or st.halofind_step:
            st.dt_old = 0.0
            self._half_drift()
            if self.on_output is not None:
                self.on_output(self)
            if st.checkpoint_step:
                st.cur_checkpoint += 1
            if st.projection_step:
                st.cur_projection += 1
            if st.halofind_step:
                st.cur_halofind += 1
            st.dt = 0.0
        return not (st.nts == self.max_nts or st.final_step or st.a > 1.0)

    def _half_drift(self):
        pm = self.pm
        if hasattr(pm, "update_position"):
            pm.update_position(self.st.dt, 0.0)
        else:
            raise NotImplementedError("output-step half drift needs the phase-level update_position of ParticleMesh")
