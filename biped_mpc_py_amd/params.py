"""Parameter bags with the reference's names and defaults, and their packing into the C ABI struct.

`MPC` and `Biped` mirror REF:22-32 and REF:34-48 (REF = the reference's bipedalLocomotionMPC.py)
attribute for attribute, so code written against the reference's objects keeps working; any other
object exposing the same attributes (e.g. the reference's own instances) is accepted too.
"""
from __future__ import annotations

import numpy as np

from . import _lib


class MPC:
    """REF:22-32."""

    def __init__(self):
        self.h = 10
        self.dt = 0.04
        self.x_cmd = np.array([0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0], dtype=float)
        self.Q = np.array([500, 100, 100, 300, 300, 700, 1, 1, 1, 1, 1, 1, 1], dtype=float)
        self.R = np.array([1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], dtype=float) * 1e-4
        self.kv = 0.01
        self.kp = np.eye(3) * 500
        self.kd = np.eye(3) * 10
        self.swingHeight = 0.1


class Biped:
    """REF:34-48."""

    def __init__(self):
        self.m = 12
        self.I = np.array([[0.932, 0, 0], [0, 0.9420, 0], [0, 0, 0.0711]])
        self.lt = 0.09
        self.lh = 0.05
        self.g = 9.81
        self.hip_offset = np.array([-0.005, 0.047, -0.126])
        self.mu = 0.5
        self.f_max = np.array([[500], [500], [500]])
        self.f_min = np.array([[0], [0], [0]])
        self.tau_max = np.array([[0], [67], [33.5]])
        self.tau_min = -self.tau_max


# solver knobs (bmpc_default_params): overridable through solver_options
SOLVER_FIELDS = ("rho", "rho_eq_scale", "rho_lo", "rho_hi_f", "rho_hi_m", "kappa", "alpha", "eps_pri", "eps_dua",
                 "max_iter", "check_every", "adapt_start", "adapt_every", "max_refactor", "warm_adapt_start", "path", "penalty_mode", "rescue", "accel",
                 "adapt_early", "adapt_late", "adapt_busy", "adapt_flips", "confirm_from", "kappa_confirm")

# bmpc_params.path (include/bmpc.h enum bmpc_path)
PATH_AUTO, PATH_DENSE, PATH_STAGE = 0, 1, 2
# bmpc_params.rescue (enum bmpc_rescue_mode)
RESCUE_AUTO, RESCUE_OFF, RESCUE_ON = -1, 0, 1


def pack_params(mpc=None, biped=None, half=None, solver_options=None):
    """Build a `bmpc_params` from reference-style objects.  `half` (gait half period of the reference-foot
    generator) defaults to `mpc.half` if that exists, else to what `bmpc_default_params` chose for this
    horizon: the reference's hard-coded 5 at h = 10 (REF:101-105), h / 2 otherwise -- the half period the
    periodic contact table of `get_contact_sequence(t, mpc, half=...)` has for that horizon."""
    mpc = mpc if mpc is not None else MPC()
    biped = biped if biped is not None else Biped()
    lib = _lib.load()
    cp = _lib.CParams()
    h = int(mpc.h)
    _lib.check(lib.bmpc_default_params(cp, h))       # solver defaults; physical fields overwritten below
    cp.h = h
    if half is not None or hasattr(mpc, "half"):
        cp.half = int(half if half is not None else mpc.half)
    cp.dt = float(mpc.dt)
    cp.kv = float(mpc.kv)
    x_cmd = np.asarray(mpc.x_cmd, float).reshape(12)
    Q = np.asarray(mpc.Q, float).reshape(-1)
    R = np.asarray(mpc.R, float).reshape(12)
    if Q.shape[0] not in (12, 13):
        raise ValueError("mpc.Q must have 12 or 13 entries")
    for i in range(12):
        cp.x_cmd[i] = x_cmd[i]
        cp.R[i] = R[i]
    for i in range(13):
        cp.Q[i] = Q[i] if i < Q.shape[0] else 1.0
    cp.m = float(biped.m)
    I = np.asarray(biped.I, float).reshape(9)
    for i in range(9):
        cp.I[i] = I[i]
    cp.lt, cp.lh, cp.g, cp.mu = float(biped.lt), float(biped.lh), float(biped.g), float(biped.mu)
    for name in ("f_max", "f_min", "tau_max", "tau_min"):
        v = np.asarray(getattr(biped, name), float).reshape(3)
        arr = getattr(cp, name)
        for i in range(3):
            arr[i] = v[i]
    for name in ("kp", "kd"):                                   # REF:30-31 (optional on foreign objects)
        if hasattr(mpc, name):
            v = np.asarray(getattr(mpc, name), float).reshape(9)
            arr = getattr(cp, name)
            for i in range(9):
                arr[i] = v[i]
    if hasattr(mpc, "swingHeight"):
        cp.swingHeight = float(mpc.swingHeight)
    if hasattr(biped, "hip_offset"):
        v = np.asarray(biped.hip_offset, float).reshape(3)
        for i in range(3):
            cp.hip_offset[i] = v[i]
    for k, v in (solver_options or {}).items():
        if k not in SOLVER_FIELDS:
            raise KeyError(f"unknown solver option {k!r}")
        setattr(cp, k, type(getattr(cp, k))(v))
    return cp


def params_key(cp):
    """Hashable identity of a parameter block (used to cache solver handles)."""
    import ctypes
    return bytes(ctypes.string_at(ctypes.addressof(cp), ctypes.sizeof(cp)))
