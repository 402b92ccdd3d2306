"""Shared helpers for the parity tests."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# fixture name -> (h, half)
BATCH_FIXTURES = {
    "cfg2_standing_h10": (10, 5),
    "cfg4_walking_h10": (10, 5),
    "edge_cases_h10": (10, 5),
    "cfg3_trot_h16": (16, 8),
    "cfg5_mu_h20": (20, 10),
}

# north_star tolerance: <= 1e-4 relative force error vs the reference optimum (fp64 -> fp32)
REL_TOL = 1e-4


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(u, u_ref):
    """||u - u_ref||_inf / max(1, ||u_ref||_inf) per instance over all h x 12 controls."""
    u = np.asarray(u, float).reshape(u_ref.shape[0], -1)
    r = np.asarray(u_ref, float).reshape(u_ref.shape[0], -1)
    return np.abs(u - r).max(1) / np.maximum(1.0, np.abs(r).max(1))


def phases(t, dt, h):
    return np.array([int(v // dt) % h for v in np.asarray(t, float).reshape(-1)], np.int32)


from biped_mpc_py_amd.synth import synth_batch  # noqa: E402,F401  (SURVEY 8(d) generator, shared with bench.py)
