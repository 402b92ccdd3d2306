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


def u0_err(u, u_ref):
    """SURVEY 8(d), "separately for u0": the same metric over row 0 of the controls alone -- the only row the reference
    applies (REF:493: `u0 = controls[0, :]` goes to lowLevelControl)."""
    u = np.asarray(u, float)
    r = np.asarray(u_ref, float)
    u0 = u.reshape(r.shape[0], -1, 12)[:, 0]
    r0 = r.reshape(r.shape[0], -1, 12)[:, 0]
    return np.abs(u0 - r0).max(1) / np.maximum(1.0, np.abs(r0).max(1))


def both_err(u, u_ref):
    """(all-controls metric, u0 metric) per instance."""
    return rel_err(u, u_ref), u0_err(u, u_ref)


def phases(t, dt, h):
    return np.array([int(v // dt) % h for v in np.asarray(t, float).reshape(-1)], np.int32)


from biped_mpc_py_amd.synth import synth_batch  # noqa: E402,F401  (SURVEY 8(d) generator, shared with bench.py)


def wrench_map(r):
    """W (6h x 12h) with b_j = W_j u_j = [tau_j; F_j], u_j = [f1 f2 m1 m2] (REF:174-180); r (h, 2, 3) lever arms."""
    from oracle import bmpc_oracle as orc
    h = r.shape[0]
    W = np.zeros((6 * h, 12 * h))
    for j in range(h):
        Wj = np.zeros((6, 12))
        Wj[0:3, 0:3], Wj[0:3, 3:6] = orc.skew(r[j, 0]), orc.skew(r[j, 1])
        Wj[0:3, 6:9] = Wj[0:3, 9:12] = np.eye(3)
        Wj[3:6, 0:3] = Wj[3:6, 3:6] = np.eye(3)
        W[6 * j:6 * j + 6, 12 * j:12 * j + 12] = Wj
    return W
