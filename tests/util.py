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


def synth_batch(B, h, seed, gait="standing", half=None, vx_cmd=False, per_step_mu=False):
    """SURVEY 8(d) synthetic generator (same distribution as oracle/gen_golden.synth_state)."""
    rng = np.random.default_rng(seed)
    x_fb = np.concatenate([
        rng.uniform(-0.2, 0.2, (B, 3)), rng.uniform(-0.5, 0.5, (B, 2)), rng.uniform(0.45, 0.60, (B, 1)),
        rng.uniform(-0.5, 0.5, (B, 3)), rng.uniform(-0.5, 0.5, (B, 2)), rng.uniform(-0.2, 0.2, (B, 1))], 1)
    foot = np.zeros((B, 6))
    for j, sgn in enumerate((1.0, -1.0)):
        foot[:, 3 * j + 0] = x_fb[:, 3] - 0.0195 + rng.uniform(-0.05, 0.05, B)
        foot[:, 3 * j + 1] = x_fb[:, 4] + sgn * (0.089 + rng.uniform(-0.03, 0.03, B))
    half = half or (5 if h == 10 else h // 2)
    x_cmd = np.tile(np.array([0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0.0]), (B, 1))
    if vx_cmd:
        x_cmd[:, 9] = rng.uniform(-0.5, 0.5, B)
    if gait == "standing":
        phase = np.zeros(B, np.int32)
        contact = np.ones((B, h, 2), np.uint8)
    else:
        leg0 = (np.arange(4 * half) // half) % 2 == 0
        table = np.stack([leg0, ~leg0], 1).astype(np.uint8)
        phase = rng.integers(0, h, B).astype(np.int32)
        contact = np.stack([table[k:k + h] for k in phase])
        if gait == "mixed":                      # config 4: standing or any walking phase
            stand = rng.integers(0, h + 1, B) == 0
            contact[stand] = 1
    mu = rng.uniform(0.3, 0.9, (B, h, 2)) if per_step_mu else None
    return dict(x_fb=x_fb, foot=foot, contact=contact, phase=phase, x_cmd=x_cmd, mu=mu, half=half)
