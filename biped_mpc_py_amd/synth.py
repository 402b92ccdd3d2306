"""Synthetic randomised CoM / stance inputs of SURVEY.md 8(d): the workload generator that bench.py, the
tools and the parity tests share (the product never needs it; it lives here so that it ships with the
package).  `np.random.default_rng(seed)`; the draw order is fixed -- the oracle-solved fixtures under
tests/golden/tuning were generated from it."""
from __future__ import annotations

import numpy as np


def synth_batch(B, h, seed, gait="standing", half=None, vx_cmd=False, per_step_mu=False, turn=False):
    """x_fb: euler ~ U(-0.2, 0.2)^3 rad, pos x, y ~ U(-0.5, 0.5), z ~ U(0.45, 0.60), omega ~ U(-0.5, 0.5)^3,
    v ~ U(-0.5, 0.5)^2 x U(-0.2, 0.2); foot_i = (x - 0.0195 + U(-0.05, 0.05), y +- (0.089 + U(-0.03, 0.03)), 0)
    (nominal stance of the reference FK, REF:478-479); x_cmd: REF:26, optionally v_x,cmd ~ U(-0.5, 0.5).
    gait: "standing" (contact = 1, phase 0), "walking" (alternating single support of half period `half`,
    phase ~ U{0..h-1}: generalises REF:52-58) or "mixed" (config 4: standing or any walking phase).
    per_step_mu: mu[k, foot] ~ U(0.3, 0.9) (config 5).
    turn (round 6; drawn after everything else, so the batches of the other options keep their draws): commands that take
    every branch of REF:64-69 -- per Euler angle an angular-rate command x_cmd[6 + i] ~ U(-0.6, 0.6) (70 %; the reference ramps
    the angle over the horizon, so Rot, R_inv and I_w of REF:148-185 differ at every step) or, with the rate exactly zero, an
    attitude set-point x_cmd[i] ~ U(-0.2, 0.2); v_y ~ U(-0.3, 0.3) (60 %, else a lateral position set-point U(-0.3, 0.3)),
    v_z ~ U(-0.15, 0.15) (50 %).  Returns a dict of fp64 / uint8 / int32 arrays."""
    rng = np.random.default_rng(seed)
    x_fb = np.concatenate([
        rng.uniform(-0.2, 0.2, (B, 3)), rng.uniform(-0.5, 0.5, (B, 2)), rng.uniform(0.45, 0.60, (B, 1)),
        rng.uniform(-0.5, 0.5, (B, 3)), rng.uniform(-0.5, 0.5, (B, 2)), rng.uniform(-0.2, 0.2, (B, 1))], 1)
    foot = np.zeros((B, 6))
    for j, sgn in enumerate((1.0, -1.0)):
        foot[:, 3 * j + 0] = x_fb[:, 3] - 0.0195 + rng.uniform(-0.05, 0.05, B)
        foot[:, 3 * j + 1] = x_fb[:, 4] + sgn * (0.089 + rng.uniform(-0.03, 0.03, B))
    half = half or (5 if h == 10 else max(1, h // 2))
    x_cmd = np.tile(np.array([0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0.0]), (B, 1))
    if vx_cmd:
        x_cmd[:, 9] = rng.uniform(-0.5, 0.5, B)
    if gait == "standing":
        phase = np.zeros(B, np.int32)
        contact = np.ones((B, h, 2), np.uint8)
    else:
        leg0 = (np.arange(max(4 * half, 2 * h)) // half) % 2 == 0        # (periodic; 2 h rows serve every phase, odd h included)
        table = np.stack([leg0, ~leg0], 1).astype(np.uint8)
        phase = rng.integers(0, h, B).astype(np.int32)
        contact = np.stack([table[k:k + h] for k in phase])
        if gait == "mixed":                      # config 4: standing or any walking phase
            stand = rng.integers(0, h + 1, B) == 0
            contact[stand] = 1
    mu = rng.uniform(0.3, 0.9, (B, h, 2)) if per_step_mu else None
    if turn:
        rate = rng.random((B, 3)) < 0.7
        x_cmd[:, 6:9] = np.where(rate, rng.uniform(-0.6, 0.6, (B, 3)), 0.0)
        x_cmd[:, 0:3] = np.where(rate, 0.0, rng.uniform(-0.2, 0.2, (B, 3)))
        lat = rng.random(B) < 0.6
        x_cmd[:, 10] = np.where(lat, rng.uniform(-0.3, 0.3, B), 0.0)
        x_cmd[:, 4] = np.where(lat, 0.0, rng.uniform(-0.3, 0.3, B))
        x_cmd[:, 11] = np.where(rng.random(B) < 0.5, rng.uniform(-0.15, 0.15, B), 0.0)
    return dict(x_fb=x_fb, foot=foot, contact=contact, phase=phase, x_cmd=x_cmd, mu=mu, half=half)


# BASELINE.json configs 2-5 as generator arguments (seed = config number - 1)
CONFIGS = {
    2: dict(h=10, gait="standing", seed=1, kw=dict(), batch=4096, label="configs[1]: randomised CoM/stance states, horizon 10, double support"),
    3: dict(h=16, gait="walking", seed=2, kw=dict(vx_cmd=True), batch=4096, label="configs[2]: horizon 16, alternating single support (half period 8), random v_x command"),
    4: dict(h=10, gait="mixed", seed=3, kw=dict(vx_cmd=True), batch=65536, label="configs[3]: horizon 10, mixed gait schedules (standing or any walking phase)"),
    5: dict(h=20, gait="walking", seed=4, kw=dict(vx_cmd=True, per_step_mu=True), batch=65536, label="configs[4]: horizon 20, walking (half period 10), per-step per-foot friction"),
    # beyond BASELINE.json: the long horizons of SURVEY 8(f) row 4 (stage-structured kernels), config-5 style inputs
    6: dict(h=32, gait="walking", seed=5, kw=dict(vx_cmd=True, per_step_mu=True), batch=4096, label="extension: horizon 32, walking (half period 16), per-step per-foot friction"),
    7: dict(h=40, gait="walking", seed=6, kw=dict(vx_cmd=True, per_step_mu=True), batch=4096, label="extension: horizon 40, walking (half period 20), per-step per-foot friction"),
}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the CODE of the kernel sources -- comments and white space stripped, so that
    editing the commentary does not orphan a measurement: stamps measurements that are replayed later
    (profiles/pmc_summary.json -> bench.py roofline.traffic), so that a figure taken with other kernels is refused."""
    import hashlib
    import os
    import re
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    # everything that decides the code object and the parameter block it runs with: the four sources, the C ABI header
    # (DevParams / bmpc_params layout, defaults) and the compile flags of __graft_entry__.build()
    try:
        import __graft_entry__ as _ge
        flags = " ".join(_ge.KERNEL_FLAGS)
    except Exception:                       # (package used outside the repository: the sources alone)
        flags = ""
    hsh.update(flags.encode())
    for path in [os.path.join(here, n) for n in ("bmpc_kernels.hip", "bmpc_stage.hip", "bmpc_capi.hip", "bmpc_lowlevel.hip")] + \
                [os.path.join(root, "include", "bmpc.h")]:
        if not os.path.exists(path):
            continue
        with open(path, "r", encoding="utf-8") as fh:
            text = fh.read()
        # (no string literal of these files holds "//" or "/*": checked by tests/test_host_logic.py)
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        hsh.update(" ".join(text.split()).encode())
    return hsh.hexdigest()[:16]
