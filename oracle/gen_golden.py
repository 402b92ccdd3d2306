#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container; TEST INFRASTRUCTURE).

Imports the reference module from /root/reference BY PATH (nothing is copied), with a capture
object registered as `cvxopt` -- the one third-party import the image lacks (REF:3).  The capture
object does no QP arithmetic of its own that is taken as "reference": it records the six
matrices the reference hands to `cvxopt.solvers.qp` at REF:297 and returns the minimiser computed
by oracle/bmpc_oracle.solve_qp so the reference's own post-processing (REF:300-304) and its
low-level controller (REF:444-470) run on it.  What the fixtures therefore pin:

  * PINNED BY THE REFERENCE ITSELF: x_ref, foot_ref, contact tables, A_k/B_k, P, q, G, h, A, b,
    FK foot positions, and tau = lowLevelControl(..., u0) for a given u0.
  * PINNED BY KKT CERTIFICATE ONLY (cvxopt absent => "parity unpinned" at the solver boundary):
    the optimum z* = [X*; U*], stored with its certificate residuals against the captured matrices.

Extensions the reference cannot produce unpatched (h != 10 walking, per-step mu) are
generated from the oracle restatement alone and flagged `extension=1` in the fixture.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import bmpc_oracle as orc  # noqa: E402

REF_PATH = "/root/reference/bipedalLocomotionMPC.py"
OUT = os.path.join(ROOT, "tests", "golden")


class _Capture:
    """Stand-in for the absent `cvxopt` module: records the REF:297 call."""

    def __init__(self):
        self.last = None
        mod = types.ModuleType("cvxopt")
        mod.matrix = lambda a, *args, **kw: np.array(a, dtype=float)
        mod.solvers = types.SimpleNamespace(qp=self._qp, options={})
        self.module = mod

    def _qp(self, P, q, G=None, h=None, A=None, b=None, **kw):
        nx = A.shape[0]
        z, lam, nu, info = orc.solve_qp(P, q, G, h, A, b, nx)
        self.last = dict(P=np.array(P), q=np.array(q).reshape(-1), G=np.array(G),
                         h=np.array(h).reshape(-1), A=np.array(A), b=np.array(b).reshape(-1),
                         z=z, lam=lam, nu=nu, info=info)
        return {"x": z.reshape(-1, 1), "status": "optimal"}


def load_reference():
    cap = _Capture()
    sys.modules["cvxopt"] = cap.module
    saved = np.get_printoptions()
    spec = importlib.util.spec_from_file_location("ref_bipedalLocomotionMPC", REF_PATH)
    ref = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(ref)           # runs the demo at REF:475-495
    np.set_printoptions(**saved)               # undo REF:4's global side effect
    return ref, cap


def sparse_triplets(M):
    r, c = np.nonzero(M)
    return np.stack([r, c]).astype(np.int32), M[r, c]


def synth_state(rng):
    """SURVEY 8(d) synthetic generator (one instance)."""
    x_fb = np.concatenate([
        rng.uniform(-0.2, 0.2, 3),
        rng.uniform(-0.5, 0.5, 2), rng.uniform(0.45, 0.60, 1),
        rng.uniform(-0.5, 0.5, 3),
        rng.uniform(-0.5, 0.5, 2), rng.uniform(-0.2, 0.2, 1)])
    foot = np.zeros(6)
    for j, sgn in enumerate((1.0, -1.0)):
        foot[3 * j + 0] = x_fb[3] - 0.0195 + rng.uniform(-0.05, 0.05)
        foot[3 * j + 1] = x_fb[4] + sgn * (0.089 + rng.uniform(-0.03, 0.03))
    return x_fb, foot


def run_reference_case(ref, cap, x_fb, t, foot, contact, x_cmd=None, bounds=None):
    """One call of the reference's own solve_mpc (REF:187) with capture; returns fixture dict.
    bounds: optional dict of Biped fields f_max / f_min / tau_max / tau_min (REF:45-48) as 3-vectors."""
    mpc, biped = ref.MPC(), ref.Biped()
    if x_cmd is not None:
        mpc.x_cmd = np.array(x_cmd, float)
    for name, v in (bounds or {}).items():
        setattr(biped, name, np.array(v, float).reshape(3, 1))          # the reference's own 3 x 1 column layout
    with contextlib.redirect_stdout(io.StringIO()):
        states, controls = ref.solve_mpc(np.array(x_fb, float), t, np.array(foot, float), mpc, biped,
                                         np.array(contact))
        x_ref = ref.get_reference_trajectory(np.array(x_fb, float), mpc)
        foot_ref = ref.get_reference_foot_trajectory(np.array(x_fb, float), t, np.array(foot, float),
                                                     mpc, np.array(contact))
        AB = [ref.get_simplified_dynamics(mpc, biped, x_ref[:, k], foot_ref[:, k])
              for k in range(mpc.h)]
    c = cap.last
    fx = dict(x_fb=np.array(x_fb, float), t=float(t), foot=np.array(foot, float),
              contact=np.array(contact).astype(np.int8), x_cmd=np.array(mpc.x_cmd, float),
              hor=np.int32(mpc.h), extension=np.int32(0),
              x_ref=x_ref, foot_ref=foot_ref,
              A_k=np.stack([a for a, _ in AB]), B_k=np.stack([b for _, b in AB]),
              q=c["q"], h=c["h"], b=c["b"],
              states=states, controls=controls, lam=c["lam"], nu=c["nu"],
              objective=np.float64(c["info"]["objective"]),
              n_pinned=np.int32(c["info"]["n_pinned"]), n_active=np.int32(c["info"]["n_active"]),
              polished=np.int32(c["info"]["polished"]),
              kkt=np.array([c["info"]["kkt"][k] for k in
                            ("stationarity", "primal_eq", "primal_ineq", "dual", "complementarity")]))
    for name in ("f_max", "f_min", "tau_max", "tau_min"):
        fx[name] = np.array(getattr(biped, name), float).reshape(3)
    for name in ("P", "G", "A"):
        idx, val = sparse_triplets(c[name])
        fx[name + "_idx"], fx[name + "_val"] = idx, val
        fx[name + "_shape"] = np.array(c[name].shape, np.int32)
    return fx


def run_extension_case(x_fb, t, foot, contact, h, half, x_cmd, mu_steps):
    """Oracle-only fixture for configs the reference cannot produce unpatched."""
    mpc, biped = orc.MPC(), orc.Biped()
    mpc.h = h
    if x_cmd is not None:
        mpc.x_cmd = np.array(x_cmd, float)
    states, controls, info = orc.solve_mpc(x_fb, t, foot, mpc, biped, contact, half=half,
                                           mu_steps=mu_steps, return_info=True)
    return dict(x_fb=np.array(x_fb, float), t=float(t), foot=np.array(foot, float),
                contact=np.array(contact).astype(np.int8), x_cmd=np.array(mpc.x_cmd, float),
                hor=np.int32(h), half=np.int32(half), extension=np.int32(1),
                mu_steps=(np.zeros((0, 2)) if mu_steps is None else np.array(mu_steps, float)),
                states=states, controls=controls, objective=np.float64(info["objective"]),
                n_pinned=np.int32(info["n_pinned"]), n_active=np.int32(info["n_active"]),
                polished=np.int32(info["polished"]),
                kkt=np.array([info["kkt"][k] for k in
                              ("stationarity", "primal_eq", "primal_ineq", "dual", "complementarity")]))


def stack_batch(cases, keys):
    return {k: np.stack([c[k] for c in cases]) for k in keys}


REGIME_KEYS = ("x_fb", "t", "foot", "contact", "x_cmd", "x_ref", "foot_ref", "A_k", "B_k", "q", "h",
               "b", "states", "controls", "objective", "n_pinned", "n_active", "polished", "kkt",
               "f_max", "f_min", "tau_max", "tau_min")


def draw_command(rng, full=True):
    """A command vector that takes every branch of REF:64-69: per coordinate i < 6 either a rate x_cmd[i + 6] != 0 (the
    reference ramps x_fb[i] + rate k dt: Euler ramps make Rot, R_inv, I_w differ at EVERY step of the horizon) or, with the rate
    exactly zero, a set-point x_cmd[i]."""
    x_cmd = np.array([0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0], float)
    for i in range(3):                                   # attitude: angular-rate command or set-point
        if rng.random() < 0.7:
            x_cmd[6 + i] = rng.uniform(-0.6, 0.6)
        else:
            x_cmd[i] = rng.uniform(-0.2, 0.2)
    if full:
        for i, (vr, pr) in enumerate(((0.5, 0.3), (0.3, 0.3), (0.15, 0.0))):     # v_x, v_y, v_z commands or position set-points
            if rng.random() < 0.6:
                x_cmd[9 + i] = rng.uniform(-vr, vr)
            elif pr:
                x_cmd[3 + i] = rng.uniform(-pr, pr)
    return x_cmd


def gen_regimes(ref, cap):
    """regimes the unpatched reference produces and no earlier fixture held -- generated by the
    reference's OWN solve_mpc (REF:187; extension = 0).
      cfg_cmd_h10    80 instances: commanded angular rates / attitude set-points / lateral and vertical velocity commands
                     (REF:64-69), standing + every walking phase: the per-step linearisation REF:148-185 at x_ref[:, k] sees a
                     different Rot, R_inv and I_w at every step.
      cfg_bounds_h10 48 instances: Biped bounds off their defaults (REF:45-48): f_min < 0 on the horizontal axes (a real friction
                     pyramid), tau_max[0] != 0 (m_x un-pinned), asymmetric tau_min; commands as above on half of them."""
    mpc0 = ref.MPC()
    rng = np.random.default_rng(600)
    cases = []
    for i in range(80):
        x_fb, foot = synth_state(rng)
        if i < 20:                                      # standing
            t, contact = 0.0, np.ones((10, 2))
        else:                                           # every walking phase six times
            k = (i - 20) % 10
            t = k * mpc0.dt + 0.5 * mpc0.dt
            contact = ref.get_contact_sequence(t, mpc0)
        cases.append(run_reference_case(ref, cap, x_fb, t, foot, contact, x_cmd=draw_command(rng)))
    np.savez_compressed(os.path.join(OUT, "cfg_cmd_h10.npz"), **stack_batch(cases, REGIME_KEYS))
    print("cfg_cmd max kkt", np.max([c["kkt"] for c in cases], axis=0), "polished", all(c["polished"] for c in cases))

    rng = np.random.default_rng(601)
    variants = (
        dict(f_min=[-500, -500, 0]),
        dict(tau_max=[20, 67, 33.5], tau_min=[-20, -67, -33.5]),
        dict(tau_min=[0, -30, -10]),                                            # asymmetric moment box (m_x still pinned)
        dict(f_min=[-500, -500, 0], tau_max=[20, 67, 33.5], tau_min=[-12, -40, -33.5]),
    )
    cases = []
    for i in range(48):
        x_fb, foot = synth_state(rng)
        if i % 3 == 0:
            t, contact = 0.0, np.ones((10, 2))
        else:
            k = int(rng.integers(0, 10))
            t = k * mpc0.dt + 0.5 * mpc0.dt
            contact = ref.get_contact_sequence(t, mpc0)
        x_cmd = draw_command(rng) if i % 2 else np.array(mpc0.x_cmd, float)
        cases.append(run_reference_case(ref, cap, x_fb, t, foot, contact, x_cmd=x_cmd, bounds=variants[i % 4]))
    np.savez_compressed(os.path.join(OUT, "cfg_bounds_h10.npz"), **stack_batch(cases, REGIME_KEYS))
    print("cfg_bounds max kkt", np.max([c["kkt"] for c in cases], axis=0), "polished", all(c["polished"] for c in cases),
          "n_pinned", sorted(set(int(c["n_pinned"]) for c in cases)))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref, cap = load_reference()
    if "--regimes" in sys.argv:                          # only the round-6 regime batches (the other fixtures regenerate bit for bit)
        gen_regimes(ref, cap)
        return
    mpc0, biped0 = ref.MPC(), ref.Biped()

    # ---- 1. the two known-answer cases, with full matrices and the consumer's tau -------------
    x_fb0 = np.array([0, 0, 0, 0, 0, 0.53, 0, 0, 0, 0, 0, 0], float)       # REF:13
    q0 = np.array([0, 0, -np.pi / 4, np.pi / 2, -np.pi / 4] * 2)           # REF:15
    qd0 = np.zeros(10)
    pf_w = ref.getFootPositionWorld(x_fb0, q0, biped0)                     # REF:478
    foot0 = pf_w.reshape(-1)
    for name, contact in (("standing", np.ones((10, 2))),                  # REF:483-484
                          ("walking_t0", ref.get_contact_sequence(0, mpc0))):  # REF:481-482
        fx = run_reference_case(ref, cap, x_fb0, 0.0, foot0, contact)
        u0 = fx["controls"][0, :].reshape(-1, 1)                           # REF:493
        fx["q_joint"], fx["qd_joint"], fx["pf_w"] = q0, qd0, pf_w
        fx["tau"] = ref.lowLevelControl(x_fb0, 0.0, pf_w, q0, qd0, mpc0, biped0,
                                        np.array(contact), u0)
        np.savez_compressed(os.path.join(OUT, f"known_{name}.npz"), **fx)
        print(name, "u0 =", np.round(fx["controls"][0], 6), "obj", fx["objective"], "kkt", fx["kkt"])

    # ---- 2. unit fixtures of the reference's small functions ---------------------------------
    rng = np.random.default_rng(100)
    eul = rng.uniform(-0.6, 0.6, (16, 3))
    vec = rng.normal(size=(16, 3))
    qj = rng.uniform(-1.0, 1.0, (16, 10))
    xs = np.stack([synth_state(rng)[0] for _ in range(16)])
    unit = dict(
        eul=eul, eul2rotm=np.stack([ref.eul2rotm(e) for e in eul]),
        vec=vec, skew=np.stack([ref.skew(v) for v in vec]),
        t_list=np.array([0.0, 0.039, 0.04, 0.12, 0.2, 0.36, 0.4, 0.55, 0.799, 1.0]),
        qj=qj, xs=xs,
        fk=np.stack([ref.getFootPositionWorld(x, q, biped0).reshape(-1) for x, q in zip(xs, qj)]),
        Jm=np.stack([np.stack([ref.getLegKinematics(*q[5 * l:5 * l + 5], 1 - 2 * l)[0]
                               for l in range(2)]) for q in qj]))
    unit["contact_seq"] = np.stack([ref.get_contact_sequence(t, mpc0) for t in unit["t_list"]])
    np.savez_compressed(os.path.join(OUT, "unit_functions.npz"), **unit)

    # ---- 3. config 2: random standing batch through the reference assembly --------------------
    keys = ("x_fb", "t", "foot", "contact", "x_cmd", "x_ref", "foot_ref", "A_k", "B_k", "q", "h",
            "b", "states", "controls", "objective", "n_pinned", "n_active", "polished", "kkt")
    rng = np.random.default_rng(1)          # seed 1 = config 2 (seeds 0..4)
    cases = []
    for _ in range(64):
        x_fb, foot = synth_state(rng)
        cases.append(run_reference_case(ref, cap, x_fb, 0.0, foot, np.ones((10, 2))))
    np.savez_compressed(os.path.join(OUT, "cfg2_standing_h10.npz"), **stack_batch(cases, keys))
    print("cfg2 max kkt", np.max([c["kkt"] for c in cases], axis=0))

    # ---- 4. all ten walking phases + random walking states (config 4 mix) ---------------------
    rng = np.random.default_rng(3)
    cases = []
    for k in range(10):                                 # default state, every phase
        t = k * mpc0.dt + 1e-9
        cases.append(run_reference_case(ref, cap, x_fb0, t, foot0, ref.get_contact_sequence(t, mpc0)))
    for i in range(54):
        x_fb, foot = synth_state(rng)
        k = int(rng.integers(0, 10))
        t = k * mpc0.dt + 0.5 * mpc0.dt
        x_cmd = np.array(mpc0.x_cmd, float)
        if i % 2:
            x_cmd[9] = rng.uniform(-0.5, 0.5)           # v_x command != 0 branch of REF:66-67
        cases.append(run_reference_case(ref, cap, x_fb, t, foot,
                                        ref.get_contact_sequence(t, mpc0), x_cmd=x_cmd))
    np.savez_compressed(os.path.join(OUT, "cfg4_walking_h10.npz"), **stack_batch(cases, keys))
    print("cfg4 max kkt", np.max([c["kkt"] for c in cases], axis=0))

    # ---- 5. edge cases through the reference ---------------------------------------------------
    cases = []
    for pitch in (-0.5, 0.5):
        x = x_fb0.copy(); x[1] = pitch
        cases.append(run_reference_case(ref, cap, x, 0.0, foot0, np.ones((10, 2))))
    cases.append(run_reference_case(ref, cap, x_fb0, 0.0, foot0, np.zeros((10, 2))))   # flight
    x = x_fb0.copy(); x[5] = 0.30; x[11] = -1.5                                        # force cap
    cases.append(run_reference_case(ref, cap, x, 0.0, foot0, ref.get_contact_sequence(0, mpc0)))
    x = x_fb0.copy(); x[0:3] = (0.4, -0.3, 0.6); x[6:9] = (1.0, -1.0, 0.5)
    cases.append(run_reference_case(ref, cap, x, 0.2, foot0, ref.get_contact_sequence(0.2, mpc0)))
    c = np.ones((10, 2)); c[3:6, 0] = 0; c[7:, 1] = 0                                  # ragged schedule
    cases.append(run_reference_case(ref, cap, x_fb0, 0.0, foot0, c))
    np.savez_compressed(os.path.join(OUT, "edge_cases_h10.npz"), **stack_batch(cases, keys))
    print("edge max kkt", np.max([c["kkt"] for c in cases], axis=0))

    # ---- 6. extensions (oracle only): config 3 (h=16 trot) and config 5 (h=20, per-step mu) ---
    ekeys = ("x_fb", "t", "foot", "contact", "x_cmd", "hor", "half", "mu_steps", "states",
             "controls", "objective", "n_pinned", "n_active", "polished", "kkt")
    for name, seed, h, use_mu, n in (("cfg3_trot_h16", 2, 16, False, 24), ("cfg5_mu_h20", 4, 20, True, 16)):
        rng = np.random.default_rng(seed)
        cases = []
        mp = orc.MPC(); mp.h = h
        for _ in range(n):
            x_fb, foot = synth_state(rng)
            k = int(rng.integers(0, h))
            t = k * mp.dt + 0.5 * mp.dt
            x_cmd = np.array(mp.x_cmd, float); x_cmd[9] = rng.uniform(-0.5, 0.5)
            contact = orc.get_contact_sequence(t, mp, half=h // 2)
            mu = rng.uniform(0.3, 0.9, (h, 2)) if use_mu else None
            cases.append(run_extension_case(x_fb, t, foot, contact, h, h // 2, x_cmd, mu))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **stack_batch(cases, ekeys))
        print(name, "max kkt", np.max([c["kkt"] for c in cases], axis=0))

    gen_regimes(ref, cap)


if __name__ == "__main__":
    main()
