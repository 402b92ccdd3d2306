"""CPU oracle for the HECTOR force-and-moment MPC hot path (TEST INFRASTRUCTURE ONLY).

This file is a clean-room fp64 NumPy restatement of what
/root/reference/bipedalLocomotionMPC.py (REF) computes on its hot path, plus an fp64 QP
solver with a KKT certificate.  It is the *checker*: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product (biped_mpc_py_amd) never does.

Parity status
-------------
* Everything up to the solver boundary (x_ref, foot_ref, A_k, B_k and the six QP matrices
  P, q, G, h, A, b handed to cvxopt at REF:297) is PINNED: tests/golden/*.npz holds those
  matrices as captured from the reference itself running in the build container
  (oracle/gen_golden.py), and tests/test_oracle_golden.py asserts this restatement reproduces
  them to 1e-12.
* The solver arithmetic itself lives in the third-party dependency `cvxopt` (version not pinned
  by the reference: no requirements/lock file; only `import cvxopt`, REF:3) which is absent from
  the image: PARITY UNPINNED AT THE SOLVER BOUNDARY.  Mitigation: the QP is strictly convex
  (P = 2 diag(Q.., R..) > 0, REF:27-28, 278-281) and always feasible (U = 0), so its minimiser is
  unique and solver independent; `solve_qp` returns that minimiser with a KKT certificate
  (stationarity / primal / dual / complementarity residuals) evaluated against the *captured
  reference matrices*.  That certificate is the ONLY meaningful parity target: cvxopt's default
  tolerances (reltol 1e-6 on an objective of ~ -2000) leave its own output free to differ from the
  minimiser by newtons along the soft directions (curvature 2R = 2e-4), so "what the reference would
  print" is not a tighter target than the certified optimum; tests/ re-derive the certificate from the
  stored primal point alone (`certificate_from_primal`) for every fixture.

Generalisations beyond the reference (used by BASELINE configs 3-5) are explicit opt-ins:
`half` (gait half period, reference hard-codes 5: REF:52-58, 101-105) and `mu_steps`
(per-step, per-foot friction, reference has one scalar: REF:44, 220-229).  With the defaults
every function reduces to the reference's behaviour, quirks included.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "MPC", "Biped", "get_contact_sequence", "get_reference_trajectory",
    "get_reference_foot_trajectory", "eul2rotm", "skew", "get_simplified_dynamics",
    "build_sparse_qp", "condense", "build_condensed_qp", "solve_qp", "kkt_residuals", "certificate_from_primal",
    "solve_mpc", "lowLevelControl", "getFootPositionWorld",
]


# --------------------------------------------------------------------------------------
# parameter bags (REF:22-48) -- same attribute names and defaults
# --------------------------------------------------------------------------------------
class MPC:
    """REF:22-32."""

    def __init__(self):
        self.h = 10
        self.dt = 0.04
        self.x_cmd = np.array([0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0], dtype=float)
        self.Q = np.array([500, 100, 100, 300, 300, 700, 1, 1, 1, 1, 1, 1, 1], dtype=float)
        self.R = np.ones(12) * 1e-4
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
        self.f_max = np.array([[500.0], [500.0], [500.0]])
        self.f_min = np.array([[0.0], [0.0], [0.0]])
        self.tau_max = np.array([[0.0], [67.0], [33.5]])
        self.tau_min = -self.tau_max


# --------------------------------------------------------------------------------------
# gait / reference generation (REF:50-109)
# --------------------------------------------------------------------------------------
def phase_index(t, mpc):
    """k = int(t // dt) % h  (REF:56-57, 99-100; floating floor division is part of the spec)."""
    return int(t // mpc.dt) % mpc.h


def get_contact_sequence(t, mpc, half=None):
    """REF:50-59.  Reference: 20x2 table of 5-on/5-off, rows k..k+9 (quirk: 10 rows whatever h is).

    With `half` given (extension) the table has 4*half rows of
    half-on/half-off and the slice is k:k+h; half=5, h=10 is the reference.
    """
    if half is None:
        half_, nrow = 5, 10
    else:
        half_, nrow = int(half), mpc.h
    # (periodic: 4 half rows are two periods, which serve every phase while h <= 2 half; shorter periods get as many rows as k + h needs)
    leg0 = (np.arange(max(4 * half_, 2 * nrow)) // half_) % 2 == 0
    table = np.stack([leg0, ~leg0], axis=1).astype(int)
    k = phase_index(t, mpc)
    return table[k:k + nrow, :]


def get_reference_trajectory(x_fb, mpc):
    """REF:61-70."""
    h = mpc.h
    x_ref = np.tile(np.append(np.asarray(mpc.x_cmd, float), 1.0), (h, 1)).T.copy()
    x_ref[:12, 0] = x_fb
    for i in range(6):
        for k in range(1, h):
            if mpc.x_cmd[i + 6] != 0:
                x_ref[i, k] = x_fb[i] + mpc.x_cmd[i + 6] * (k * mpc.dt)
            else:
                x_ref[i, k] = mpc.x_cmd[i]
    return x_ref


def get_reference_foot_trajectory(x_fb, t, foot, mpc, contact, half=None):
    """REF:72-109, including the x_fb[10] quirk in foot_des_y_2 (REF:87).

    half=None reproduces the hard-coded 5 (REF:101-105); an explicit half generalises the tile
    counts to (half-kk, half, kk).
    """
    h, dt = mpc.h, mpc.dt
    fx1 = x_fb[3] + x_fb[9] * 1 / 2 * h / 2 * dt + mpc.kv * (x_fb[3] - mpc.x_cmd[3])
    fx2 = x_fb[3] + x_fb[9] * 1 / 2 * h * dt + mpc.kv * (x_fb[3] - mpc.x_cmd[3])
    fy1 = x_fb[4] + x_fb[10] * 1 / 2 * h / 2 * dt + mpc.kv * (x_fb[4] - mpc.x_cmd[4])
    fy2 = x_fb[10] + x_fb[10] * 1 / 2 * h * dt + mpc.kv * (x_fb[4] - mpc.x_cmd[4])
    foot_1 = np.array([fx1, fy1, 0.0, fx1, fy1, 0.0]).reshape(-1, 1)
    foot_2 = np.array([fx2, fy2, 0.0, fx2, fy2, 0.0]).reshape(-1, 1)
    foot = np.asarray(foot, float).reshape(-1, 1)
    hp = 5 if half is None else int(half)
    k = phase_index(t, mpc)
    kk = k % hp
    if np.sum(contact[0, :]) == 1:
        out = np.concatenate(
            (np.tile(foot, (1, hp - kk)), np.tile(foot_1, (1, hp)), np.tile(foot_2, (1, kk))), axis=1)
        if half is not None and out.shape[1] != h:
            # (extension: a horizon that is not two half periods -- odd h, or a half period of the caller's choice -- keeps the
            #  second touch-down point to the end of the horizon, resp. stops at the horizon: what the kernels' column rule
            #  j < half - kk / j < 2 half - kk / else gives.  With half = h / 2 nothing changes.)
            out = np.concatenate((out, np.tile(foot_2, (1, max(h - out.shape[1], 0)))), axis=1)[:, :h]
        return out
    return np.tile(foot, (1, h))


# --------------------------------------------------------------------------------------
# SRBM linearisation (REF:111-185)
# --------------------------------------------------------------------------------------
def eul2rotm(eul):
    """REF:111-138: Rz(eul[2]) @ Ry(eul[1]) @ Rx(eul[0])."""
    cr, cp, cy = np.cos(eul)
    sr, sp, sy = np.sin(eul)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def skew(v):
    """REF:140-146."""
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _rot_zyx_extrinsic(yaw, pitch, roll):
    """scipy Rotation.from_euler('zyx', [yaw, pitch, roll]).as_matrix() (REF:154-156).

    Lower-case axes = extrinsic rotations: first about z by yaw, then about fixed y by pitch, then
    about fixed x by roll, i.e. Rx(roll) @ Ry(pitch) @ Rz(yaw) (checked against SciPy in
    tests/test_oracle_golden.py).
    """
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rx @ Ry @ Rz


def get_simplified_dynamics(mpc, biped, x_ref, foot_ref):
    """REF:148-185: forward-Euler SRBM  A = I + Ac dt (13x13), B = Bc dt (13x12)."""
    roll, pitch, yaw = x_ref[2], x_ref[1], x_ref[0]
    Rot = _rot_zyx_extrinsic(yaw, pitch, roll)
    Iw = Rot.T @ np.asarray(biped.I, float) @ Rot
    R_inv = np.linalg.inv(np.array([
        [np.cos(yaw) * np.cos(pitch), -np.sin(yaw), 0],
        [np.sin(yaw) * np.cos(pitch), np.cos(yaw), 0],
        [-np.sin(pitch), 0, 1]]))
    Ac = np.zeros((13, 13))
    Ac[0:3, 6:9] = R_inv
    Ac[3:6, 9:12] = np.eye(3)
    Ac[11, 12] = -biped.g
    Iw_inv = np.linalg.inv(Iw)
    Bc = np.zeros((13, 12))
    Bc[6:9, 0:3] = Iw_inv @ skew(-x_ref[3:6] + foot_ref[0:3])
    Bc[6:9, 3:6] = Iw_inv @ skew(-x_ref[3:6] + foot_ref[3:6])
    Bc[6:9, 6:9] = Iw_inv
    Bc[6:9, 9:12] = Iw_inv
    Bc[9:12, 0:3] = np.eye(3) / biped.m
    Bc[9:12, 3:6] = np.eye(3) / biped.m
    return Ac * mpc.dt + np.eye(13), Bc * mpc.dt


# --------------------------------------------------------------------------------------
# QP assembly, sparse (reference) form: z = [X(13h); U(12h)]  (REF:187-286)
# --------------------------------------------------------------------------------------
def _mu_table(biped, h, mu_steps):
    if mu_steps is None:
        return np.full((h, 2), float(biped.mu))
    mu_steps = np.asarray(mu_steps, float)
    assert mu_steps.shape == (h, 2)
    return mu_steps


def build_sparse_qp(x_fb, t, foot, mpc, biped, contact, half=None, mu_steps=None):
    """Returns dict with P, q, G, h, A, b exactly as handed to cvxopt at REF:297, plus
    x_ref, foot_ref, A_list, B_list.  """
    h = mpc.h
    x_fb = np.asarray(x_fb, float)
    contact = np.asarray(contact)
    x_ref = get_reference_trajectory(x_fb, mpc)
    foot_ref = get_reference_foot_trajectory(x_fb, t, foot, mpc, contact, half=half)
    R = eul2rotm(x_fb[0:3])                                             # REF:193
    A_list, B_list = [], []
    for k in range(h):                                                  # REF:197-200
        A, B = get_simplified_dynamics(mpc, biped, x_ref[:, k], foot_ref[:, k])
        A_list.append(A)
        B_list.append(B)

    Aeq = np.zeros((13 * h, 25 * h))                                    # REF:203-216
    beq = np.zeros(13 * h)
    x0 = np.append(x_fb, 1.0)
    beq[0:13] = A_list[0] @ x0
    for i in range(h):
        Aeq[13 * i:13 * (i + 1), 13 * i:13 * (i + 1)] = np.eye(13)
        Aeq[13 * i:13 * (i + 1), 13 * h + 12 * i:13 * h + 12 * (i + 1)] = -B_list[i]
        if i > 0:
            Aeq[13 * i:13 * (i + 1), 13 * (i - 1):13 * i] = -A_list[i]

    mu = _mu_table(biped, h, mu_steps)
    A_mu = np.zeros((8 * h, 25 * h))                                    # REF:220-232
    for k in range(h):
        for j in range(2):
            c0 = 13 * h + 12 * k + 3 * j
            r0 = 8 * k + 4 * j
            for r, (ax, sg) in enumerate(((0, 1.0), (1, 1.0), (0, -1.0), (1, -1.0))):
                A_mu[r0 + r, c0 + ax] = sg
                A_mu[r0 + r, c0 + 2] = -mu[k, j]
    b_mu = np.zeros((8 * h, 1))

    A_f = np.zeros((24 * h, 25 * h))                                    # REF:235-251
    b_f = np.zeros((24 * h, 1))
    f_max = np.asarray(biped.f_max, float).reshape(3)
    f_min = np.asarray(biped.f_min, float).reshape(3)
    t_max = np.asarray(biped.tau_max, float).reshape(3)
    t_min = np.asarray(biped.tau_min, float).reshape(3)
    for k in range(h):
        A_f[24 * k:24 * k + 12, 13 * h + 12 * k:13 * h + 12 * (k + 1)] = np.eye(12)
        A_f[24 * k + 12:24 * k + 24, 13 * h + 12 * k:13 * h + 12 * (k + 1)] = -np.eye(12)
        c0, c1 = float(contact[k, 0]), float(contact[k, 1])
        b_f[24 * k:24 * (k + 1), 0] = np.concatenate([
            c0 * f_max, c1 * f_max, c0 * t_max, c1 * t_max,
            c0 * -f_min, c1 * -f_min, c0 * -t_min, c1 * -t_min])

    lt = biped.lt - 0.01                                                # REF:254-271
    lh = biped.lh - 0.02
    ez = np.array([0, 0, 1.0]) @ R.T
    ey = np.array([0, 1.0, 0]) @ R.T
    z3 = np.zeros(3)
    A_LF1 = np.vstack([
        np.hstack([-lh * ez, z3, ey, z3]),
        np.hstack([-lt * ez, z3, -ey, z3]),
        np.hstack([z3, -lh * ez, z3, ey]),
        np.hstack([z3, -lt * ez, z3, -ey])])
    A_LF = np.hstack([np.zeros((4 * h, 13 * h)), np.kron(np.eye(h), A_LF1)])
    b_LF = np.zeros((4 * h, 1))

    G = np.vstack([A_mu, A_f, A_LF])                                    # REF:273-274
    hvec = np.vstack([b_mu, b_f, b_LF])

    Qbar = np.kron(np.eye(h), np.diag(np.asarray(mpc.Q, float)))        # REF:278-286
    Rbar = np.kron(np.eye(h), np.diag(np.asarray(mpc.R, float)))
    P = 2 * np.block([[Qbar, np.zeros((13 * h, 12 * h))], [np.zeros((12 * h, 13 * h)), Rbar]])
    q = 2 * np.hstack([-Qbar @ x_ref.T.flatten(), np.zeros(12 * h)])
    return dict(P=P, q=q, G=G, h=hvec, A=Aeq, b=beq, x_ref=x_ref, foot_ref=foot_ref,
                A_list=A_list, B_list=B_list, R=R)


# --------------------------------------------------------------------------------------
# Condensed form: X = s + Bqp U,  min 1/2 U'Hc U + gc'U  s.t. C U <= d
# --------------------------------------------------------------------------------------
def condense(P, q, G, hvec, A, b, nx):
    """Generic elimination of the first nx variables through the (square, invertible) equality
    block A[:, :nx].  Works on the captured reference matrices without assuming their structure.
    Returns Hc, gc, C, d, s, Bqp with z = [s + Bqp U; U]."""
    AX, AU = A[:, :nx], A[:, nx:]
    assert AX.shape[0] == nx
    assert not np.any(G[:, :nx]), "inequalities touch the state block"
    s = np.linalg.solve(AX, np.asarray(b, float).reshape(-1))
    Bqp = -np.linalg.solve(AX, AU)
    PXX, PXU, PUU = P[:nx, :nx], P[:nx, nx:], P[nx:, nx:]
    q = np.asarray(q, float).reshape(-1)
    Hc = Bqp.T @ PXX @ Bqp + Bqp.T @ PXU + PXU.T @ Bqp + PUU
    gc = Bqp.T @ (PXX @ s + q[:nx]) + PXU.T @ s + q[nx:]
    return Hc, gc, G[:, nx:], np.asarray(hvec, float).reshape(-1), s, Bqp


def build_condensed_qp(x_fb, t, foot, mpc, biped, contact, half=None, mu_steps=None):
    """Structured condensing by the recurrences of SURVEY App. B (what the HIP kernels do):
    s_i = A_i s_{i-1} (s_-1 = x0),  Gamma_{i,j} = A_i Gamma_{i-1,j},  Gamma_{j,j} = B_j.
    Returns dict(Hc, gc, s, Bqp, lb, ub, mu, lf) where the constraints are kept structured:
    lb <= U <= ub (12h), friction |f_xy| <= mu f_z, line-foot rows lf (4 x 12 per step, <= 0)."""
    sp = build_sparse_qp(x_fb, t, foot, mpc, biped, contact, half=half, mu_steps=mu_steps)
    h = mpc.h
    A_list, B_list = sp["A_list"], sp["B_list"]
    s = np.zeros((h, 13))
    Bqp = np.zeros((13 * h, 12 * h))
    prev = np.append(np.asarray(x_fb, float), 1.0)
    for i in range(h):
        prev = A_list[i] @ prev
        s[i] = prev
        Bqp[13 * i:13 * i + 13, 12 * i:12 * i + 12] = B_list[i]
        if i > 0:
            Bqp[13 * i:13 * i + 13, :12 * i] = A_list[i] @ Bqp[13 * (i - 1):13 * i, :12 * i]
    Qbar = np.kron(np.eye(h), np.diag(np.asarray(mpc.Q, float)))
    Rbar = np.kron(np.eye(h), np.diag(np.asarray(mpc.R, float)))
    Hc = 2 * (Bqp.T @ Qbar @ Bqp + Rbar)
    gc = 2 * Bqp.T @ Qbar @ (s.reshape(-1) - sp["x_ref"].T.flatten())
    d = sp["h"].reshape(-1)
    ub = np.concatenate([d[8 * h + 24 * k:8 * h + 24 * k + 12] for k in range(h)])
    lb = -np.concatenate([d[8 * h + 24 * k + 12:8 * h + 24 * k + 24] for k in range(h)])
    return dict(Hc=Hc, gc=gc, s=s.reshape(-1), Bqp=Bqp, lb=lb, ub=ub,
                C=sp["G"][:, 13 * h:], d=d, sparse=sp)


# --------------------------------------------------------------------------------------
# fp64 QP solver with KKT certificate
# --------------------------------------------------------------------------------------
def _pinned_pairs(C, d, tol=0.0):
    """Rows i, j of C U <= d with C_j = -C_i and d_j = -d_i pin  C_i U = d_i  (REF:240-249 with
    contact = 0, or tau_max[0] = 0: SURVEY A.6 items 7, 8).  Returns list of (i, j)."""
    m = C.shape[0]
    key = {}
    pairs = []
    for i in range(m):
        row = tuple(np.round(C[i], 14)) + (round(float(d[i]), 14),)
        neg = tuple(-v if v != 0 else 0.0 for v in row)
        row = tuple(v if v != 0 else 0.0 for v in row)
        if neg in key and key[neg]:
            pairs.append((key[neg].pop(), i))
        else:
            key.setdefault(row, []).append(i)
    return pairs


def _ipm(H, g, C, d, iters=120, tol=1e-12):
    """Mehrotra predictor-corrector on min 1/2 u'Hu + g'u s.t. Cu <= d (strict interior needed)."""
    n, m = H.shape[0], C.shape[0]
    u = np.zeros(n)
    if m == 0:
        return np.linalg.solve(H, -g), np.zeros(0)
    sl = np.maximum(d - C @ u, 1.0)
    lam = np.ones(m)
    mu_hist = []
    stalled = False
    for _ in range(iters):
        rd = H @ u + g + C.T @ lam
        rp = C @ u + sl - d
        mu = sl @ lam / m
        if max(np.abs(rd).max(), np.abs(rp).max(), mu) < tol:
            break
        # Mehrotra's heuristic centring can lock into a short cycle at a fixed complementarity level (seen on 1 of 32768
        # standing instances: mu cycling through 1e-3 .. 3e-3 for ever, polish then starts from a poor point and fails).
        # A stall -- no 10 % decrease of mu over five iterations -- switches to plain damped path following for the rest.
        mu_hist.append(mu)
        if len(mu_hist) > 5 and mu > 0.9 * mu_hist[-6]:
            stalled = True
        W = lam / sl
        K = H + C.T @ (W[:, None] * C)
        try:
            L = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:        # barrier weights past fp64 range: keep the iterate
            break

        def solve(rc):
            rhs = -rd + C.T @ ((rc - lam * rp) / sl)
            du_ = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            ds_ = -rp - C @ du_
            dl_ = -(rc + lam * ds_) / sl
            return du_, ds_, dl_

        # residual of complementarity: S lam = 0 (affine), then corrected
        rc = sl * lam
        du, ds, dl = solve(rc)

        def step(v, dv):
            neg = dv < 0
            return min(1.0, (-v[neg] / dv[neg]).min()) if neg.any() else 1.0

        aa = min(step(sl, ds), step(lam, dl))
        mu_aff = (sl + aa * ds) @ (lam + aa * dl) / m
        sigma = (mu_aff / mu) ** 3
        rc = sl * lam + ds * dl - sigma * mu
        if stalled:                              # fixed centring, no second-order term, shorter steps
            rc = sl * lam - 0.2 * mu
        du, ds, dl = solve(rc)
        a = min(1.0, (0.9 if stalled else 0.995) * min(step(sl, ds), step(lam, dl)))
        u, sl, lam = u + a * du, sl + a * ds, lam + a * dl
    return u, lam


def _polish(H, g, C, d, u, lam):
    """Active-set polish of an IPM point, safe for degenerate vertices.

    Active set = rows where the multiplier dominates the slack.  The equality QP on those rows is
    solved by the null-space method (rank-revealing SVD of the active rows), and a non-negative
    multiplier vector is recovered by NNLS, which exists iff the point is dual feasible even when
    the active rows are linearly dependent.  Returns (u, lam, ok)."""
    from scipy.optimize import nnls
    m, n = C.shape
    if m == 0:
        return u, lam, True
    slack = d - C @ u
    act = lam > slack
    for _ in range(8):
        idx = np.flatnonzero(act)
        Ca, da = C[idx], d[idx]
        if len(idx):
            Um, sv, Vt = np.linalg.svd(Ca, full_matrices=True)
            rank = int(np.sum(sv > 1e-11 * max(1.0, sv[0])))
            up = Vt[:rank].T @ ((Um[:, :rank].T @ da) / sv[:rank])
            Z = Vt[rank:].T
        else:
            up, Z = np.zeros(n), np.eye(n)
        if Z.shape[1]:
            w = np.linalg.solve(Z.T @ H @ Z, -Z.T @ (H @ up + g))
            un = up + Z @ w
        else:
            un = up
        grad = H @ un + g
        lam_full = np.zeros(m)
        if len(idx):
            la, _ = nnls(Ca.T, -grad, maxiter=50 * n)
            lam_full[idx] = la
        stat = np.abs(grad + C.T @ lam_full).max()
        viol = C @ un - d
        worst = int(np.argmax(viol))
        if viol[worst] > 1e-11 * (1 + abs(d[worst])):
            act = act.copy()
            act[worst] = True                     # a row the IPM called inactive is violated
            continue
        if stat > 1e-9 * (1 + np.abs(g).max()):
            # not dual feasible on this set: release the row with the most negative LS multiplier
            ls = np.linalg.lstsq(Ca.T, -grad, rcond=None)[0]
            act = act.copy()
            act[idx[int(np.argmin(ls))]] = False
            continue
        return un, lam_full, True
    return u, lam, False


def solve_qp(P, q, G, hvec, A, b, nx, polish=True):
    """Unique minimiser of the reference QP (REF:297 arguments) in fp64.

    Steps: condense through the equality block; turn opposing inequality pairs into fixed
    directions (pinned variables) and eliminate them; drop rows that became empty;
    Mehrotra IPM on the strictly feasible remainder; active-set polish (`polish=False` stops after the
    IPM: the plain fp64 solve bench.py times as the reference-style CPU baseline).  Returns
    (z, lam (rows of G), nu (rows of A), info) with info['kkt'] the certificate residuals.
    """
    P = np.asarray(P, float)
    G = np.asarray(G, float)
    A = np.asarray(A, float)
    Hc, gc, C, d, s, Bqp = condense(P, q, G, hvec, A, b, nx)
    n = Hc.shape[0]
    pairs = _pinned_pairs(C, d)
    fixed = np.zeros(n, bool)
    uval = np.zeros(n)
    pair_rows = set()
    for i, j in pairs:
        nz = np.flatnonzero(C[i])
        if len(nz) != 1:
            raise NotImplementedError("pinned direction is not a single variable")
        fixed[nz[0]] = True
        uval[nz[0]] = d[i] / C[i, nz[0]]
        pair_rows.update((i, j))
    free = ~fixed
    rows = np.array([r for r in range(C.shape[0]) if r not in pair_rows], int)
    Cr = C[rows][:, free]
    dr = d[rows] - C[rows][:, fixed] @ uval[fixed]
    nonempty = np.abs(Cr).sum(axis=1) > 0
    if np.any(dr[~nonempty] < 0):
        raise RuntimeError("infeasible: empty row with negative rhs")
    rows, Cr, dr = rows[nonempty], Cr[nonempty], dr[nonempty]
    Hr = Hc[np.ix_(free, free)]
    gr = gc[free] + Hc[np.ix_(free, fixed)] @ uval[fixed]
    ur, lr = _ipm(Hr, gr, Cr, dr)
    polished = False
    if polish:
        ur, lr, polished = _polish(Hr, gr, Cr, dr, ur, lr)
    U = uval.copy()
    U[free] = ur
    X = s + Bqp @ U
    z = np.concatenate([X, U])
    lam = np.zeros(G.shape[0])
    lam[rows] = lr
    # multipliers: nu from the state block of stationarity, pinned pairs from the control block
    qv = np.asarray(q, float).reshape(-1)
    nu = -np.linalg.solve(A[:, :nx].T, P[:nx] @ z + qv[:nx])
    resid_u = P[nx:] @ z + qv[nx:] + G[:, nx:].T @ lam + A[:, nx:].T @ nu
    for i, j in pairs:
        v = int(np.flatnonzero(C[i])[0])
        need = -resid_u[v] / C[i, v]         # lam_i - lam_j
        lam[i], lam[j] = max(need, 0.0), max(-need, 0.0)
    info = dict(kkt=kkt_residuals(P, q, G, hvec, A, b, z, lam, nu),
                polished=bool(polished), n_pinned=int(fixed.sum()), n_active=int((lam[rows] > 0).sum()),
                objective=float(0.5 * z @ P @ z + qv @ z))
    return z, lam, nu, info


def kkt_residuals(P, q, G, hvec, A, b, z, lam, nu):
    """Max-abs KKT residuals of (z, lam, nu) for min 1/2 z'Pz + q'z, Gz <= h, Az = b."""
    q = np.asarray(q, float).reshape(-1)
    hvec = np.asarray(hvec, float).reshape(-1)
    b = np.asarray(b, float).reshape(-1)
    stat = P @ z + q + G.T @ lam + A.T @ nu
    slack = hvec - G @ z
    return dict(stationarity=float(np.abs(stat).max()),
                primal_eq=float(np.abs(A @ z - b).max()),
                primal_ineq=float(max(0.0, (-slack).max())),
                dual=float(max(0.0, (-lam).max())),
                complementarity=float(np.abs(lam * slack).max()))


def certificate_from_primal(P, q, G, hvec, A, b, nx, U, act_tol=1e-8):
    """KKT certificate RECOMPUTED from a primal point alone (tests: a stored optimum is checked against
    matrices rebuilt by the reference-pinned assembly, not against numbers its generator wrote).
    X follows from the equality block; nu from the state block of stationarity; the multipliers of the
    active inequality rows (slack <= act_tol (1 + |h|)) by non-negative least squares on the control block
    -- they exist iff the point is the minimiser (strictly convex QP), also at degenerate vertices.
    Returns the same residual dict as `kkt_residuals` plus `n_active`."""
    from scipy.optimize import nnls
    P = np.asarray(P, float)
    G = np.asarray(G, float)
    A = np.asarray(A, float)
    q = np.asarray(q, float).reshape(-1)
    hvec = np.asarray(hvec, float).reshape(-1)
    b = np.asarray(b, float).reshape(-1)
    U = np.asarray(U, float).reshape(-1)
    X = np.linalg.solve(A[:, :nx], b - A[:, nx:] @ U)
    z = np.concatenate([X, U])
    nu = -np.linalg.solve(A[:, :nx].T, P[:nx] @ z + q[:nx])
    slack = hvec - G @ z
    act = np.flatnonzero(slack <= act_tol * (1.0 + np.abs(hvec)))
    ru = P[nx:] @ z + q[nx:] + A[:, nx:].T @ nu
    lam = np.zeros(G.shape[0])
    if len(act):
        la, _ = nnls(G[act][:, nx:].T, -ru, maxiter=100 * len(U))
        lam[act] = la
    out = kkt_residuals(P, q, G, hvec, A, b, z, lam, nu)
    out["n_active"] = int(len(act))
    return out


# --------------------------------------------------------------------------------------
# drop-in (REF:187-304) and the consumer side (REF:306-470) for end-to-end checks
# --------------------------------------------------------------------------------------
def solve_mpc(x_fb, t, foot, mpc, biped, contact, half=None, mu_steps=None, return_info=False):
    """Same call surface and return shapes as REF:187, 304 (silent: REF:190-192 prints dropped)."""
    sp = build_sparse_qp(x_fb, t, foot, mpc, biped, contact, half=half, mu_steps=mu_steps)
    h = mpc.h
    z, lam, nu, info = solve_qp(sp["P"], sp["q"], sp["G"], sp["h"], sp["A"], sp["b"], 13 * h)
    states = z[:13 * h].reshape((h, 13))
    controls = z[13 * h:].reshape((h, 12))
    if return_info:
        return states, controls, info
    return states, controls


def getLegKinematics(q0, q1, q2, q3, q4, side):
    """REF:306-365: closed-form 6x5 leg Jacobian."""
    s, c = np.sin, np.cos
    a = 0.04 * s(q2 + q3 + q4) + 0.22 * s(q2 + q3) + 0.22 * s(q2)
    bq = 0.04 * c(q2 + q3 + q4) + 0.22 * c(q2 + q3) + 0.22 * c(q2)
    a3 = 0.04 * s(q2 + q3 + q4) + 0.22 * s(q2 + q3)
    b3 = 0.04 * c(q2 + q3 + q4) + 0.22 * c(q2 + q3)
    a4 = 0.04 * s(q2 + q3 + q4)
    b4 = 0.04 * c(q2 + q3 + q4)
    e = 0.018 * side + 0.0025
    Jm = np.zeros((6, 5))
    Jm[0, 0] = s(q0) * (a + 0.0135) + c(q0) * (0.015 * side + c(q1) * e - s(q1) * bq)
    Jm[1, 0] = s(q0) * (0.015 * side + c(q1) * e - s(q1) * bq) - c(q0) * (a + 0.0135)
    Jm[5, 0] = 1.0
    Jm[0, 1] = -s(q0) * (s(q1) * e + c(q1) * bq)
    Jm[1, 1] = c(q0) * (s(q1) * e + c(q1) * bq)
    Jm[2, 1] = s(q1) * bq - c(q1) * e
    Jm[3, 1] = c(q0)
    Jm[4, 1] = s(q0)
    for col, (aa, bb) in zip((2, 3, 4), ((a, bq), (a3, b3), (a4, b4))):
        Jm[0, col] = s(q0) * s(q1) * aa - c(q0) * bb
        Jm[1, col] = -s(q0) * bb - c(q0) * s(q1) * aa
        Jm[2, col] = c(q1) * aa
        Jm[3, col] = -c(q1) * s(q0)
        Jm[4, col] = c(q0) * c(q1)
        Jm[5, col] = s(q1)
    return Jm, Jm[0:3, :]


def getFootPositionBody(q0, q1, q2, q3, q4, side):
    """REF:367-404: closed-form foot position in the body frame."""
    s, c = np.sin, np.cos
    u = c(q0) * s(q2) + c(q2) * s(q0) * s(q1)
    v = c(q0) * c(q2) - s(q0) * s(q1) * s(q2)
    w = s(q0) * s(q2) - c(q0) * c(q2) * s(q1)
    y = c(q2) * s(q0) + c(q0) * s(q1) * s(q2)
    pf = np.zeros(3)
    pf[0] = (-3 * c(q0) / 200 - 9 * s(q4) * (c(q3) * v - s(q3) * u) / 250 - 11 * c(q0) * s(q2) / 50
             - side * s(q0) / 50 - 11 * c(q3) * u / 50 - 11 * s(q3) * v / 50
             - 9 * c(q4) * (c(q3) * u + s(q3) * v) / 250 - 23 * c(q1) * side * s(q0) / 1000
             - 11 * c(q2) * s(q0) * s(q1) / 50)
    pf[1] = (c(q0) * side / 50 - 9 * s(q4) * (c(q3) * y - s(q3) * w) / 250 - 3 * s(q0) / 200
             - 11 * s(q0) * s(q2) / 50 - 11 * c(q3) * w / 50 - 11 * s(q3) * y / 50
             - 9 * c(q4) * (c(q3) * w + s(q3) * y) / 250 + 23 * c(q0) * c(q1) * side / 1000
             + 11 * c(q0) * c(q2) * s(q1) / 50)
    pf[2] = (23 * side * s(q1) / 1000 - 11 * c(q1) * c(q2) / 50
             - 9 * c(q4) * (c(q1) * c(q2) * c(q3) - c(q1) * s(q2) * s(q3)) / 250
             + 9 * s(q4) * (c(q1) * c(q2) * s(q3) + c(q1) * c(q3) * s(q2)) / 250
             - 11 * c(q1) * c(q2) * c(q3) / 50 + 11 * c(q1) * s(q2) * s(q3) / 50 - 3.0 / 50.0)
    return pf


def getFootPositionWorld(x_fb, q, biped):
    """REF:406-424 (uses R.T for body->world: SURVEY A.6 item 14)."""
    R = eul2rotm(x_fb[0:3])
    pf_w = np.zeros((6, 1))
    for leg in range(2):
        side = 1 if leg == 0 else -1
        pf_b = getFootPositionBody(*q[5 * leg:5 * leg + 5], side).reshape(-1, 1)
        hip = np.array([[biped.hip_offset[0]], [side * biped.hip_offset[1]], [biped.hip_offset[2]]])
        pf_w[3 * leg:3 * leg + 3] = np.asarray(x_fb[3:6], float).reshape(-1, 1) + R.T @ (pf_b + hip)
    return pf_w


def swingLegControl(x_fb, t, pf_w, vf_w, mpc, side):
    """REF:426-442."""
    fx = x_fb[3] + x_fb[9] * 1 / 2 * mpc.h / 2 * mpc.dt + mpc.kv * (x_fb[3] - mpc.x_cmd[3])
    fy = (x_fb[4] + x_fb[10] * 1 / 2 * mpc.h / 2 * mpc.dt + mpc.kv * (x_fb[4] - mpc.x_cmd[4])
          + 0.04 * side)
    tt = np.remainder(t, mpc.dt * mpc.h / 2)
    fz = mpc.swingHeight * np.sin(np.pi * tt / (mpc.dt * mpc.h / 2))
    foot_des = np.array([[fx], [fy], [fz]])
    return mpc.kp @ (foot_des - pf_w) + mpc.kd @ (np.zeros((3, 1)) - vf_w)


def lowLevelControl(x_fb, t, pf_w, q, qd, mpc, biped, contact, u):
    """REF:444-470: force/moment -> joint torque map; consumer of controls[0] as a (12,1) array."""
    tau = np.zeros((10, 1))
    c = contact[0, 0:2]
    R = eul2rotm(x_fb[0:3])
    for leg in range(2):
        side = 1 if leg == 0 else -1
        Jm, Jf = getLegKinematics(*q[5 * leg:5 * leg + 5], side)
        vf_w = R.T @ Jf @ qd[5 * leg:5 * leg + 5].reshape(-1, 1)
        F_swing = swingLegControl(x_fb, t, pf_w[3 * leg:3 * leg + 3], vf_w, mpc, side)
        u_w = -np.vstack([R.T @ u[3 * leg:3 * leg + 3], R.T @ u[3 * leg + 6:3 * leg + 9]])
        tau[5 * leg:5 * leg + 5, :] = Jm.T @ u_w * c[leg]
        tau[5 * leg:5 * leg + 5, :] += Jf.T @ F_swing * -(c[leg] - 1)
    return tau
