"""NumPy model of the wrench-space ADMM the HIP kernels implement (TEST INFRASTRUCTURE ONLY).

Not a restatement of the reference (that is bmpc_oracle.py) but an executable specification of
the *product's* algorithm, written batch-vectorised so that a test can compare every
intermediate the kernels can dump (references, wrench-space Hessian Gt, gradient qt, optimum)
in fp64 or fp32.  Only tests/ may import it.

Formulation.  Per horizon step j the 12 controls u_j = [f1 f2 m1 m2] act on
the single rigid body only through the net wrench b_j = W_j u_j = [tau_j; F_j] (6), W_j =
[[r1]x [r2]x I I; I I 0 0] (REF:174-180).  With X = s + Gam_t b (Gam_t: 12h x 6h, closed form
from REF:165-184) the condensed Hessian of SURVEY App. B factors as
    Hc = Wbar' Gt Wbar + 2 Rbar,   gc = Wbar' qt,   Gt = 2 Gam_t' Qbar Gam_t,  qt = 2 Gam_t' Qbar (s - x_ref)
and the ADMM x-update matrix K = Hc + A' diag(rho) A  (A = [I; friction; line-foot] is block
diagonal per (step, foot)) inverts exactly as
    K^-1 = N Ka^-1 N' + L (Gt + F)^-1 L',   D = 2R + A' rho A (6x6 blocks),  F_j = (W_j D_j^-1 W_j')^-1,
    L_j = D_j^-1 W_j' F_j,  N_j = null(W_j) (closed form),  Ka_j = N_j' D_j N_j.
Only V = (Gt + F)^-1 (6h x 6h) is dense.  gc and Hc are never formed in control space, which is
what keeps fp32 accurate: directions in null(Wbar) see exactly 2R.
"""
from __future__ import annotations

import numpy as np


class Params:
    """Flat parameter block shared by host and device (mirrors include/bmpc.h bmpc_params)."""

    def __init__(self, mpc=None, biped=None, h=None, half=None):
        from . import bmpc_oracle as orc
        mpc = mpc or orc.MPC()
        biped = biped or orc.Biped()
        self.h = int(h if h is not None else mpc.h)
        self.half = int(half if half is not None else 5)
        self.dt = float(mpc.dt)
        self.kv = float(mpc.kv)
        self.x_cmd = np.asarray(mpc.x_cmd, float).copy()
        self.Q = np.asarray(mpc.Q, float).copy()
        self.R = np.asarray(mpc.R, float).copy()
        self.m = float(biped.m)
        self.g = float(biped.g)
        self.I = np.asarray(biped.I, float).copy()
        self.mu = float(biped.mu)
        self.lt = float(biped.lt) - 0.01           # REF:254
        self.lh = float(biped.lh) - 0.02           # REF:255
        self.f_max = np.asarray(biped.f_max, float).reshape(3)
        self.f_min = np.asarray(biped.f_min, float).reshape(3)
        self.tau_max = np.asarray(biped.tau_max, float).reshape(3)
        self.tau_min = np.asarray(biped.tau_min, float).reshape(3)
        # solver
        self.rho = 0.03
        self.rho_eq_scale = 1e3
        self.rho_lo = 3e-4
        self.rho_hi_f = 1.0
        self.rho_hi_m = 100.0
        self.adapt_start = 10
        self.adapt_every = 10
        self.adapt_early = 0       # two-rate schedule (bmpc_params.adapt_early / adapt_late): the first adapt_early re-classifications
        self.adapt_late = 0        # adapt_every apart, the later ones adapt_late; 0: one rate
        self.adapt_busy = 0        # ... or adapt_busy after one that found more than adapt_flips rows in another class (0: off)
        self.adapt_flips = 1
        self.kappa_confirm = 0.0   # a row found in the same class as at the previous re-classification moves by this (0: off),
        self.confirm_from = 0      # from re-classification number confirm_from + 1 on
        self.blocks_hi = True
        self.kappa = 20.0
        self.alpha = 1.6
        self.max_iter = 400
        self.check_every = 5
        self.eps_pri = 1e-7
        self.eps_dua = 1e-7
        self.max_refactor = 24
        self.accel = True          # secant extrapolation at the stopping tests (bmpc_params.accel; dense family)


def _skew(v):
    z = np.zeros(v.shape[:-1], v.dtype)
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def references(P, x_fb, foot, contact, phase, x_cmd, dt_):
    """x_ref (B,h,12) and foot_ref (B,h,6)  (REF:61-109)."""
    B, h = x_fb.shape[0], P.h
    dt = dt_(P.dt)
    j = np.arange(h).astype(x_fb.dtype)
    x_ref = np.repeat(x_cmd[:, None, :], h, axis=1).copy()
    moving = x_cmd[:, 6:12] != 0
    ramp = x_fb[:, None, 0:6] + x_cmd[:, None, 6:12] * (j[None, :, None] * dt)
    x_ref[:, :, 0:6] = np.where(moving[:, None, :], ramp, x_cmd[:, None, 0:6])
    x_ref[:, 0, :] = x_fb
    kv = dt_(P.kv)
    half_h = dt_(0.5) * dt_(P.h) / dt_(2) * dt
    full_h = dt_(0.5) * dt_(P.h) * dt
    fx1 = x_fb[:, 3] + x_fb[:, 9] * half_h + kv * (x_fb[:, 3] - x_cmd[:, 3])
    fx2 = x_fb[:, 3] + x_fb[:, 9] * full_h + kv * (x_fb[:, 3] - x_cmd[:, 3])
    fy1 = x_fb[:, 4] + x_fb[:, 10] * half_h + kv * (x_fb[:, 4] - x_cmd[:, 4])
    fy2 = x_fb[:, 10] + x_fb[:, 10] * full_h + kv * (x_fb[:, 4] - x_cmd[:, 4])     # REF:87 quirk
    z = np.zeros_like(fx1)
    foot_1 = np.stack([fx1, fy1, z, fx1, fy1, z], -1)
    foot_2 = np.stack([fx2, fy2, z, fx2, fy2, z], -1)
    single = (contact[:, 0, 0].astype(int) + contact[:, 0, 1].astype(int)) == 1
    kk = phase % P.half
    jj = np.arange(h)[None, :]
    sel1 = (jj >= (P.half - kk)[:, None]) & (jj < (2 * P.half - kk)[:, None])
    sel2 = jj >= (2 * P.half - kk)[:, None]
    foot_ref = np.repeat(foot[:, None, :], h, axis=1).copy()
    foot_ref = np.where((single[:, None] & sel1)[..., None], foot_1[:, None, :], foot_ref)
    foot_ref = np.where((single[:, None] & sel2)[..., None], foot_2[:, None, :], foot_ref)
    return x_ref, foot_ref


def step_quantities(P, x_ref, foot_ref, dt_):
    """Per step: Iw^-1 (B,h,3,3), Rinv (B,h,3,3), r (B,h,2,3)  (REF:151-175)."""
    yaw, pitch, roll = x_ref[..., 0], x_ref[..., 1], x_ref[..., 2]
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    one, zero = np.ones_like(cy), np.zeros_like(cy)
    Rz = np.stack([np.stack([cy, -sy, zero], -1), np.stack([sy, cy, zero], -1), np.stack([zero, zero, one], -1)], -2)
    Ry = np.stack([np.stack([cp, zero, sp], -1), np.stack([zero, one, zero], -1), np.stack([-sp, zero, cp], -1)], -2)
    Rx = np.stack([np.stack([one, zero, zero], -1), np.stack([zero, cr, -sr], -1), np.stack([zero, sr, cr], -1)], -2)
    Rot = Rx @ Ry @ Rz
    Iinv_b = np.linalg.inv(P.I).astype(x_ref.dtype)
    Iw_inv = np.swapaxes(Rot, -1, -2) @ Iinv_b @ Rot
    tp = sp / cp
    Rinv = np.stack([np.stack([cy / cp, sy / cp, zero], -1), np.stack([-sy, cy, zero], -1),
                     np.stack([cy * tp, sy * tp, one], -1)], -2)
    r = foot_ref.reshape(foot_ref.shape[:-1] + (2, 3)) - x_ref[..., None, 3:6]
    return Iw_inv, Rinv, r


def wrench_hessian(P, x_fb, x_ref, Iw_inv, Rinv, dt_):
    """Gt (B,6h,6h), qt (B,6h), free response s (B,h,12) -- wrench order per step [tau(3), F(3)]."""
    B, h = x_fb.shape[0], P.h
    dtp = x_fb.dtype
    dt = dt_(P.dt)
    Q = P.Q.astype(dtp)
    Pre = np.cumsum(Rinv, axis=1)                                       # P_i = sum_{l<=i} Rinv_l
    i1 = np.arange(1, h + 1).astype(dtp)
    s = np.zeros((B, h, 12), dtp)
    s[:, :, 0:3] = x_fb[:, None, 0:3] + dt * np.einsum("bhij,bj->bhi", Pre, x_fb[:, 6:9])
    s[:, :, 3:6] = x_fb[:, None, 3:6] + dt * i1[None, :, None] * x_fb[:, None, 9:12]
    s[:, :, 5] -= dt_(P.g) * dt * dt * (i1 - 1) * i1 / 2
    s[:, :, 6:9] = x_fb[:, None, 6:9]
    s[:, :, 9:12] = x_fb[:, None, 9:12]
    s[:, :, 11] -= dt_(P.g) * dt * i1
    err = s - x_ref                                                     # X_i tracks x_ref[:, i] (A.6 item 9)
    # Gam_t blocks (i >= j):  e<-tau: dt^2 (P_i - P_j) Iw_j^-1 ; w<-tau: dt Iw_j^-1 ; p<-F: dt^2/m (i-j) ; v<-F: dt/m
    Me = dt * dt * np.einsum("bijkl,bjlm->bijkm", Pre[:, :, None] - Pre[:, None, :], Iw_inv)  # (B,i,j,3,3)
    Nw = dt * Iw_inv
    mask = (np.arange(h)[:, None] >= np.arange(h)[None, :]).astype(dtp)   # i >= j
    Me = Me * mask[None, :, :, None, None]
    Gt = np.zeros((B, h, 6, h, 6), dtp)
    QeMe = Q[0:3][None, None, None, :, None] * Me
    Gtt = 2 * np.einsum("bijkl,bimkn->bjlmn", Me, QeMe)                   # sum_i Me_ij' Qe Me_im
    cnt = (h - np.maximum(np.arange(h)[:, None], np.arange(h)[None, :])).astype(dtp)
    Gtt += 2 * cnt[None, :, None, :, None] * np.einsum("bjkl,k,bmkn->bjlmn", Nw, Q[6:9], Nw)
    Gt[:, :, 0:3, :, 0:3] = Gtt
    ii = np.arange(h).astype(dtp)
    lag = (ii[:, None] - ii[None, :]) * mask                              # (i-j) for i>=j
    cp_ = (dt * dt / dt_(P.m)) ** 2 * np.einsum("ij,im->jm", lag, lag)
    cv_ = (dt / dt_(P.m)) ** 2 * cnt
    for a in range(3):
        Gt[:, :, 3 + a, :, 3 + a] = 2 * (Q[3 + a] * cp_ + Q[9 + a] * cv_)[None]
    qt = np.zeros((B, h, 6), dtp)
    qt[:, :, 0:3] = 2 * (np.einsum("bijkl,k,bik->bjl", Me, Q[0:3], err[:, :, 0:3])
                         + np.einsum("bjkl,k,bik,ij->bjl", Nw, Q[6:9], err[:, :, 6:9], mask))
    qt[:, :, 3:6] = 2 * ((dt * dt / dt_(P.m)) * np.einsum("ij,k,bik->bjk", lag, Q[3:6], err[:, :, 3:6])
                         + (dt / dt_(P.m)) * np.einsum("ij,k,bik->bjk", mask, Q[9:12], err[:, :, 9:12]))
    return Gt.reshape(B, 6 * h, 6 * h), qt.reshape(B, 6 * h), s


def constraint_blocks(P, x_fb, contact, mu, dt_):
    """Per (step, foot) block over v = [f(3), m(3)]: A (B,h,2,12,6), l, u (B,h,2,12).
    Rows: 6 box, 4 friction (REF:220-229 order +x,+y,-x,-y), 2 line-foot (REF:259-262)."""
    B, h = x_fb.shape[0], P.h
    dtp = x_fb.dtype
    cr, cp, cy = np.cos(x_fb[:, 0]), np.cos(x_fb[:, 1]), np.cos(x_fb[:, 2])
    sr, sp, sy = np.sin(x_fb[:, 0]), np.sin(x_fb[:, 1]), np.sin(x_fb[:, 2])
    # R = Rz(e2) Ry(e1) Rx(e0) (REF:124-138); ey = R[:,1], ez = R[:,2]
    ey = np.stack([cy * sp * sr - sy * cr, sy * sp * sr + cy * cr, cp * sr], -1)
    ez = np.stack([cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr], -1)
    A = np.zeros((B, h, 2, 12, 6), dtp)
    for i in range(6):
        A[..., i, i] = 1
    sg = [(0, 1.0), (1, 1.0), (0, -1.0), (1, -1.0)]
    for r_, (ax, s_) in enumerate(sg):
        A[..., 6 + r_, ax] = s_
        A[..., 6 + r_, 2] = -mu
    A[..., 10, 0:3] = (-dt_(P.lh) * ez)[:, None, None, :]
    A[..., 10, 3:6] = ey[:, None, None, :]
    A[..., 11, 0:3] = (-dt_(P.lt) * ez)[:, None, None, :]
    A[..., 11, 3:6] = -ey[:, None, None, :]
    c = contact.astype(dtp)[..., None]                                    # (B,h,2,1)
    ub = np.concatenate([c * P.f_max.astype(dtp), c * P.tau_max.astype(dtp)], -1)
    lb = np.concatenate([c * P.f_min.astype(dtp), c * P.tau_min.astype(dtp)], -1)
    big = dtp.type(np.inf)
    u = np.concatenate([ub, np.zeros((B, h, 2, 6), dtp)], -1)
    l = np.concatenate([lb, np.full((B, h, 2, 6), -big, dtp)], -1)
    return A, l, u


def _factor(P, Gt, A, rv, Rblk, Wf, Nf, dtp, blk=None, ric=None):
    """Everything that depends on the per-row penalties rv: L, Na (block diagonal) and V (dense).
    blk: arithmetic of the 6x6 block algebra (default dtp); the dense sweep is always dtp."""
    B, h = rv.shape[0], P.h
    bt = np.dtype(blk or dtp)
    A_, rv_, Wf_, Nf_ = A.astype(bt), rv.astype(bt), Wf.astype(bt), Nf.astype(bt)
    D = np.einsum("bhfri,bhfr,bhfrj->bhfij", A_, rv_, A_)
    D = D + (Rblk.astype(bt)[None, None, :, :, None] * np.eye(6, dtype=bt))
    Dinv = np.linalg.inv(D)
    E = np.einsum("bhfij,bhfjk,bhflk->bhil", Wf_, Dinv, Wf_)
    F = np.linalg.inv(E)
    L = np.einsum("bhfij,bhfkj,bhkl->bhfil", Dinv, Wf_, F).astype(bt if getattr(P, "f64_L", False) else dtp)    # (B,h,2,6,6): D^-1 W' F
    Ka = np.einsum("bhfij,bhfik,bhfkl->bhjl", Nf_, D, Nf_)
    Kainv = np.linalg.inv(Ka)
    Na = np.einsum("bhfij,bhjk,bhglk->bhfigl", Nf_, Kainv, Nf_).astype(bt if getattr(P, "f64_Na", False) else dtp)  # (B,h,f,6,g,6)
    F32 = F.astype(dtp)
    if ric is not None:
        # stage-structured path: V is never formed, gamma = V beta is a Riccati solve
        from . import riccati_model
        return L, Na, riccati_model.factor(P, F32, ric[0], ric[1], dtp)
    K = Gt.astype(dtp).copy()
    for j in range(h):
        K[:, 6 * j:6 * j + 6, 6 * j:6 * j + 6] += F32[:, j]
    V = np.linalg.inv(K).astype(dtp)
    return L, Na, V


def solve_batch(P, x_fb, foot, contact, phase, x_cmd=None, mu=None, dtype=np.float32,
                res_dtype=None, iters=None, return_debug=False, round_data=False):
    """Model of bmpc_solve_batch.  Returns states (B,h,13), controls (B,h,12), info dict.

    dtype      arithmetic of the set-up and of the preconditioner K^-1 (the bulk of the flops)
    res_dtype  arithmetic of the iterates and of the residual (defaults to dtype)
    """
    dtp = np.dtype(dtype)
    rdt = np.dtype(res_dtype or dtype)
    pdt, pdt_ = dtp, dtp.type                  # preconditioner arithmetic
    dtp = rdt                                  # problem data + iterates + residual arithmetic
    dt_ = dtp.type
    x_fb = np.asarray(np.asarray(x_fb, np.float32 if pdt == np.float32 else float), dtp)   # fp32 inputs at the ABI
    foot = np.asarray(np.asarray(foot, np.float32 if pdt == np.float32 else float), dtp)
    B, h = x_fb.shape[0], P.h
    contact = np.asarray(contact).reshape(B, h, 2)
    phase = np.asarray(phase, int).reshape(B)
    x_cmd = (np.repeat(P.x_cmd[None], B, 0) if x_cmd is None else np.asarray(x_cmd)).astype(dtp)
    mu = (np.full((B, h, 2), P.mu) if mu is None else np.asarray(mu)).astype(dtp)
    x_ref, foot_ref = references(P, x_fb, foot, contact, phase, x_cmd, dt_)
    Iw_inv, Rinv, r = step_quantities(P, x_ref, foot_ref, dt_)
    Gt, qt, s = wrench_hessian(P, x_fb, x_ref, Iw_inv, Rinv, dt_)
    A, l, u = constraint_blocks(P, x_fb, contact, mu, dt_)
    if round_data:          # experiment: problem data held in the preconditioner's precision
        Gt, qt, A = (a.astype(pdt).astype(dtp) for a in (Gt, qt, A))
        r, Iw_inv, Rinv = (a.astype(pdt).astype(dtp) for a in (r, Iw_inv, Rinv))
    eq = l == u
    rho0 = dt_(P.rho)
    rho_eq = dt_(P.rho * P.rho_eq_scale)
    # row classes for the active-set adaptive penalties: force-like rows / moment-like rows
    hi = np.empty(12, dtp)
    hi[[0, 1, 2, 6, 7, 8, 9]] = P.rho_hi_f
    hi[[3, 4, 5, 10, 11]] = P.rho_hi_m
    ds = getattr(P, "ds_unscaled", None)
    if ds is not None:
        # double-support steps keep the reference ceilings: an active row there sees the soft curvature (the other foot
        # takes over), only single-support rows see the stiff one that grows with the horizon
        other = contact[:, :, ::-1].astype(bool)[..., None]                # (B,h,2,1): the other foot stands
        hi_ds = np.empty(12, dtp)
        hi_ds[[0, 1, 2, 6, 7, 8, 9]] = ds[0]
        hi_ds[[3, 4, 5, 10, 11]] = ds[1]
        hi = np.where(other, np.minimum(hi, hi_ds), hi)                    # (B,h,2,12)
    rv = np.where(eq, rho_eq, rho0).astype(dtp)                            # (B,h,2,12)
    if getattr(P, "rv_init", None) is not None:                            # (experiments: start from given penalties)
        rv = np.where(eq, rho_eq, np.asarray(P.rv_init)).astype(dtp)
    R2 = 2 * P.R.astype(dtp)
    Rblk = np.stack([np.concatenate([R2[0:3], R2[6:9]]), np.concatenate([R2[3:6], R2[9:12]])])  # (2,6)
    Wf = np.zeros((B, h, 2, 6, 6), dtp)                                    # per foot [[r]x I; I 0]
    Wf[..., 0:3, 0:3] = _skew(r)
    Wf[..., 0:3, 3:6] = np.eye(3, dtype=dtp)
    Wf[..., 3:6, 0:3] = np.eye(3, dtype=dtp)
    # N (null of W): free (phi, nu) -> foot1: (phi, nu); foot2: (-phi, -nu - (r1-r2) x phi)
    Nf = np.zeros((B, h, 2, 6, 6), dtp)
    Nf[:, :, 0] = np.eye(6, dtype=dtp)
    Nf[:, :, 1] = -np.eye(6, dtype=dtp)
    Nf[:, :, 1, 3:6, 0:3] = -_skew(r[:, :, 0] - r[:, :, 1])
    ric = (Iw_inv, Rinv) if getattr(P, "solver", "dense") == "riccati" else None
    fac = lambda rv_: _factor(P, Gt, A, rv_, Rblk, Wf, Nf, pdt, blk=(rdt if P.blocks_hi else pdt), ric=ric)
    L, Na, V = fac(rv)
    n_factor = np.ones(B, int)
    next_adapt = np.full(B, P.adapt_start if P.adapt_every else 10 ** 9)       # per instance, like the kernels' counters
    n_adapt = np.zeros(B, int)
    act_prev = np.zeros(rv.shape, bool)
    seen_act = np.zeros(B, bool)
    alpha = rdt.type(P.alpha)
    x = np.zeros((B, h, 2, 6), rdt)
    z = np.zeros((B, h, 2, 12), rdt)
    y = np.zeros((B, h, 2, 12), rdt)
    Ar, lr, ur = A.astype(rdt), l.astype(rdt), u.astype(rdt)
    Wr, Rr = Wf.astype(rdt), Rblk.astype(rdt)
    Gt_b = Gt.reshape(B, h, 6, h, 6).astype(rdt)
    qt_r = qt.reshape(B, h, 6).astype(rdt)
    n_it = iters if iters is not None else P.max_iter
    # secant extrapolation (Anderson acceleration, memory one) as in bmpc_kernels.hip: g' = state change of the iteration
    # before a stopping test, dropped by a factorisation in between
    accel = bool(getattr(P, "accel", False))
    aa_x_only = getattr(P, "solver", "dense") == "riccati"      # the stage family's metric: the x part of the state alone
    aa_prev = None                                # (g' of every instance, flattened)
    aa_prev_fac = None                            # n_factor when g' was taken
    it_done = np.full(B, n_it)
    active = np.ones(B, bool)
    for it in range(n_it):
        n_factor_at_step = n_factor.copy()
        rvr = rv.astype(rdt)
        # residual (rdt), formed as ONE control-space vector so that it is small at convergence:
        #   r = W'(Gt W x + qt) + 2R x + A'(y + rho (A x - z))
        Ax = np.einsum("bhfri,bhfi->bhfr", Ar, x)
        b = np.einsum("bhfij,bhfj->bhi", Wr, x)
        gb = np.einsum("bhijk,bjk->bhi", Gt_b, b) + qt_r
        r_u = (np.einsum("bhfij,bhi->bhfj", Wr, gb) + Rr[None, None] * x
               + np.einsum("bhfri,bhfr->bhfi", Ar, y + rvr * (Ax - z)))
        # preconditioner (pdt): dx = K^-1 r = Na r + L V L' r
        r32 = r_u.astype(rdt if getattr(P, "f64_r", False) else pdt)
        beta = np.einsum("bhfij,bhfi->bhj", L, r32).reshape(B, 6 * h)
        if ric is not None:
            from . import riccati_model
            gam = riccati_model.solve(V, beta.reshape(B, h, 6))[0]
        else:
            gam = np.einsum("bij,bj->bi", V, beta).reshape(B, h, 6)
        dx = np.einsum("bhfigl,bhgl->bhfi", Na, r32) + np.einsum("bhfil,bhl->bhfi", L, gam)
        xt = x - dx.astype(rdt)
        zt = np.einsum("bhfri,bhfi->bhfr", Ar, xt)
        xn = alpha * xt + (1 - alpha) * x
        zr = alpha * zt + (1 - alpha) * z
        zn = np.clip(zr + y / rvr, lr, ur)
        yn = y + rvr * (zr - zn)
        if (it + 1) % P.check_every == 0:
            rp = np.abs(zt - zn).reshape(B, -1).max(1)
            rs = np.abs(xt - x).reshape(B, -1).max(1)                     # preconditioned residual K^-1 r
            sc_p = np.maximum(np.abs(zt).reshape(B, -1).max(1), 1)
            sc_d = np.maximum(np.abs(xt).reshape(B, -1).max(1), 1)
            done = (rp <= P.eps_pri * sc_p) & (rs <= P.eps_dua * sc_d)
            guard = getattr(P, "slow_guard", 0.0)
            if guard:
                # An inactive row whose penalty is still far above the floor follows at 1 - alpha c / rho per iteration, c the
                # curvature it sees: the residuals are then small because the steps are, not because the iterate has arrived.
                # What such a row still pulls with, rho |z~ - z|, equals c |error|; bounded against the softest curvature
                # 2 R_min it bounds the error.  An instance that fails this test does not stop; it re-classifies at once.
                actn = ((zn <= lr) | (zn >= ur)) & (yn != 0)
                pull = np.where((~actn) & (~eq), rvr * np.abs(zt - zn), 0).reshape(B, -1).max(1)
                slow_any = pull > guard * 2 * P.R.min() * sc_d
                force = done & slow_any & active
                done = done & ~slow_any
                if force.any() and P.kappa:
                    kap = np.where(n_factor <= 10, P.kappa, np.where(n_factor <= 16, P.kappa ** 0.5, P.kappa ** 0.25)).astype(dtp)[:, None, None, None]
                    rvc = np.clip(rv, dt_(P.rho_lo), hi)
                    rnew = np.where(eq, rho_eq, np.where(actn, np.minimum(rvc * kap, hi), np.maximum(rvc / kap, dt_(P.rho_lo)))).astype(dtp)
                    ch = force & (rnew != rv).reshape(B, -1).any(1) & (n_factor <= P.max_refactor)
                    if ch.any():
                        rv = np.where(ch[:, None, None, None], rnew, rv)
                        L, Na, V = fac(rv)
                        n_factor += ch
            newly = active & done
            it_done[newly] = it + 1
            if iters is None:
                active &= ~done
        keep = active[:, None, None, None]
        aa_g = None
        if accel and ((it + 2) % P.check_every == 0 or (it + 1) % P.check_every == 0):
            aa_g = np.concatenate([(xn - x).reshape(B, -1), (zn - z).reshape(B, -1), (yn - y).reshape(B, -1)], 1).astype(np.float32)
        x = np.where(keep, xn, x)
        z = np.where(keep, zn, z)
        y = np.where(keep, yn, y)
        if not active.any():
            break
        due = active & ((it + 1) == next_adapt) if P.adapt_every else np.zeros(B, bool)
        if due.any():
            # per-instance schedule (the kernels: every workgroup has its own counters).  Two rates: the first adapt_early
            # re-classifications adapt_every apart, the later ones adapt_late (bmpc_params.adapt_early / adapt_late)
            n_adapt = n_adapt + due
            act = ((z <= lr) | (z >= ur)) & (y != 0)
            flips = np.where(seen_act, (act != act_prev).reshape(B, -1).sum(1), 0)       # rows whose class changed since the instance's last re-classification
            late = (getattr(P, "adapt_late", 0) > 0) & (n_adapt >= getattr(P, "adapt_early", 0))
            # (... but adapt_busy after a re-classification that still found more than adapt_flips rows in another class)
            busy = (flips > getattr(P, "adapt_flips", 0)) & (getattr(P, "adapt_busy", 0) > 0)
            period = np.where(late, np.where(busy, getattr(P, "adapt_busy", 0), getattr(P, "adapt_late", 0)), P.adapt_every)
            if P.kappa:
                # damping as in the kernel: sqrt(kappa) after 10 factorisations, its square root after 16
                kap = np.where(n_factor <= 10, P.kappa, np.where(n_factor <= 16, P.kappa ** 0.5, P.kappa ** 0.25)).astype(dtp)[:, None, None, None]
                kc = getattr(P, "kappa_confirm", 0)
                if kc:
                    # a row found in the same class as at the instance's previous re-classification is taken at its word: it
                    # moves by kappa_confirm (>= the distance to its limit: straight there) instead of walking its ladder
                    conf = (act == act_prev) & (seen_act & (n_adapt > getattr(P, "confirm_from", 0)))[:, None, None, None] & (n_factor <= 10)[:, None, None, None]
                    kap = np.where(conf, dt_(kc), kap)
                rvc = np.clip(rv, dt_(P.rho_lo), hi)
                up = np.minimum(rvc * kap, hi)
                dn = np.maximum(rvc / kap, dt_(P.rho_lo))
                rnew = np.where(eq, rho_eq, np.where(act, up, dn)).astype(dtp)
                rhook = getattr(P, "rnew_hook", None)          # (experiments: any rule; gets the state it may look at)
                if rhook is not None:
                    rnew = rhook(P, dict(it=it + 1, n_factor=n_factor, rv=rv, rnew=rnew, act=act, eq=eq, hi=hi, z=z, y=y, l=lr, u=ur,
                                         rho_eq=rho_eq, x=x, due=due, flips=flips)).astype(dtp)
            else:
                rnew = np.where(eq, rho_eq, np.where(act, hi, dt_(P.rho_lo))).astype(dtp)
            phook = getattr(P, "period_hook", None)            # (experiments: the next period from what this re-classification saw)
            if phook is not None:
                period = phook(P, dict(it=it + 1, n_adapt=n_adapt, flips=flips, moved=(rnew != rv).reshape(B, -1).sum(1), period=period, n_factor=n_factor))
            next_adapt = np.where(due, next_adapt + period, next_adapt)
            act_prev = np.where(due[:, None, None, None], act, act_prev)
            seen_act = seen_act | due
            changed = due & (rnew != rv).reshape(B, -1).any(1) & (n_factor <= P.max_refactor)
            if changed.any():
                if getattr(P, "trace", None) is not None:           # (tools: how many steps a re-factorisation really touches)
                    nst = (rnew != rv).any(axis=(2, 3)).sum(1)
                    P.trace.append((it + 1, n_factor[changed].copy(), nst[changed].copy()))
                rv = np.where(changed[:, None, None, None], rnew, rv)
                L, Na, V = fac(rv)                                   # model: refactor all
                n_factor += changed
        # (end of the iteration) secant step at a stopping test for the instances that go on
        if accel and aa_g is not None:
            if (it + 1) % P.check_every == 0:
                if aa_prev is not None and (it + 1) < P.max_iter:
                    d = aa_g - aa_prev
                    nx_ = x.size // B if aa_x_only else d.shape[1]
                    s1 = np.einsum("bn,bn->b", d[:, :nx_], aa_g[:, :nx_])
                    s2 = np.einsum("bn,bn->b", d[:, :nx_], d[:, :nx_])
                    with np.errstate(divide="ignore", invalid="ignore"):
                        gam = s1 / s2
                    okg = active & (s2 > 0) & (np.abs(gam) < 100.0) & (aa_prev_fac == n_factor_at_step) & (n_factor_at_step <= (16 if (aa_x_only and P.h > 20) else 8))
                    gam = np.where(okg, gam, 0.0).astype(rdt)
                    n1, n2 = x.size // B, z.size // B
                    x = x - gam[:, None, None, None] * aa_g[:, :n1].reshape(x.shape).astype(rdt)
                    z = np.clip(z - gam[:, None, None, None] * aa_g[:, n1:n1 + n2].reshape(z.shape).astype(rdt), lr, ur)
                    y = y - gam[:, None, None, None] * aa_g[:, n1 + n2:].reshape(y.shape).astype(rdt)
                aa_prev = None
            else:
                aa_prev, aa_prev_fac = aa_g, n_factor_at_step
    # controls in reference order [f1 f2 m1 m2]
    ctrl = np.concatenate([x[:, :, 0, 0:3], x[:, :, 1, 0:3], x[:, :, 0, 3:6], x[:, :, 1, 3:6]], -1)
    wr = np.einsum("bhfij,bhfj->bhi", Wr, x).astype(dtp)                  # wrench per step
    states = rollout(P, x_fb, Iw_inv, Rinv, wr, dt_)
    info = dict(iters=it_done, n_factor=n_factor)
    if return_debug:
        info.update(x_ref=x_ref, foot_ref=foot_ref, Gt=Gt, qt=qt, s=s, V=V, Iw_inv=Iw_inv, Rinv=Rinv, r=r,
                    rv=rv, x=x, z=z, y=y, l=l, u=u, A=A)
    return states, ctrl, info


def rollout(P, x_fb, Iw_inv, Rinv, wr, dt_):
    """X_i = A_i X_{i-1} + B_i U_i evaluated through the wrench (REF:203-216 semantics)."""
    B, h = x_fb.shape[0], P.h
    dtp = x_fb.dtype
    dt = dt_(P.dt)
    X = np.zeros((B, h, 13), dtp)
    prev = np.concatenate([x_fb, np.ones((B, 1), dtp)], -1)
    for i in range(h):
        cur = prev.copy()
        cur[:, 0:3] = prev[:, 0:3] + dt * np.einsum("bij,bj->bi", Rinv[:, i], prev[:, 6:9])
        cur[:, 3:6] = prev[:, 3:6] + dt * prev[:, 9:12]
        cur[:, 6:9] = prev[:, 6:9] + dt * np.einsum("bij,bj->bi", Iw_inv[:, i], wr[:, i, 0:3])
        cur[:, 9:12] = prev[:, 9:12] + (dt / dt_(P.m)) * wr[:, i, 3:6]
        cur[:, 11] -= dt_(P.g) * dt
        X[:, i] = cur
        prev = cur
    return X
