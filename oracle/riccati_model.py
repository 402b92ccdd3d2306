"""NumPy model of the stage-structured (Riccati) wrench-space solve of the long-horizon kernels (TEST INFRASTRUCTURE ONLY).

SURVEY 8(f) row 4: the dense kernels invert K' = Gt + F (6h x 6h) explicitly, O(h^3) work and O(h^2) state.  But
(Gt + F) gamma = beta is the optimality system of an LQ problem in the 12 SRBM states (REF:165-184, 203-216),

    min sum_i  1/2 gamma_i' F_i gamma_i - beta_i' gamma_i + 1/2 xi_i' (2Q) xi_i,     xi_i = A_i xi_{i-1} + Bh_i gamma_i,  xi_{-1} = 0,

with A_i = [[I, C_i], [0, I]], C_i = dt blkdiag(Rinv_i, I) (REF:165-171, 183) and Bh_i = [0; E_i], E_i = dt blkdiag(Iw_i^-1, I/m)
(REF:174-180, 184), so a backward Riccati recursion factorises it in h steps of 6x6 / 6x12 blocks and a solve is one
backward and one forward pass over the steps.  This file states that recursion the way the kernels run it (in the
acceleration variables a = E gamma, with the cancellation-free form of the Schur complement) so that tests can compare
it with the dense inverse of ws_model and measure what f32 does to it.  Only tests/ and tools/ import it.
"""
from __future__ import annotations

import numpy as np


def stage_matrices(P, Iw_inv, Rinv, dtp):
    """C (B,h,6,6) and E (B,h,6,6) of the stage dynamics above."""
    B, h = Rinv.shape[:2]
    dt = dtp.type(P.dt)
    C = np.zeros((B, h, 6, 6), dtp)
    C[:, :, 0:3, 0:3] = dt * Rinv
    C[:, :, 3:6, 3:6] = dt * np.eye(3, dtype=dtp)
    E = np.zeros((B, h, 6, 6), dtp)
    E[:, :, 0:3, 0:3] = dt * Iw_inv
    E[:, :, 3:6, 3:6] = (dt / dtp.type(P.m)) * np.eye(3, dtype=dtp)
    return C, E


def factor(P, F, Iw_inv, Rinv, dtype=np.float64):
    """Backward Riccati recursion.  F (B,h,6,6) wrench-space stage cost.  Returns dict(K (B,h,6,12), Sinv (B,h,6,6), C, E)
    in the acceleration variables: a_i = E_i gamma_i = -K_i xi_{i-1} - Sinv_i (p2_{i+1} - bt_i), bt = E^-T beta."""
    dtp = np.dtype(dtype)
    B, h = F.shape[:2]
    C, E = stage_matrices(P, Iw_inv.astype(dtp), Rinv.astype(dtp), dtp)
    Einv = np.linalg.inv(E.astype(np.float64)).astype(dtp)
    Ft = np.einsum("bhji,bhjk,bhkl->bhil", Einv, F.astype(dtp), Einv)          # E^-T F E^-1
    Q2 = (2 * P.Q[:12]).astype(dtp)
    P11 = np.zeros((B, 6, 6), dtp)
    P12 = np.zeros((B, 6, 6), dtp)
    P22 = np.zeros((B, 6, 6), dtp)
    K = np.zeros((B, h, 6, 12), dtp)
    Sinv = np.zeros((B, h, 6, 6), dtp)
    for i in range(h - 1, -1, -1):
        Pi11 = P11 + np.diag(Q2[0:6])
        Pi12 = P12
        Pi22 = P22 + np.diag(Q2[6:12])
        S = Ft[:, i] + Pi22
        Si = np.linalg.inv(S.astype(np.float64)).astype(dtp) if dtp == np.float64 else _inv_spd(S)
        Sinv[:, i] = Si
        Ci = C[:, i]
        M1 = np.swapaxes(Pi12, -1, -2)                         # Pi21
        M2 = M1 @ Ci + Pi22                                    # [Pi21, Pi21 C + Pi22] = B' Pi A
        K[:, i, :, 0:6] = Si @ M1
        K[:, i, :, 6:12] = Si @ M2
        # Schur complement Pi - Pi[:,2] S^-1 Pi[2,:] without cancellation in the (., 2) blocks: I - S^-1 Pi22 = S^-1 Ft
        T = Si @ Ft[:, i]                                      # S^-1 Ft
        Z11 = Pi11 - Pi12 @ Si @ M1
        Z12 = Pi12 @ T
        Z22 = Pi22 @ T                                         # = Pi22 - Pi22 S^-1 Pi22
        Z22 = 0.5 * (Z22 + np.swapaxes(Z22, -1, -2))
        # P = A' Z A,  A = [[I, C], [0, I]]
        P11 = Z11
        P12 = Z11 @ Ci + Z12
        P22 = np.swapaxes(Ci, -1, -2) @ (Z11 @ Ci + Z12) + np.swapaxes(Z12, -1, -2) @ Ci + Z22
        P11 = 0.5 * (P11 + np.swapaxes(P11, -1, -2))
        P22 = 0.5 * (P22 + np.swapaxes(P22, -1, -2))
    return dict(K=K, Sinv=Sinv, C=C, E=E, Einv=Einv)


def _inv_spd(S):
    """6x6 SPD inverse in the array's own precision (Gauss-Jordan without pivoting, as a lane would do it)."""
    dtp = S.dtype
    n = S.shape[-1]
    A = S.copy()
    Inv = np.broadcast_to(np.eye(n, dtype=dtp), S.shape).copy()
    for k in range(n):
        piv = (dtp.type(1) / A[:, k, k])[:, None]
        rowA = A[:, k, :] * piv
        rowI = Inv[:, k, :] * piv
        colk = A[:, :, k].copy()
        A = A - colk[:, :, None] * rowA[:, None, :]
        Inv = Inv - colk[:, :, None] * rowI[:, None, :]
        A[:, k, :] = rowA
        Inv[:, k, :] = rowI
    return Inv


def solve(fac, beta):
    """gamma (B,h,6) with (Gt + F) gamma = beta (B,h,6), and the state deviation xi = Gam_t gamma (B,h,12)."""
    K, Sinv, C, E, Einv = fac["K"], fac["Sinv"], fac["C"], fac["E"], fac["Einv"]
    dtp = K.dtype
    B, h = beta.shape[:2]
    bt = np.einsum("bhji,bhj->bhi", Einv, beta.astype(dtp))                    # E^-T beta
    p1 = np.zeros((B, 6), dtp)
    p2 = np.zeros((B, 6), dtp)
    g = np.zeros((B, h, 6), dtp)
    for i in range(h - 1, -1, -1):
        g[:, i] = p2 - bt[:, i]
        # p_i = A' p_{i+1} - K_i' g_i
        kg = np.einsum("bij,bi->bj", K[:, i], g[:, i])
        n1 = p1 - kg[:, 0:6]
        n2 = np.einsum("bji,bj->bi", C[:, i], p1) + p2 - kg[:, 6:12]
        p1, p2 = n1, n2
    w = np.einsum("bhij,bhj->bhi", Sinv, g)
    xi = np.zeros((B, h, 12), dtp)
    a = np.zeros((B, h, 6), dtp)
    x1 = np.zeros((B, 6), dtp)
    x2 = np.zeros((B, 6), dtp)
    for i in range(h):
        a[:, i] = -np.einsum("bij,bj->bi", K[:, i], np.concatenate([x1, x2], -1)) - w[:, i]
        x1 = x1 + np.einsum("bij,bj->bi", C[:, i], x2)
        x2 = x2 + a[:, i]
        xi[:, i, 0:6] = x1
        xi[:, i, 6:12] = x2
    gamma = np.einsum("bhij,bhj->bhi", Einv, a)
    return gamma, xi
