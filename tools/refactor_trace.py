#!/usr/bin/env python3
"""How many horizon steps does a re-factorisation really touch?  (CPU; the NumPy model of the product's algorithm.)

A re-classification changes the penalties of some rows, and with them the 6 x 6 blocks F_j of the steps those rows belong to; if
only one or two steps changed, V = (Gt + F)^-1 could follow by a rank-6 / rank-12 update instead of a new 6h x 6h sweep.  This
tool counts, per ordinal of the factorisation, the steps whose block changed, on the BASELINE shapes.  Round-4 answer: the rows
walk to their ceilings and floors in moves of kappa, so every re-factorisation but the last stragglers touches every step
(h = 10: 0.11 of 4.66 re-factorisations per solve touch <= 3 steps; h = 16 / 20: none) -- a low-rank update has nothing to work on.

Usage: python tools/refactor_trace.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from oracle import ws_model as wm
    from biped_mpc_py_amd import synth
    # (config: instances, re-classification period, first re-classification, rho0 -- bmpc_default_params of the horizon)
    for c, (B, every, start, rho) in {2: (192, 10, 10, 0.03), 3: (64, 20, 10, 0.03), 5: (48, 20, 20, 0.045)}.items():
        cf = synth.CONFIGS[c]
        h = cf["h"]
        s = synth.synth_batch(B, h, cf["seed"], gait=cf["gait"], **cf["kw"])
        P = wm.Params(h=h, half=s["half"])
        P.adapt_every, P.adapt_start, P.rho, P.rho_eq_scale, P.slow_guard = every, start, rho, 30.0 / rho, 1e-6
        P.trace = []
        _, _, info = wm.solve_batch(P, s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"] if cf["kw"].get("vx_cmd") else None,
                                    mu=s["mu"], dtype=np.float32, res_dtype=np.float64)
        print("config %d: iterations %.1f, factorisations %.2f" % (c, info["iters"].mean(), info["n_factor"].mean()))
        byord = {}
        for _, nf, nst in P.trace:
            for a, b in zip(nf, nst):
                byord.setdefault(int(a) + 1, []).append(int(b))
        few = 0
        for k in sorted(byord):
            v = np.array(byord[k])
            few += int((v <= 3).sum())
            print("  factorisation #%d: %.2f per solve, steps changed mean %.1f of %d, <= 3 steps in %.0f %%" % (k, len(v) / B, v.mean(), h, 100 * (v <= 3).mean()))
        print("  re-factorisations per solve %.2f, of which %.2f touch <= 3 steps" % (sum(len(v) for v in byord.values()) / B, few / B))


if __name__ == "__main__":
    main()
