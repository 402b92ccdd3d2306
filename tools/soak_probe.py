"""One soak shape on both kernel families and both penalty modes:  python tools/soak_probe.py h gait seed [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biped_mpc_py_amd as bm
from tests import util
h, gait, seed = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
kw = dict(vx_cmd=(gait != "standing"), per_step_mu=(h >= 20))
s = util.synth_batch(B, h, seed, gait=gait, **kw)
for path in (1, 2):
    if path == 1 and h > 20:
        continue
    for mode in (0, 1):
        m = bm.MPC(); m.h = h
        opts = dict(path=path, penalty_mode=mode)
        sol = bm.BatchSolver(mpc=m, half=s["half"], max_batch=B, solver_options=opts)
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
        sol.close()
        st = info["status"]
        print("h %d %s seed %d path %d mode %s: not converged %d (status 2: %d) iters mean %.1f max %d nfac %.2f nan %d" % (
            h, gait, seed, path, "scaled" if mode == 0 else "absolute", int((st != 0).sum()), int((st == 2).sum()), info["iters"].mean(),
            info["iters"].max(), info["nfactor"].mean(), int(np.isnan(u).any())), flush=True)
