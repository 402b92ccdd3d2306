"""Long-horizon shapes on the stage path with the ceilings scaled relative to h = 10 (A) or h = 20 (B = library default)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biped_mpc_py_amd as bm
from tests import util
def S2(h): return sum(k * k for k in range(1, h))
for h, gait, seed in ((36, "standing", 1301), (36, "standing", 1303), (32, "walking", 1303)):
    B = 8192
    kw = dict(vx_cmd=(gait != "standing"), per_step_mu=True)
    s = util.synth_batch(B, h, seed, gait=gait, **kw)
    for lab in ("A", "D", "E", "F"):
        opts = dict(path=2)
        if lab in ("A", "D", "E", "F"):
            phi = S2(h) / S2(10)
            top = 500.0
            rho0 = 0.045 * np.sqrt(phi)
            opts.update(penalty_mode=1, rho=rho0, rho_hi_f=min(phi, top), rho_hi_m=min(100 * phi, top), rho_eq_scale=min(30 * phi, top) / rho0)
            if lab == "D":
                opts.update(max_refactor=60, max_iter=1000)
            if lab == "E":
                opts.update(max_refactor=12)
            if lab == "F":
                opts.update(adapt_every=15, adapt_start=15)
        m = bm.MPC(); m.h = h
        sol = bm.BatchSolver(mpc=m, half=s["half"], max_batch=B, solver_options=opts)
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
        sol.close()
        st = info["status"]
        bad = np.flatnonzero(st != 0)
        print("h %d %s seed %d %s: not converged %d iters mean %.1f max %d nfac %.2f | nfac of the failures %s" % (
            h, gait, seed, lab, len(bad), info["iters"].mean(), info["iters"].max(), info["nfactor"].mean(), info["nfactor"][bad][:6]), "resid", info["residuals"][bad][:4].tolist(), "iters", info["iters"][bad][:6], flush=True)
