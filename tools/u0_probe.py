"""Worst instances of a scale_probe case on the u0 metric: what the row looks like (needs the cached references)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.scale_probe import CASES, refs
import biped_mpc_py_amd as bm
from tests import util

if __name__ == "__main__":
    case = int(sys.argv[1]); path = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    opts = dict(path=path)
    if len(sys.argv) > 3: opts.update(eps_pri=float(sys.argv[3]), eps_dua=float(sys.argv[3]))
    s, ref, ok = refs(case)
    B, h = CASES[case][0], CASES[case][1]
    mpc = bm.MPC(); mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=opts)
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    e0 = util.u0_err(u, ref) * ok
    e = util.rel_err(u, ref) * ok
    print("case", case, "path", path, opts, "u0 max %.2e all max %.2e iters %.2f" % (e0.max(), e.max(), info["iters"].mean()))
    np.set_printoptions(precision=5, suppress=True, linewidth=200)
    for i in np.argsort(-e0)[:4]:
        print("inst", i, "u0 err %.2e all err %.2e iters %d nfac %d resid %s |u|max %.1f |u0|max %.3f contact0 %s" % (
            e0[i], e[i], info["iters"][i], info["nfactor"][i], info["residuals"][i], np.abs(ref[i]).max(), np.abs(ref[i][0]).max(), s["contact"][i][0]))
        print("   u0  ", u[i][0]); print("   ref ", ref[i][0]); print("   diff", u[i][0] - ref[i][0])
