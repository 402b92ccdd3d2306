"""Every instance of the config-2 bench batch on both kernel families against the polished oracle (16 host cores)."""
import os, sys
import numpy as np
from multiprocessing import get_context
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def one(a):
    from threadpoolctl import threadpool_limits
    from oracle import bmpc_oracle as orc
    with threadpool_limits(limits=1):
        x, f, c = a
        _, ct, info = orc.solve_mpc(x, 0.02, f, orc.MPC(), orc.Biped(), c, return_info=True)
        return ct, info["polished"], info["kkt"]["stationarity"], info["n_active"]
if __name__ == "__main__":
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd.synth import synth_batch
    from tests import util
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-7
    s = synth_batch(B, 10, seed, gait="standing")
    out = {}
    for path in (1, 2):
        sol = bm.BatchSolver(max_batch=B, solver_options=dict(path=path, eps_pri=eps, eps_dua=eps))
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], want_states=False)
        out[path] = (u, info)
    args = [(s["x_fb"][i].astype(np.float32).astype(float), s["foot"][i].astype(np.float32).astype(float), s["contact"][i]) for i in range(B)]
    with get_context("spawn").Pool(16) as p:
        res = p.map(one, args, chunksize=16)
    ref = np.stack([r[0] for r in res])
    e1, e2 = util.rel_err(out[1][0], ref), util.rel_err(out[2][0], ref)
    print("B %d seed %d eps %g: dense: max %.2e p99.9 %.2e > 5e-5: %d > 1e-4: %d | stage: max %.2e p99.9 %.2e > 5e-5: %d > 1e-4: %d | iters %.1f" % (
        B, seed, eps, e1.max(), np.quantile(e1, 0.999), (e1 > 5e-5).sum(), (e1 > 1e-4).sum(), e2.max(), np.quantile(e2, 0.999), (e2 > 5e-5).sum(), (e2 > 1e-4).sum(),
        out[1][1]["iters"].mean()))
    for i in np.argsort(-e1)[:5]:
        print("inst %4d dense %.2e (iters %d resid %s) stage %.2e (iters %d) oracle polished %s stat %.1e active %d" % (
            i, e1[i], out[1][1]["iters"][i], out[1][1]["residuals"][i], e2[i], out[2][1]["iters"][i], res[i][1], res[i][2], res[i][3]))
