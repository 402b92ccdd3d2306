"""One-off validation (run on a GPU box): the HARDEST instances of large random batches -- the ones that take the
most iterations -- against the fp64 oracle.  python tests/check_hard_instances.py [n_per_config]"""
import os
import sys
from multiprocessing import get_context

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util                    # noqa: E402


def solve_one(args):
    from oracle import bmpc_oracle as orc
    h, half, x, f, c, ph, xc, mu = args
    m, b = orc.MPC(), orc.Biped()
    m.h = h
    m.x_cmd = xc
    _, ct = orc.solve_mpc(x, ph * m.dt + 0.5 * m.dt, f, m, b, c, half=half, mu_steps=mu)
    return ct


if __name__ == "__main__":
    import biped_mpc_py_amd as bm
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    worst = 0.0
    for h, gait, seed, kw in ((10, "mixed", 501, dict(vx_cmd=True)), (10, "standing", 502, {}),
                              (16, "walking", 503, dict(vx_cmd=True)), (20, "walking", 504, dict(vx_cmd=True, per_step_mu=True))):
        B = 16384
        s = util.synth_batch(B, h, seed, gait=gait, **kw)
        mpc = bm.MPC()
        mpc.h = h
        sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
        _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
        sol.close()
        idx = np.argsort(-info["iters"])[:n]
        x32 = s["x_fb"].astype(np.float32).astype(float)
        f32 = s["foot"].astype(np.float32).astype(float)
        mu32 = None if s["mu"] is None else s["mu"].astype(np.float32).astype(float)
        args = [(h, s["half"], x32[i], f32[i], s["contact"][i], int(s["phase"][i]), s["x_cmd"][i],
                 None if mu32 is None else mu32[i]) for i in idx]
        with get_context("spawn").Pool(16) as p:
            ref = np.stack(p.map(solve_one, args))
        rel = np.abs(u[idx] - ref).reshape(n, -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(n, -1).max(1))
        worst = max(worst, rel.max())
        print("h=%d %s: hardest %d of %d (iterations %d..%d): max rel err %.2e, median %.2e" %
              (h, gait, n, B, info["iters"][idx].min(), info["iters"][idx].max(), rel.max(), np.median(rel)), flush=True)
    print("worst", worst, "tolerance", util.REL_TOL)
    sys.exit(0 if worst <= util.REL_TOL else 1)
