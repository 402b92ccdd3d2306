"""Regenerates tests/golden/tuning/big_*.npz: 768 random instances of the BASELINE config shapes (SURVEY 8(d)
generator, tests/util.synth_batch) solved by the fp64 oracle (oracle/bmpc_oracle.solve_mpc, KKT-certified).
Inputs are rounded to fp32 first, because that is what crosses the C ABI.  About a minute on 8 cores:

    python tests/gen_tuning_sets.py [out_dir]

The sets serve two purposes: a wider parity net than the fixtures (tests/test_gpu_parity.py::
test_oracle_solved_sets) and the objective of tools/param_sweep.py when solver parameters are tuned on a GPU.
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bmpc_oracle as orc     # noqa: E402
from tests import util                    # noqa: E402

CFGS = [("big_stand10", 10, "standing", 11, {}, 256),
        ("big_mixed10", 10, "mixed", 13, dict(vx_cmd=True), 256),
        ("big_walk16", 16, "walking", 12, dict(vx_cmd=True), 128),
        ("big_walk20", 20, "walking", 14, dict(vx_cmd=True, per_step_mu=True), 128)]


def solve_one(args):
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1):          # one BLAS thread per worker process
        return _solve_one(args)


def _solve_one(args):
    h, half, x, f, c, ph, xc, mu = args
    m, b = orc.MPC(), orc.Biped()
    m.h = h
    m.x_cmd = xc
    t = ph * m.dt + 0.5 * m.dt
    _, ct, info = orc.solve_mpc(x, t, f, m, b, c, half=half, mu_steps=mu, return_info=True)
    k = info["kkt"]
    return ct, np.array([k["stationarity"], k["primal_eq"], k["primal_ineq"], k["dual"], k["complementarity"]]), int(info["polished"])


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "tuning")
    os.makedirs(out, exist_ok=True)
    for name, h, gait, seed, kw, B in CFGS:
        s = util.synth_batch(B, h, seed, gait=gait, **kw)
        x32 = s["x_fb"].astype(np.float32).astype(float)
        f32 = s["foot"].astype(np.float32).astype(float)
        mu32 = None if s["mu"] is None else s["mu"].astype(np.float32).astype(float)
        args = [(h, s["half"], x32[i], f32[i], s["contact"][i], int(s["phase"][i]), s["x_cmd"][i],
                 None if mu32 is None else mu32[i]) for i in range(B)]
        t0 = time.time()
        with Pool(min(8, os.cpu_count() or 1)) as p:
            res = p.map(solve_one, args)
        ref = np.stack([o[0] for o in res])
        kkt = np.stack([o[1] for o in res])
        polished = np.array([o[2] for o in res], np.int32)
        print(name, "oracle time %.0f s" % (time.time() - t0))
        np.savez(os.path.join(out, name + ".npz"), ref=ref, x_fb=x32, foot=f32, contact=s["contact"], phase=s["phase"],
                 x_cmd=s["x_cmd"], mu=(np.zeros(0) if mu32 is None else mu32), h=h, half=s["half"], kkt=kkt, polished=polished)
