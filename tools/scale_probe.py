"""Every instance of a batch of one BASELINE config shape against the certified oracle, per installed library: the
distribution of the error (all controls and u0) and the iteration statistics.  References are cached under gpurun_out/.

    python tools/scale_probe.py --libs a.so,b.so --cases 2,4,3,5 [--n 8192] [--path 1]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {2: (8192, 10, "standing", 31, {}), 4: (8192, 10, "mixed", 3, dict(vx_cmd=True)),
         3: (4096, 16, "walking", 2, dict(vx_cmd=True)), 5: (4096, 20, "walking", 4, dict(vx_cmd=True, per_step_mu=True))}


def _one(a):
    from threadpoolctl import threadpool_limits
    from oracle import bmpc_oracle as orc
    with threadpool_limits(limits=1):
        x, f, c, xc, mu, h, half, ph = a
        mpc = orc.MPC()
        mpc.h = h
        mpc.x_cmd = xc
        _, ct, info = orc.solve_mpc(x, (ph + 0.5) * mpc.dt, f, mpc, orc.Biped(), c, half=half, mu_steps=mu, return_info=True)
        k = info["kkt"]
        return ct, bool(info["polished"]) and max(k["stationarity"], k["primal_ineq"], k["complementarity"]) <= 1e-7


def refs(case):
    import multiprocessing as mp
    from biped_mpc_py_amd.synth import synth_batch
    B, h, gait, seed, kw = CASES[case]
    s = synth_batch(B, h, seed, gait=gait, **kw)
    # (gpurun_out/ does not travel to the GPU box, build_tmp/ does: a cache copied there after a run is found again)
    path = os.path.join(ROOT, "gpurun_out", "scale_ref_%d.npz" % case)
    for cand in (os.path.join(ROOT, "build_tmp", "refs", os.path.basename(path)), path):
        if os.path.exists(cand):
            d = np.load(cand)
            return s, d["ref"], d["ok"]
    r32 = lambda v: v.astype(np.float32).astype(float)
    args = [(r32(s["x_fb"][i]), r32(s["foot"][i]), s["contact"][i], r32(s["x_cmd"][i]),
             None if s["mu"] is None else r32(s["mu"][i]), h, s["half"], int(s["phase"][i])) for i in range(B)]
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_one, args, chunksize=16)
    ref, ok = np.stack([r[0] for r in res]), np.array([r[1] for r in res])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, ref=ref, ok=ok)
    return s, ref, ok


def worker(case, path):
    import biped_mpc_py_amd as bm
    from tests import util
    s, ref, ok = refs(case)
    B, h = CASES[case][0], CASES[case][1]
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    e, e0 = util.rel_err(u, ref)[ok], util.u0_err(u, ref)[ok]
    w = int(np.argmax(util.rel_err(u, ref) * ok))
    print(json.dumps(dict(case=case, path=int(sol._lib.bmpc_solver_path(sol._h)), emax=float(e.max()), e999=float(np.quantile(e, 0.999)),
                          n1e5=int((e > 1e-5).sum()), n3e6=int((e > 3e-6).sum()), u0max=float(e0.max()), u0999=float(np.quantile(e0, 0.999)),
                          iters=float(info["iters"].mean()), imax=int(info["iters"].max()), nfac=float(info["nfactor"].mean()),
                          lost=int((info["status"] != 0).sum()), worst=w, worst_iters=int(info["iters"][w]))), flush=True)
    sol.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--cases", default="2,4,3,5")
    ap.add_argument("--path", type=int, default=1)
    ap.add_argument("--worker", type=int, default=0)
    a = ap.parse_args()
    if a.worker:
        worker(a.worker, a.path)
        sys.exit(0)
    cases = [int(c) for c in a.cases.split(",")]
    for c in cases:
        refs(c)
    for lib in a.libs.split(","):
        shutil.copy(os.path.join(ROOT, lib), os.path.join(ROOT, "biped_mpc_py_amd", "libbmpc.so"))
        for c in cases:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(c), "--path", str(a.path)],
                               capture_output=True, text=True, cwd=ROOT)
            ln = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if not ln:
                print("FAILED", lib, c, r.stderr[-1500:])
                continue
            d = json.loads(ln[-1])
            print("%-14s case %d path %d: err max %.2e p99.9 %.2e >1e-5: %d >3e-6: %d | u0 max %.2e p99.9 %.2e | iters %.2f max %d nfac %.2f lost %d | worst %d (%d its)" % (
                os.path.basename(lib), c, d["path"], d["emax"], d["e999"], d["n1e5"], d["n3e6"], d["u0max"], d["u0999"], d["iters"], d["imax"],
                d["nfac"], d["lost"], d["worst"], d["worst_iters"]), flush=True)
