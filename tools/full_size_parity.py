#!/usr/bin/env python3
"""BASELINE's FULL batch sizes against the certified fp64 oracle, every instance (GPU box).

tests/test_gpu_parity.py::test_parity_against_the_oracle_at_scale holds 8192 / 4096 instances per shape (what a test run can
afford); this tool takes the 65536-instance batches BASELINE.json names for configs[3] (h = 10, mixed gait schedules) and
configs[4] (h = 20, walking, per-step friction) -- the SAME seeded batches bench.py's `strong` record and `--config 5` solve --
through the product's default path and reports both metrics of SURVEY 8(d) over all of them.  The oracle runs on the box's
host cores (16 processes, ~1 / ~4 minutes for the two batches).

    python tools/full_size_parity.py [--configs 4,5] [--n 65536]      -> gpurun_out/full_size_parity.txt"""
import argparse
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _init():
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[k] = "1"


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="4,5")
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--libs", default="", help="library builds to install and evaluate in turn (default: the installed one)")
    ap.add_argument("--worker", type=int, default=0)
    ap.add_argument("--oracle-s", type=float, default=0.0)
    ap.add_argument("--cache", action="store_true", help="keep the reference chunks under gpurun_out/ (for a batch that needs two calls)")
    ap.add_argument("--budget", type=float, default=700.0, help="seconds after which no further reference chunk is started")
    a = ap.parse_args()
    if a.worker:
        return worker(a.worker, a.n, a.oracle_s)
    t_start = time.time()
    import multiprocessing as mp
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd.synth import CONFIGS, synth_batch
    from tests import util
    lines = ["# tools/full_size_parity.py, MI355X: BASELINE's full batch sizes, every instance against the certified fp64 oracle "
             "(inputs fp32-rounded as at the ABI), product default path"]
    for c in [int(v) for v in a.configs.split(",")]:
        cf = CONFIGS[c]
        h, B = cf["h"], a.n
        s = synth_batch(B, h, cf["seed"], gait=cf["gait"], **cf["kw"])
        mpc = bm.MPC()
        mpc.h = h
        r32 = lambda v: v.astype(np.float32).astype(float)
        args = [(r32(s["x_fb"][i]), r32(s["foot"][i]), s["contact"][i], r32(s["x_cmd"][i]),
                 None if s["mu"] is None else r32(s["mu"][i]), h, s["half"], int(s["phase"][i])) for i in range(B)]
        t0 = time.time()
        ref, ok = np.empty((B, h, 12)), np.zeros(B, bool)
        # the references in chunks, cached (gpurun_out/ comes back from the box, build_tmp/refs/ travels to it): a batch whose
        # oracle time exceeds one call's budget is finished by a second call
        CH = 8192
        missing = []
        for lo in range(0, B, CH):
            name = "fsp_ref_c%d_n%d_%d.npz" % (c, B, lo)
            for cand in (os.path.join(ROOT, "build_tmp", "refs", name), os.path.join(ROOT, "gpurun_out", name)):
                if os.path.exists(cand):
                    d = np.load(cand)
                    ref[lo:lo + CH], ok[lo:lo + CH] = d["ref"], d["ok"]
                    break
            else:
                missing.append(lo)
        t_or = 0.0
        if missing:
            with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1), initializer=_init) as pool:
                for lo in list(missing):
                    if time.time() - t_start > a.budget:
                        break
                    hi = min(lo + CH, B)
                    last = time.time()
                    for i, (ct, good) in enumerate(pool.imap(_one, args[lo:hi], chunksize=32)):
                        ref[lo + i], ok[lo + i] = ct, good
                        if time.time() - last > 30:
                            last = time.time()
                            print("config %d: oracle chunk at %d: %d / %d (%.0f s since start)" % (c, lo, i + 1, hi - lo, last - t_start), flush=True)
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    if a.cache:                 # (a 65536 x 20 batch is 126 MB of references: more than gpurun brings back)
                        np.savez(os.path.join(ROOT, "gpurun_out", "fsp_ref_c%d_n%d_%d.npz" % (c, B, lo)), ref=ref[lo:hi], ok=ok[lo:hi])
                    missing.remove(lo)
            t_or = time.time() - t0
        if missing:
            print("config %d: %d of %d reference chunks still missing (budget): copy gpurun_out/fsp_ref_* to build_tmp/refs/ and run again"
                  % (c, len(missing), -(-B // CH)), flush=True)
            continue
        # the solve itself in a child process per library (a process loads ONE libbmpc.so)
        tmp = "/tmp/fsp_ref_c%d.npz" % c
        np.savez(tmp, ref=ref, ok=ok)
        for lib in ([v for v in a.libs.split(",") if v] or [None]):
            if lib:
                shutil.copy(os.path.join(ROOT, lib), os.path.join(ROOT, "biped_mpc_py_amd", "libbmpc.so"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(c), "--n", str(B), "--oracle-s", "%.0f" % t_or],
                               capture_output=True, text=True, cwd=ROOT)
            out = [x for x in r.stdout.splitlines() if x.startswith(("config", "    instance"))]
            if r.returncode != 0 or not out:
                print("FAILED", lib, r.stderr[-2000:], flush=True)
                continue
            if lib:
                out = ["[%s]" % os.path.basename(lib)] + out
            print("\n".join(out), flush=True)
            lines += out
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "full_size_parity.txt"), "w") as fh:
                fh.write("\n".join(lines) + "\n")
        os.remove(tmp)


def worker(c, B, t_or):
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd.synth import CONFIGS, synth_batch
    from tests import util
    cf = CONFIGS[c]
    h = cf["h"]
    s = synth_batch(B, h, cf["seed"], gait=cf["gait"], **cf["kw"])
    d = np.load("/tmp/fsp_ref_c%d.npz" % c)
    ref, ok = d["ref"], d["ok"]
    mpc = bm.MPC()
    mpc.h = h
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    path = {1: "dense", 2: "stage"}[int(sol._lib.bmpc_solver_path(sol._h))]
    sol.close()
    e, e0 = util.rel_err(u, ref)[ok], util.u0_err(u, ref)[ok]
    st = info["status"]
    print("config %d (%s, h = %d), %d instances, path %s (%d references certified; oracle %.0f s on %d processes in this call): all controls max %.2e "
          "p99.9 %.2e above 1e-5: %d above 5e-5: %d | u0 max %.2e p99.9 %.2e above 2e-5: %d above 5e-5: %d | iterations %.2f (max %d), "
          "factorisations %.2f, not converged %d" % (
              c, cf["gait"], h, B, path, int(ok.sum()), t_or, min(16, os.cpu_count() or 1), e.max(), np.quantile(e, 0.999),
              int((e > 1e-5).sum()), int((e > 5e-5).sum()), e0.max(), np.quantile(e0, 0.999), int((e0 > 2e-5).sum()), int((e0 > 5e-5).sum()),
              info["iters"].mean(), int(info["iters"].max()), info["nfactor"].mean(), int((st != 0).sum())), flush=True)
    # the worst instances on the applied row, with what they looked like
    ef, e0f = util.rel_err(u, ref) * ok, util.u0_err(u, ref) * ok
    for i in np.argsort(-e0f)[:6]:
        d0 = u[i, 0] - ref[i, 0]
        k = int(np.argmax(np.abs(d0)))
        print("    instance %5d: u0 error %.2e (all controls %.2e), |u0_ref|_inf %.2f, largest deviation %.2e N(m) on component %d "
              "(reference %.4f); iterations %d, factorisations %d; contact at step 0 %s" % (
                  i, e0f[i], ef[i], np.abs(ref[i, 0]).max(), d0[k], k, ref[i, 0, k], info["iters"][i], info["nfactor"][i],
                  s["contact"][i, 0].tolist()), flush=True)


if __name__ == "__main__":
    main()
