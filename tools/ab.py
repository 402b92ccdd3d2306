"""A/B of libbmpc.so builds on ONE GPU box (boxes differ by 1-6 %; run-to-run noise on a box is ~0.3 %): kernel time,
iteration statistics and -- optionally -- parity against the oracle, per BASELINE config shape, the variants interleaved.

    python tools/ab.py --libs build_tmp/a.so,build_tmp/b.so --configs 2,3,5 --reps 3 [--parity 512] [--path 0|1|2]

Every measurement runs in its own process (the library is loaded once per process); the parent copies the variant over
biped_mpc_py_amd/libbmpc.so, like tools/ab_build.sh.  The oracle references of --parity are solved once (all host cores)
and cached under gpurun_out/.  Leaves the LAST variant installed on the box (the box is thrown away after the call)."""
import argparse
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _oracle_one(a):
    from threadpoolctl import threadpool_limits
    from oracle import bmpc_oracle as orc
    with threadpool_limits(limits=1):
        x, f, c, xc, mu, h, half, ph = a
        mpc = orc.MPC()
        mpc.h = h
        mpc.x_cmd = xc
        t = ph * mpc.dt + 0.5 * mpc.dt
        _, ct, info = orc.solve_mpc(x, t, f, mpc, orc.Biped(), c, half=half, mu_steps=mu, return_info=True)
        k = info["kkt"]
        return ct, bool(info["polished"]) and max(k["stationarity"], k["primal_ineq"], k["complementarity"]) <= 1e-7


def references(cfg, n):
    """Oracle controls of the first n instances of the config's batch on the fp32-rounded inputs the GPU sees."""
    import multiprocessing as mp
    from biped_mpc_py_amd import synth
    # (gpurun_out/ does not travel to the GPU box, build_tmp/ does: a cache copied there after a run is found again)
    path = os.path.join(ROOT, "gpurun_out", "ab_ref_cfg%d_%d.npz" % (cfg, n))
    for cand in (os.path.join(ROOT, "build_tmp", "refs", os.path.basename(path)), path):
        if os.path.exists(cand):
            d = np.load(cand)
            return d["ref"], d["ok"]
    c = synth.CONFIGS[cfg]
    B = 4096 if cfg != 5 else 8192
    s = synth.synth_batch(B, c["h"], c["seed"], gait=c["gait"], **c["kw"])
    r32 = lambda v: v.astype(np.float32).astype(float)
    args = [(r32(s["x_fb"][i]), r32(s["foot"][i]), s["contact"][i], r32(s["x_cmd"][i]),
             None if s["mu"] is None else r32(s["mu"][i]), c["h"], s["half"], int(s["phase"][i])) for i in range(n)]
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        res = pool.map(_oracle_one, args, chunksize=8)
    ref = np.stack([r[0] for r in res])
    ok = np.array([r[1] for r in res])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, ref=ref, ok=ok)
    return ref, ok


def worker(cfg, path, parity, launches):
    import torch
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import synth
    dev = torch.device("cuda:0")
    c = synth.CONFIGS[cfg]
    B = 4096 if cfg != 5 else 8192
    s = synth.synth_batch(B, c["h"], c["seed"], gait=c["gait"], **c["kw"])
    t = {k: (None if s[k] is None else torch.from_numpy(np.ascontiguousarray(s[k].astype(np.float32) if s[k].dtype == np.float64 else s[k])).to(dev))
         for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
    mpc = bm.MPC()
    mpc.h = c["h"]
    sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
    it = torch.empty(B, dtype=torch.int32, device=dev)
    nf = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    u = torch.empty((B, c["h"], 12), dtype=torch.float32, device=dev)
    ms = []
    for _ in range(launches + 2):
        sol.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], t["x_cmd"], t["mu"], controls=u, iters=it, nfactor=nf, status=st)
        torch.cuda.synchronize()
        ms.append(sol.last_kernel_ms())
    out = dict(cfg=cfg, ms_median=float(np.median(ms[2:])), ms_min=float(min(ms[2:])), iters=float(it.float().mean()),
               iters_max=int(it.max()), nfac=float(nf.float().mean()), not_converged=int((st != 0).sum()),
               path=int(sol._lib.bmpc_solver_path(sol._h)))
    if parity:
        ref, ok = references(cfg, parity)
        uu = u[:parity].cpu().numpy().astype(float)
        e = np.abs(uu - ref).reshape(parity, -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(parity, -1).max(1))
        e0 = np.abs(uu[:, 0] - ref[:, 0]).max(1) / np.maximum(1.0, np.abs(ref[:, 0]).max(1))
        out.update(err_max=float(e[ok].max()), err_p99=float(np.quantile(e[ok], 0.99)), u0_err_max=float(e0[ok].max()), certified=int(ok.sum()))
    sol.close()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--configs", default="2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--parity", type=int, default=0)
    ap.add_argument("--path", type=int, default=0)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--worker", type=int, default=0)
    a = ap.parse_args()
    if a.worker:
        return worker(a.worker, a.path, a.parity, a.launches)
    cfgs = [int(c) for c in a.configs.split(",")]
    if a.parity:
        for c in cfgs:
            references(c, a.parity)
    libs = a.libs.split(",")
    rows = {}
    for rep in range(a.reps):
        for lib in libs:
            shutil.copy(os.path.join(ROOT, lib), os.path.join(ROOT, "biped_mpc_py_amd", "libbmpc.so"))
            for c in cfgs:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(c), "--path", str(a.path),
                                    "--parity", str(a.parity if rep == 0 else 0), "--launches", str(a.launches)],
                                   capture_output=True, text=True, cwd=ROOT)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    print("FAILED", lib, c, r.stderr[-2000:], flush=True)
                    continue
                d = json.loads(line[-1])
                rows.setdefault((lib, c), []).append(d)
                extra = "" if "err_max" not in d else "  err max %.2e p99 %.2e u0 %.2e (%d certified)" % (d["err_max"], d["err_p99"], d["u0_err_max"], d["certified"])
                print("%-28s cfg %d path %d: %.4f ms (min %.4f)  iters %.2f max %d  nfac %.2f  lost %d%s" % (
                    os.path.basename(lib), c, d["path"], d["ms_median"], d["ms_min"], d["iters"], d["iters_max"], d["nfac"], d["not_converged"], extra), flush=True)
    print("---- medians over the repetitions")
    for (lib, c), v in rows.items():
        print("%-28s cfg %d: %.4f ms" % (os.path.basename(lib), c, float(np.median([d["ms_median"] for d in v]))))


if __name__ == "__main__":
    main()
