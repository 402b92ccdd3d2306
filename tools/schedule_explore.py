#!/usr/bin/env python3
"""Re-classification schedules of the adaptive-penalty ADMM (bmpc_params.adapt_start / adapt_every / adapt_early / adapt_late).

Two modes over the same list of schedules (start, early period, early count, late period):

    python tools/schedule_explore.py model [sets]     CPU: the NumPy model of the product's algorithm (oracle/ws_model.py) on the
                                                      oracle-solved tuning sets; cost = iterations + c_h x factorisations, c_h the
                                                      measured cost of a factorisation in iterations (10.1 / 15.6 / 21.5 at h = 10 / 16 / 20)
    python tools/schedule_explore.py gpu [configs]    GPU: the library on the same sets (worst error against the oracle, lost
                                                      instances, counts) and the BASELINE batches' counts (for kernel TIMES use
                                                      tools/option_timing.py: device-resident, interleaved, several batches)

Round-5 finding (model, then confirmed on MI355X): the active set is found early -- of 240 rows 45 change class between
iterations 10 and 20, < 1 after iteration 40 -- so re-classifying 5 apart at first and 20 apart later takes fewer iterations AND
fewer factorisations than every 10.  Test infrastructure / tuning aid: the product never imports this."""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAC_COST = {10: 10.1, 16: 15.6, 20: 21.5}
# (start, every, early, late[, busy, flips, kappa_confirm, confirm_from]); early = 0: one rate (the schedule up to round 4);
# busy = 0: adapt_late always; kappa_confirm = 0: no confirmation.  First entry of a horizon: the round-4 default; second: round 5's.
SCHEDULES = {
    10: [(10, 10, 0, 0), (5, 5, 3, 20, 10, 1, 400.0, 3), (5, 5, 3, 20), (5, 5, 3, 20, 10, 1), (5, 5, 3, 20, 0, 1, 400.0, 3), (5, 5, 4, 20), (5, 5, 3, 15),
         (5, 5, 2, 15), (5, 10, 0, 0), (6, 6, 3, 20), (4, 4, 4, 20), (5, 5, 3, 20, 10, 1, 400.0, 0), (5, 5, 3, 20, 5, 0, 400.0, 3)],
    16: [(10, 20, 0, 0), (10, 10, 2, 20, 0, 1, 400.0, 2), (10, 10, 3, 20), (10, 10, 2, 20), (10, 5, 3, 20), (8, 8, 2, 20), (10, 10, 3, 30, 15, 1, 400.0, 3)],
    20: [(20, 20, 0, 0), (10, 10, 3, 20), (10, 10, 2, 30), (20, 10, 2, 20, 0, 1, 400.0, 2), (10, 10, 4, 30), (15, 15, 3, 30, 0, 1, 400.0, 3)],
}
RHO = {10: 0.03, 16: 0.03, 20: 0.045}


def load_sets():
    d = os.path.join(ROOT, "tests", "golden", "tuning")
    out = {}
    for n in sorted(os.listdir(d)):
        if n.startswith("big_") and n.endswith(".npz"):
            z = np.load(os.path.join(d, n))
            out[n[4:-4]] = {k: z[k] for k in z.files}
    return out


def _model_job(a):
    from threadpoolctl import threadpool_limits
    from oracle import ws_model as wm
    name, z, sched = a
    h = int(z["h"])
    P = wm.Params(h=h, half=int(z["half"]))
    sc = tuple(sched) + (0, 1, 0.0, 0)[len(sched) - 4:]
    P.adapt_start, P.adapt_every, P.adapt_early, P.adapt_late, P.adapt_busy, P.adapt_flips, P.kappa_confirm, P.confirm_from = sc
    P.rho = RHO[h]
    P.rho_eq_scale = 30.0 / P.rho
    P.slow_guard, P.max_iter = 1e-6, 400
    mu = z["mu"] if z["mu"].size else None
    with threadpool_limits(limits=1):
        _, u, info = wm.solve_batch(P, z["x_fb"], z["foot"], z["contact"], z["phase"], x_cmd=z["x_cmd"], mu=mu, dtype=np.float32, res_dtype=np.float64)
    ref = z["ref"]
    rel = np.abs(u - ref).reshape(len(u), -1).max(1) / np.maximum(1, np.abs(ref).reshape(len(u), -1).max(1))
    it, nf = info["iters"], info["n_factor"]
    cost = it + FAC_COST[h] * nf
    return name, sched, (it.mean(), nf.mean(), cost.mean(), np.percentile(cost, 95), cost.max(), int((it >= P.max_iter).sum()), rel.max())


def run_model(which):
    from multiprocessing import Pool
    sets = load_sets()
    jobs = [(n, z, s) for n, z in sets.items() if (not which or n in which) for s in SCHEDULES[int(z["h"])]]
    with Pool(min(8, os.cpu_count() or 1)) as p:
        res = p.map(_model_job, jobs, chunksize=1)
    for n in sets:
        rows = [r for r in res if r[0] == n]
        if not rows:
            continue
        print("set %s (h = %d, %d instances): start/every/early/late -> iterations, factorisations, cost mean / p95 / max, lost, worst error" % (n, int(sets[n]["h"]), len(sets[n]["ref"])))
        base = rows[0][2][2]
        for _, s, r in rows:
            print("  %-28s it %5.1f nf %4.2f cost %6.1f (%+5.1f %%) p95 %6.1f max %6.1f lost %d err %.1e" % ("/".join("%g" % v for v in s), r[0], r[1], r[2], 100 * (r[2] / base - 1), r[3], r[4], r[5], r[6]))


def run_gpu(configs):
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import synth
    sets = load_sets()
    for c in configs:
        cf = synth.CONFIGS[c]
        h = cf["h"]
        B = min(cf["batch"], 8192 if h == 20 else 4096)
        s = synth.synth_batch(B, h, cf["seed"], gait=cf["gait"], **cf["kw"])
        mpc = bm.MPC()
        mpc.h = h
        print("config %d (h = %d, B = %d): start/every/early/late -> kernel ms (min of 4), iterations / factorisations, lost | tuning sets of this horizon: worst error, lost, counts" % (c, h, B), flush=True)
        for sched in SCHEDULES[h]:
            sc = tuple(sched) + (0, 1, 0.0, 0)[len(sched) - 4:]
            opts = dict(zip(("adapt_start", "adapt_every", "adapt_early", "adapt_late", "adapt_busy", "adapt_flips", "kappa_confirm", "confirm_from"), sc))
            sv = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=opts)
            ms = []
            for _ in range(5):
                _, u, info = sv.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"] if cf["kw"].get("vx_cmd") else None, mu=s["mu"], want_states=False)
                ms.append(sv.last_kernel_ms())
            sv.close()
            cols = []
            for n, z in sets.items():
                if int(z["h"]) != h:
                    continue
                sv = bm.BatchSolver(mpc=mpc, half=int(z["half"]), max_batch=len(z["ref"]), solver_options=opts)
                mu = z["mu"] if z["mu"].size else None
                _, u2, i2 = sv.solve(z["x_fb"], z["foot"], z["contact"], z["phase"], x_cmd=z["x_cmd"], mu=mu, want_states=False)
                sv.close()
                ref = z["ref"]
                rel = np.abs(u2 - ref).reshape(len(u2), -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(len(u2), -1).max(1))
                cols.append("%s %.1e lost %d %5.1f/%4.2f" % (n, rel.max(), int((i2["status"] != 0).sum()), i2["iters"].mean(), i2["nfactor"].mean()))
            print("  %-28s %8.4f ms  %5.1f / %4.2f  lost %d  max it %d | %s" % ("/".join("%g" % v for v in sched), min(ms[1:]), info["iters"].mean(), info["nfactor"].mean(),
                                                                          int((info["status"] != 0).sum()), int(info["iters"].max()), " | ".join(cols)), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "model"
    if mode == "model":
        run_model(sys.argv[2].split(",") if len(sys.argv) > 2 else None)
    else:
        run_gpu([int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["2", "4"])])
