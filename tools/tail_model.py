#!/usr/bin/env python3
"""What would cutting the tail of a 4096-instance batch buy?  

A batch of 4096 instances runs on 1024 instance slots (4 per CU x 256 CUs); workgroups are dispatched in index order as
slots free up, an instance's duration varies with its iteration and factorisation counts (35-105 iterations at h = 10),
and the end of the launch waits for the last instances.  This tool measures the per-instance counts of BASELINE config C on
the GPU, calibrates a small event model of the dispatch (per-CU occupancy slows an instance down: the measured occupancy
curve of docs/history_r04.md), and evaluates on the SAME instances:

    as dispatched           the model of today's launch (calibrated to the measured kernel time)
    perfect packing         total work / slots: the floor any reordering or splitting could reach
    longest first (oracle)  dispatch sorted by the true cost (not available to a cold solve)
    cap + continuation      VERDICT's proposal: a first launch capped at K iterations, the unfinished instances compacted
                            and continued from their stored state in a second launch.  A continuation pays for what a second
                            launch cannot keep: the set-up (references, step data, Hessian rows: 36 k cycles) and one
                            factorisation unless the instance was about to re-factor anyway (33.4 k cycles) -- V lives
                            in registers.  Variants: both paid / only the set-up paid (V and the step data saved to HBM,
                            ~35 KB per instance) / nothing paid (an upper bound of the idea).

Usage: python tools/tail_model.py [config] [batch]      (GPU box; prints a table, writes gpurun_out/tail_model_cfgC.txt)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SLOTS_PER_CU, CUS = 4, 256
# relative speed of an instance when k instances share its CU (docs/history_r04.md: 1 / 2 / 3 / 4 per CU -> 2.33 / 1.37 /
# 1.05 / 0.90 ms per 4096, i.e. per-instance latency 0.146 / 0.171 / 0.197 / 0.225 ms), normalised to 4 per CU
SPEED = np.array([0.0, 0.225 / 0.146, 0.225 / 0.171, 0.225 / 0.197, 1.0])


def simulate(cost, order=None, dt=2.0):
    """Makespan (in the units of `cost` at full occupancy) of dispatching the jobs in `order` onto CUS x SLOTS_PER_CU slots:
    time-stepped, a job on a CU with k resident jobs progresses at SPEED[k]."""
    cost = np.asarray(cost, float)
    order = np.arange(len(cost)) if order is None else np.asarray(order)
    rem = np.zeros((CUS, SLOTS_PER_CU))
    nxt, t, n = 0, 0.0, len(order)
    while True:
        free = rem <= 0
        if nxt < n and free.any():
            cu, sl = np.nonzero(free)
            # the dispatcher fills free slots CU by CU, least loaded CUs first
            load = (~free).sum(1)[cu]
            idx = np.argsort(load, kind="stable")
            k = min(n - nxt, len(idx))
            rem[cu[idx[:k]], sl[idx[:k]]] = cost[order[nxt:nxt + k]]
            nxt += k
        busy = rem > 0
        if not busy.any():
            return t
        kk = busy.sum(1)
        step = dt
        # advance to the next completion if that is sooner than dt
        with np.errstate(divide="ignore", invalid="ignore"):
            need = np.where(busy, rem / SPEED[kk][:, None], np.inf)
        step = min(dt, float(need.min()))
        rem = np.where(busy, rem - step * SPEED[kk][:, None], rem)
        rem[np.abs(rem) < 1e-9] = 0.0
        t += step


def main():
    import torch
    from biped_mpc_py_amd import MPC, BatchSolver, synth
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    c = synth.CONFIGS[cfg]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    d = synth.synth_batch(B, c["h"], c["seed"], gait=c["gait"], **c["kw"])
    m = MPC()
    m.h = c["h"]
    s = BatchSolver(mpc=m, half=d["half"], max_batch=B)
    dev = torch.device("cuda:0")
    t = {k: (None if d[k] is None else torch.from_numpy(np.ascontiguousarray(d[k].astype(np.float32) if d[k].dtype == np.float64 else d[k])).to(dev))
         for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
    it = torch.empty(B, dtype=torch.int32, device=dev)
    nf = torch.empty(B, dtype=torch.int32, device=dev)
    ms = []
    for _ in range(12):
        s.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], t["x_cmd"], t["mu"], iters=it, nfactor=nf)
        torch.cuda.synchronize()
        ms.append(s.last_kernel_ms())
    measured = float(np.median(ms[2:]))
    it, nf = it.cpu().numpy().astype(float), nf.cpu().numpy().astype(float)
    every = float(s.cparams.adapt_every)
    SETUP, FACTOR, ITER = 36.0, 33.4, 3.3          # k cycles per instance at 4 per CU (profiles/r03_cfg2_phase_cycles.txt)
    if c["h"] != 10:
        SETUP, FACTOR, ITER = {16: (95.0, 120.0, 7.7), 20: (170.0, 260.0, 12.0)}.get(c["h"], (SETUP, FACTOR, ITER))
    cost = SETUP + FACTOR * nf + ITER * it
    base = simulate(cost)
    scale = measured / base                          # ms per k-cycle unit of the model
    lines = [f"config {cfg}, {B} instances, h = {c['h']}: measured kernel {measured:.4f} ms; iterations {it.mean():.1f} "
             f"(min {it.min():.0f}, max {it.max():.0f}), factorisations {nf.mean():.2f}; cost per instance {cost.mean():.0f} k cycles "
             f"(max {cost.max():.0f})",
             f"{'as dispatched (model, calibrated)':52s} {base * scale:.4f} ms  1.000"]

    def row(name, v):
        lines.append(f"{name:52s} {v * scale:.4f} ms  {v / base:.3f}")
    row("perfect packing (sum of work / slots, full speed)", cost.sum() / (CUS * SLOTS_PER_CU))
    row("longest first by the true cost (oracle order)", simulate(cost, np.argsort(-cost, kind="stable")))
    for K in (40, 50, 60, 70):
        # factorisations an instance has had by iteration K: the first one + one per re-classification point before K
        nf1 = np.minimum(nf, 1 + np.floor((K - 1e-9) / every))
        first = SETUP + FACTOR * nf1 + ITER * np.minimum(it, K)
        go = it > K
        rest_it, rest_nf = (it - K)[go], (nf - nf1)[go]
        # a re-classification falls on iteration K if K is a multiple of the period: then the factorisation is due anyway
        due = (K % int(every) == 0)
        for name, c_setup, c_fac in (("set-up and one factorisation paid again", SETUP, FACTOR),
                                     ("only the set-up paid again (V saved to HBM)", SETUP, 0.0),
                                     ("nothing paid again (upper bound)", 0.0, 0.0)):
            extra_fac = np.where((rest_nf > 0) & due, 0.0, c_fac)         # (an instance about to re-factor pays nothing extra)
            second = c_setup + extra_fac + FACTOR * rest_nf + ITER * rest_it
            total = simulate(first) + (simulate(second) if go.any() else 0.0)
            row(f"cap {K} + continuation ({go.mean() * 100:.0f} % go on): {name}", total)
    out = "\n".join(lines)
    print(out)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/tail_model_cfg{cfg}.txt", "w") as fh:
        fh.write(out + "\n")


if __name__ == "__main__":
    main()
