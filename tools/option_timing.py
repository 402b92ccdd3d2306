"""Kernel time of BASELINE config shapes under solver options, on one box: python tools/option_timing.py 2,3,5 "" adapt_every=20,adapt_start=20 rho=0.045 ...
(each argument after the config list is one comma-separated set of bmpc_params overrides; "" = defaults).  Three interleaved
rounds; prints the median kernel ms per round, iterations, factorisations, instances not converged.
BMPC_OT_SEEDS=a,b,c: the config's shape with these generator seeds instead of its own (the time of a 4096-instance launch depends on
where its longest instances sit in the dispatch order by +-3 %: compare schedules over several batches); BMPC_OT_BATCH=n: batch size."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import biped_mpc_py_amd as bm
from biped_mpc_py_amd import synth
dev = torch.device("cuda:0")
cfgs = [int(a) for a in sys.argv[1].split(",")]
def parse(o):
    d = {}
    for kv in o.split(","):
        if not kv: continue
        k, v = kv.split("=")
        d[k] = float(v) if "." in v or "e" in v else int(v)
    return d
opts = [parse(o) for o in sys.argv[2:]]
seeds = [int(x) for x in os.environ.get("BMPC_OT_SEEDS", "").split(",") if x]
runs = [(cfg, sd) for cfg in cfgs for sd in (seeds or [None])]
for cfg, sd in runs:
    c = synth.CONFIGS[cfg]; B = int(os.environ.get("BMPC_OT_BATCH", 4096 if cfg != 5 else 8192))
    s = synth.synth_batch(B, c["h"], c["seed"] if sd is None else sd, gait=c["gait"], **c["kw"])
    t = {k: (None if s[k] is None else torch.from_numpy(np.ascontiguousarray(s[k].astype(np.float32) if s[k].dtype == np.float64 else s[k])).to(dev)) for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
    mpc = bm.MPC(); mpc.h = c["h"]
    res = {}
    for rnd in range(3):
        for o in opts:
            sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=o)
            it = torch.empty(B, dtype=torch.int32, device=dev); nf = torch.empty(B, dtype=torch.int32, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev)
            ms = []
            for _ in range(6):
                sol.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"], t["x_cmd"], t["mu"], iters=it, nfactor=nf, status=st)
                torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
            res.setdefault(str(o), []).append(np.median(ms[1:]))
            info = (it.float().mean().item(), it.max().item(), nf.float().mean().item(), int((st != 0).sum().item()))
            sol.close()
            if rnd == 2: print("cfg", cfg, "seed", sd, o, "kernel ms", ["%.3f" % x for x in res[str(o)]], "iters %.1f max %d nfac %.2f notconv %d" % info)
