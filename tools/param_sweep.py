"""Solver-parameter sweep on the GPU (tuning aid, not part of the product path).

For every combination given on the command line (name=v1,v2,... ...) it reports, on oracle-solved
sample sets (npz files with x_fb, foot, contact, phase, x_cmd, mu, ref) the worst relative force
error (sets: tests/gen_tuning_sets.py), the instances that did not converge, mean iterations / factorisations, and the kernel time
of the BASELINE configs[1] batch (4096 standing instances, h = 10).

    python tools/param_sweep.py tests/golden/tuning kappa=10,30 adapt_every=10,15
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import biped_mpc_py_amd as bm           # noqa: E402
from biped_mpc_py_amd.synth import synth_batch  # noqa: E402


def load_sets(d):
    out = []
    for n in sorted(os.listdir(d)):
        if n.startswith("big_") and n.endswith(".npz"):
            z = np.load(os.path.join(d, n))
            out.append((n[4:-4], {k: z[k] for k in z.files}))
    return out


def main():
    d = sys.argv[1]
    grid = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        grid[k] = [float(x) if "." in x or "e" in x else int(x) for x in v.split(",")]
    sets = load_sets(d)
    _s = synth_batch(4096, 10, 1)
    xb, fb, cb, pb = _s["x_fb"], _s["foot"], _s["contact"], _s["phase"]
    keys = list(grid)
    print("%-40s %9s | " % ("options", "ms/4096") + " | ".join("%-22s" % n for n, _ in sets) + " | bench its/nf")
    for combo in itertools.product(*[grid[k] for k in keys]):
        opts = dict(zip(keys, combo))
        cols = []
        for name, z in sets:
            h = int(z["h"])
            mpc = bm.MPC()
            mpc.h = h
            s = bm.BatchSolver(mpc=mpc, half=int(z["half"]), max_batch=4096, solver_options=opts)
            mu = z["mu"] if z["mu"].size else None
            _, u, info = s.solve(z["x_fb"], z["foot"], z["contact"], z["phase"], x_cmd=z["x_cmd"], mu=mu, want_states=False)
            ref = z["ref"]
            rel = np.abs(u - ref).reshape(len(u), -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(len(u), -1).max(1))
            cols.append("%.1e nc%d %5.1f/%4.2f" % (rel.max(), int((info["status"] != 0).sum()), info["iters"].mean(), info["nfactor"].mean()))
            s.close()
        s = bm.BatchSolver(max_batch=4096, solver_options=opts)
        ms = []
        for _ in range(4):
            _, u, info = s.solve(xb, fb, cb, pb, want_states=False)
            ms.append(s.last_kernel_ms())
        print("%-40s %9.4f | " % (" ".join("%s=%g" % kv for kv in opts.items()), min(ms[1:])) + " | ".join(cols) +
              " | %5.1f/%4.2f nc%d" % (info["iters"].mean(), info["nfactor"].mean(), int((info["status"] != 0).sum())), flush=True)
        s.close()


if __name__ == "__main__":
    main()
