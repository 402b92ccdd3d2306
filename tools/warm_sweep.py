"""Warm-start tuning on the GPU (not part of the product path): mean iterations per control period of a closed-loop
roll-out, cold against warm, over theta (penalty pull-back) and the first re-classification of a warm solve.
    python tools/warm_sweep.py [B] [K]"""
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import biped_mpc_py_amd as bm           # noqa: E402


def run(gait, B, K, warm, theta, was, shift):
    dev = torch.device("cuda", 0)
    mpc, biped = bm.MPC(), bm.Biped()
    rng = np.random.default_rng(5)
    x0 = np.zeros((B, 12), np.float32)
    x0[:, 5] = 0.55 + rng.uniform(-0.02, 0.02, B)
    x0[:, 0:3] = rng.uniform(-0.03, 0.03, (B, 3))
    x0[:, 9:12] = rng.uniform(-0.05, 0.05, (B, 3))
    foot = np.tile(np.array([-0.0195, 0.089, 0, -0.0195, -0.089, 0], np.float32), (B, 1))
    t0 = rng.uniform(0.0, 0.4, B)
    s = bm.BatchSolver(mpc=mpc, biped=biped, max_batch=B, solver_options=dict(warm_adapt_start=was))
    if warm:
        s.set_warm_start(True, shift=shift, theta=theta)
    x, t = torch.from_numpy(x0).to(dev), torch.from_numpy(t0).to(dev)
    r = s.rollout_device(x, torch.from_numpy(foot).to(dev), t, K, period=(10 if gait == "standing" else None),
                         duty=((10, 10) if gait == "standing" else None))
    torch.cuda.synchronize()
    it = r["iters"].cpu().numpy()
    bad = int((r["status_any"] != 0).sum())
    s.close()
    return it[1:].mean(), it[1:].max(), bad


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for gait in ("standing", "walking"):
        cold = run(gait, B, K, False, 0.5, 0, 0)
        print(f"{gait}: cold mean {cold[0]:.1f} max {cold[1]} bad {cold[2]}", flush=True)
        for theta, was, shift in itertools.product((0.25, 0.5, 0.75, 1.0), (0, 5), ((0,) if gait == "standing" else (0, 1))):
            w = run(gait, B, K, True, theta, was, shift)
            print(f"   theta {theta:4.2f} warm_adapt_start {was:2d} shift {shift}: mean {w[0]:.1f} ({w[0] / cold[0]:.2f}x) max {w[1]} bad {w[2]}", flush=True)
