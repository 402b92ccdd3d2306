"""GPU probe of the two kernel families (dense inverse / stage-structured): parity on the fixtures of every horizon,
kernel time, iteration counts and the in-kernel cycle stamps.   python tools/stage_probe.py [parity] [time B h,h,..] [prof B h]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import biped_mpc_py_amd as bm                      # noqa: E402
from biped_mpc_py_amd import _lib                  # noqa: E402
from biped_mpc_py_amd.synth import synth_batch     # noqa: E402
from tests import util                             # noqa: E402


def solver(h, half, path, B, **opts):
    m = bm.MPC()
    m.h = h
    return bm.BatchSolver(mpc=m, half=half, max_batch=B, solver_options=dict(path=path, **opts))


def parity():
    rows = []
    for name in ("cfg2_standing_h10", "cfg4_walking_h10", "edge_cases_h10", "cfg3_trot_h16", "cfg5_mu_h20", "cfg_h32", "cfg_h40"):
        d = util.load(name)
        h = int(d["hor"][0]) if "hor" in d.files else 10
        half = int(d["half"][0]) if "half" in d.files else 5
        mu = d["mu_steps"] if "mu_steps" in d.files and d["mu_steps"].size else None
        ph = util.phases(d["t"], 0.04, h)
        for path in (1, 2):
            if path == 1 and h > 20:
                continue
            s = solver(h, half, path, len(ph))
            st, ct, info = s.solve(d["x_fb"], d["foot"], d["contact"], ph, d["x_cmd"], mu)
            e = util.rel_err(ct, d["controls"])
            ex = util.rel_err(st, d["states"])
            rows.append((name, path, e.max(), ex.max(), info["iters"].mean(), info["iters"].max(), info["nfactor"].mean(), int((info["status"] != 0).sum())))
            print("%-20s path %d  err u %.2e x %.2e  iters %.1f max %d nfac %.2f  unsolved %d" % rows[-1], flush=True)
    d = util.load("cfg_hgen")
    for h in d["horizons"]:
        g = lambda k: d["h%d_%s" % (h, k)]
        ph = util.phases(g("t"), 0.04, int(h))
        for path in (1, 2):
            if path == 1 and h > 20:
                continue
            s = solver(int(h), int(g("half")[0]), path, len(ph))
            st, ct, info = s.solve(g("x_fb"), g("foot"), g("contact"), ph, g("x_cmd"), g("mu_steps"))
            print("cfg_hgen h=%-2d           path %d  err u %.2e x %.2e  iters %.1f max %d nfac %.2f  unsolved %d" % (
                h, path, util.rel_err(ct, g("controls")).max(), util.rel_err(st, g("states")).max(), info["iters"].mean(),
                info["iters"].max(), info["nfactor"].mean(), int((info["status"] != 0).sum())), flush=True)


def dev_inputs(s_, dev):
    x, f, c, p = s_["x_fb"].astype(np.float32), s_["foot"].astype(np.float32), s_["contact"], s_["phase"]
    t = [torch.from_numpy(a).to(dev) for a in (x, f, c, p)]
    xc = torch.from_numpy(s_["x_cmd"].astype(np.float32)).to(dev)
    mu = None if s_["mu"] is None else torch.from_numpy(s_["mu"].astype(np.float32)).to(dev)
    return t + [xc, mu]


def timing(B, hs, opts=None):
    dev = torch.device("cuda", 0)
    for h in hs:
        gait = "standing" if h == 10 else "walking"
        s_ = synth_batch(B, h, 7, gait=gait, vx_cmd=(h != 10), per_step_mu=(h >= 20))
        for path in (1, 2):
            if path == 1 and h > 20:
                continue
            s = solver(h, s_["half"], path, B, **(opts or {}))
            args = dev_inputs(s_, dev)
            o = dict(iters=torch.zeros(B, dtype=torch.int32, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
                     nfactor=torch.zeros(B, dtype=torch.int32, device=dev))
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(2):
                    s.solve_device(*args, **o)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    s.solve_device(*args, **o)
                e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            it = o["iters"].cpu().numpy(); nf = o["nfactor"].cpu().numpy(); stt = o["status"].cpu().numpy()
            print("h %2d B %5d path %d: %.3f ms  %.3f M solves/s  iters %.1f max %d nfac %.2f unsolved %d" % (
                h, B, path, ms, B / ms / 1e3, it.mean(), it.max(), nf.mean(), int((stt != 0).sum())), flush=True)


def prof(B, h, path=2):
    dev = torch.device("cuda", 0)
    gait = "standing" if h == 10 else "walking"
    s_ = synth_batch(B, h, 7, gait=gait, vx_cmd=(h != 10), per_step_mu=(h >= 20))
    s = solver(h, s_["half"], path, B)
    args = dev_inputs(s_, dev)
    pr_t = torch.zeros((B, 16), dtype=torch.int64, device=dev)
    _lib.check(s._lib.bmpc_debug_set_profile(s._h, pr_t.data_ptr()))
    for _ in range(2):
        s.solve_device(*args)
    torch.cuda.synchronize()
    pr = pr_t.cpu().numpy().astype(float)
    print("h %d B %d path %d cycles mean: setup %.0f blocks %.0f riccati/sweeps %.0f total %.0f | iters %.1f nfac %.2f" % ((h, B, path) + tuple(pr[:, :6].mean(0))))
    it = pr[:, 3] - pr[:, 0] - pr[:, 1] - pr[:, 2]
    print("  per iteration %.0f ; blocks per factor %.0f ; riccati/sweep per factor %.0f" % ((it / pr[:, 4]).mean(), (pr[:, 1] / pr[:, 5]).mean(), (pr[:, 2] / pr[:, 5]).mean()))
    names = ["adjoint+P1", "P2", "P3", "w+forward", "P5", "tail", "backward"] if path == 2 else ["P0", "P1", "P2", "P3", "P4", "P5", "tail"]
    print("  phases, cycles per iteration: " + " ".join("%s %.0f" % (n, v) for n, v in zip(names, (pr[:, 8:15] / pr[:, 4:5]).mean(0))))


if __name__ == "__main__":
    a = sys.argv[1:]
    if not a or a[0] == "parity":
        parity()
    elif a[0] == "time":
        timing(int(a[1]), [int(v) for v in a[2].split(",")])
    elif a[0] == "prof":
        prof(int(a[1]), int(a[2]), int(a[3]) if len(a) > 3 else 2)
