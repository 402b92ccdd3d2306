#!/usr/bin/env python3
"""Extension fixtures for the horizons the reference cannot produce unpatched (TEST INFRASTRUCTURE; oracle only).

REF:24 makes the horizon a plain field, but REF:58 and REF:101-106 hard-code 10 rows / 5 steps, so every walking
case with h != 10 is an *extension* (SURVEY 8(c)): generated from the oracle restatement (reference-pinned at h = 10,
h-generic code) and pinned by the KKT certificate alone, exactly like cfg3_trot_h16 / cfg5_mu_h20 of gen_golden.py
(whose fixtures this script does not touch).

  cfg_h32, cfg_h40      64 instances each: walking (half = h/2, random phase), commanded v_x, per-step per-foot mu
                        -- the long-horizon cases of SURVEY 8(f) row 4 (stage-structured kernels)
  cfg_hgen              4 instances for every other even horizon in [8, 38]: the horizon as a launch parameter
  cfg_hodd              4 instances for h = 1, 2, 3, 4, 5, 7, 9, 15, 21, 33: short and odd horizons (REF:24 takes any int;
                        half = max(1, h // 2), the second touch-down point kept to the end of the horizon --
                        oracle.get_reference_foot_trajectory)

Usage:  python oracle/gen_golden_ext.py [names...]        (writes tests/golden/*.npz; minutes on 8 cores)
"""
from __future__ import annotations

import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import bmpc_oracle as orc          # noqa: E402
from oracle.gen_golden import synth_state      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
EKEYS = ("x_fb", "t", "foot", "contact", "x_cmd", "hor", "half", "mu_steps", "states", "controls", "objective",
         "n_pinned", "n_active", "polished", "kkt")


def _case(args):
    from threadpoolctl import threadpool_limits
    from oracle.gen_golden import run_extension_case
    x_fb, t, foot, contact, h, half, x_cmd, mu = args
    with threadpool_limits(limits=1):
        return run_extension_case(x_fb, t, foot, contact, h, half, x_cmd, mu)


def make_args(rng, h, n, use_mu=True):
    mp = orc.MPC()
    mp.h = h
    out = []
    for _ in range(n):
        x_fb, foot = synth_state(rng)
        # inputs cross the C ABI as fp32: the fixture holds exactly what the kernels see
        x_fb = x_fb.astype(np.float32).astype(float)
        foot = foot.astype(np.float32).astype(float)
        k = int(rng.integers(0, h))
        t = k * mp.dt + 0.5 * mp.dt
        x_cmd = np.array(mp.x_cmd, float)
        x_cmd[9] = float(np.float32(rng.uniform(-0.5, 0.5)))
        half = max(1, h // 2)
        contact = orc.get_contact_sequence(t, mp, half=half)
        mu = rng.uniform(0.3, 0.9, (h, 2)).astype(np.float32).astype(float) if use_mu else None
        out.append((x_fb, t, foot, contact, h, half, x_cmd, mu))
    return out


def main(names):
    os.makedirs(OUT, exist_ok=True)
    jobs = {"cfg_h32": [(32, 64, 32)], "cfg_h40": [(40, 64, 40)],
            "cfg_hgen": [(h, 4, 500 + h) for h in range(8, 40, 2) if h not in (10, 16, 20, 32)],
            "cfg_hodd": [(h, 4, 600 + h) for h in (1, 2, 3, 4, 5, 7, 9, 15, 21, 33)]}
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        for name in names or list(jobs):
            t0 = time.time()
            args = []
            for h, n, seed in jobs[name]:
                args += make_args(np.random.default_rng(seed), h, n)
            cases = pool.map(_case, args, chunksize=1)
            if name in ("cfg_hgen", "cfg_hodd"):   # ragged horizons: one record per horizon inside one file
                rec = {}
                for h, n, _ in jobs[name]:
                    cs = [c for c in cases if int(c["hor"]) == h]
                    for k in EKEYS:
                        rec[f"h{h}_{k}"] = np.stack([c[k] for c in cs])
                rec["horizons"] = np.array([h for h, _, _ in jobs[name]], np.int32)
                np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
            else:
                np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.stack([c[k] for c in cases]) for k in EKEYS})
            print(name, "max kkt", np.max([c["kkt"] for c in cases], axis=0), "polished", int(np.sum([c["polished"] for c in cases])),
                  "of", len(cases), "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
