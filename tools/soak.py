"""Convergence soaks (GPU box): how many instances of large random batches do not converge, and the worst iteration count.

    python tools/soak.py shapes [--set base|turn|long|stage|short] [--seeds LO HI]
        random batches of the BASELINE config shapes over a range of seeds (base: the four shapes, 65536 / 16384 per seed;
        long: h = 16 / 20 variants; stage: six long-horizon shapes on the stage-structured kernels, 8192 per seed)
    python tools/soak.py params [--cases Q_x10,R_div100,...] [--rescue off|auto|on] [--paths 1,2] [--horizons 10,20] [--batch N]
        parameter cases away from the reference's weights (REF:27-28 are user fields), per kernel family; with
        --rescue off the dense family's own losses show (the rescue pass is on by default there)
    python tools/soak.py one H GAIT SEED [--batch N]
        one shape on both kernel families and both penalty modes (scaled / absolute)
    python tools/soak.py options H GAIT SEED --opt name=value,... [--opt ...]
        one shape under several sets of solver options (bmpc_params fields), e.g. penalty ceilings of the long horizons

(Rounds 2-3 kept these as four scripts -- soak.py, soak_params.py, soak_probe.py, soak_probe2.py.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPE_SETS = {
    "base": ((10, "mixed", dict(vx_cmd=True)), (10, "standing", {}), (16, "walking", dict(vx_cmd=True)),
             (20, "walking", dict(vx_cmd=True, per_step_mu=True))),
    "long": ((16, "walking", dict(vx_cmd=True)), (20, "walking", dict(vx_cmd=True, per_step_mu=True)),
             (16, "mixed", dict(vx_cmd=True)), (20, "standing", dict(per_step_mu=True))),
    # round 6: commanded angular rates / attitude set-points / lateral commands (synth_batch turn=True)
    "turn": ((10, "mixed", dict(vx_cmd=True, turn=True)), (10, "standing", dict(turn=True)), (16, "walking", dict(vx_cmd=True, turn=True)),
             (20, "walking", dict(vx_cmd=True, per_step_mu=True, turn=True))),
    # odd and short horizons (stage-structured family; round 5)
    "short": ((1, "walking", dict(vx_cmd=True)), (2, "mixed", dict(vx_cmd=True)), (3, "walking", dict(vx_cmd=True, per_step_mu=True)),
              (5, "mixed", dict(vx_cmd=True)), (7, "walking", dict(vx_cmd=True, per_step_mu=True)), (9, "standing", {}),
              (15, "walking", dict(vx_cmd=True)), (21, "mixed", dict(vx_cmd=True, per_step_mu=True)), (33, "walking", dict(vx_cmd=True))),
    # the stage-structured family: every NP / NW variant, the long horizons twice
    "stage": ((24, "walking", dict(vx_cmd=True, per_step_mu=True)), (28, "mixed", dict(vx_cmd=True)),
              (32, "walking", dict(vx_cmd=True, per_step_mu=True)), (40, "walking", dict(vx_cmd=True, per_step_mu=True)),
              (40, "mixed", dict(vx_cmd=True)), (36, "standing", dict(per_step_mu=True))),
}

PARAM_CASES = {
    "Q_x100": lambda m, b: setattr(m, "Q", np.asarray(m.Q, float) * 100), "Q_x10": lambda m, b: setattr(m, "Q", np.asarray(m.Q, float) * 10),
    "R_div100": lambda m, b: setattr(m, "R", np.asarray(m.R, float) / 100), "R_x100": lambda m, b: setattr(m, "R", np.asarray(m.R, float) * 100),
    "dt_0.02": lambda m, b: setattr(m, "dt", 0.02), "dt_0.05": lambda m, b: setattr(m, "dt", 0.05), "m_20": lambda m, b: setattr(b, "m", 20.0),
    "reference": lambda m, b: None,
}


def solve(h, s, B, opts=None, mod=None):
    import biped_mpc_py_amd as bm
    m, b = bm.MPC(), bm.Biped()
    m.h = h
    if mod:
        mod(m, b)
    sol = bm.BatchSolver(mpc=m, biped=b, half=s["half"], max_batch=B, solver_options=opts)
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
    used = int(sol._lib.bmpc_solver_path(sol._h))
    sol.close()
    return u, info, used


def report(tag, u, info):
    st = info["status"]
    bad = np.flatnonzero(st != 0)
    print("%s: not converged %d (NaN/Inf %d) iters mean %.1f max %d nfac %.2f%s" % (
        tag, len(bad), int((st == 2).sum()), info["iters"].mean(), info["iters"].max(), info["nfactor"].mean(),
        "" if not len(bad) else " | first failures %s nfac %s" % (bad[:6].tolist(), info["nfactor"][bad][:6].tolist())), flush=True)
    return len(bad)


def main():
    from tests import util
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("shapes")
    a.add_argument("--set", default="base", choices=sorted(SHAPE_SETS))
    a.add_argument("--seeds", type=int, nargs=2, default=(500, 540))
    a = sub.add_parser("params")
    a.add_argument("--cases", default="Q_x10,R_div100,dt_0.02")
    a.add_argument("--rescue", default="off", choices=("off", "auto", "on"))
    a.add_argument("--paths", default="1,2")
    a.add_argument("--horizons", default="10,20")
    a.add_argument("--gait", default="standing")
    a.add_argument("--batch", type=int, default=16384)
    for name in ("one", "options"):
        a = sub.add_parser(name)
        a.add_argument("h", type=int)
        a.add_argument("gait")
        a.add_argument("seed", type=int)
        a.add_argument("--batch", type=int, default=16384)
        if name == "options":
            a.add_argument("--opt", action="append", default=[], help="name=value,... (one set of bmpc_params solver fields per --opt)")
            a.add_argument("--case", default="reference", choices=sorted(PARAM_CASES), help="parameter case (model / weights)")
    args = ap.parse_args()

    if args.cmd == "shapes":
        tot = bad = worst = 0
        for h, gait, kw in SHAPE_SETS[args.set]:
            B = 65536 if h == 10 else (16384 if h <= 20 else 8192)
            for seed in range(*args.seeds):
                s = util.synth_batch(B, h, seed, gait=gait, **kw)
                u, info, _ = solve(h, s, B)
                bad += report("h %d %s seed %d" % (h, gait, seed), u, info)
                tot += B
                worst = max(worst, int(info["iters"].max()))
        print("TOTAL", tot, "instances, not converged", bad, "worst iterations", worst)
    elif args.cmd == "params":
        rescue = {"off": 0, "auto": -1, "on": 1}[args.rescue]
        lost = 0
        for h in [int(v) for v in args.horizons.split(",")]:
            s = util.synth_batch(args.batch, h, 77 + h, gait=args.gait, vx_cmd=(args.gait != "standing"), per_step_mu=(h >= 20))
            for name in args.cases.split(","):
                for path in [int(v) for v in args.paths.split(",")]:
                    u, info, used = solve(h, s, args.batch, dict(path=path, rescue=rescue), PARAM_CASES[name])
                    n = report("h %d %-8s %-8s path %d (ran on %d) rescue %s" % (h, args.gait, name, path, used, args.rescue), u, info)
                    lost += n if path != 2 else 0
        print("TOTAL not converged off the stage path:", lost)
    elif args.cmd == "one":
        kw = dict(vx_cmd=(args.gait != "standing"), per_step_mu=(args.h >= 20))
        s = util.synth_batch(args.batch, args.h, args.seed, gait=args.gait, **kw)
        for path in (1, 2):
            if path == 1 and args.h > 20:
                continue
            for mode in (0, 1):
                u, info, _ = solve(args.h, s, args.batch, dict(path=path, penalty_mode=mode, rescue=0))
                report("h %d %s seed %d path %d penalties %s" % (args.h, args.gait, args.seed, path, "scaled" if mode == 0 else "absolute"), u, info)
    else:
        kw = dict(vx_cmd=(args.gait != "standing"), per_step_mu=(args.h >= 20))
        s = util.synth_batch(args.batch, args.h, args.seed, gait=args.gait, **kw)
        for spec in args.opt or [""]:
            opts = {}
            for item in filter(None, spec.split(",")):
                k, v = item.split("=")
                opts[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
            u, info, used = solve(args.h, s, args.batch, opts or None, PARAM_CASES[args.case])
            report("h %d %s seed %d %s path %d options {%s}" % (args.h, args.gait, args.seed, args.case, used, spec), u, info)


if __name__ == "__main__":
    main()
