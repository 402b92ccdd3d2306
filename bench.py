#!/usr/bin/env python3
"""bench.py -- throughput of the MPC hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `bmpc_solve_batch_device` on synthetic randomised
CoM / stance states (SURVEY 8(d) generator), inputs already resident in HBM.  Default workload:
BASELINE.json configs[1] (B = 4096 per GPU, horizon 10, double support).

    python bench.py [--gpus N] [--steps K] [--warmup W]            the driver's contract
    python bench.py --config {2,3,4,5}                              another BASELINE config (own roofline line)
    python bench.py --config {6,7}                                  the long-horizon extensions h = 32 / 40 (stage-structured kernels)
    python bench.py --gpus 8 --config 4 --scaling strong            ONE 65536 batch sharded N ways

--gpus N > 1 works with or without a launcher: under torch.distributed.run (RANK / WORLD_SIZE in the
environment) this process is one rank; started bare, it spawns the N ranks itself -- before torch or HIP is
touched, as child processes, never by exec -- and relays rank 0's JSON line.  One process per GPU,
backend "nccl" (= RCCL over xGMI).

What a multi-GPU step contains (instances are independent, SURVEY 8(e): no exchange step inside the solve):
  weak   every rank solves its own `--batch` instances, then the results are collected with ONE RCCL
         all_gather of the controls (the north_star's "gather"); value = N * batch * steps / time.  The gather
         of step k is asynchronous and overlaps the solve of step k + 1 (two output buffers), as a production
         loop would run it; every gather is complete before the timed region ends.
  strong ONE seeded batch of `--total` instances; per step: RCCL broadcast of the ~0.7 KB parameter block,
         every rank solves its contiguous shard, all_gather of the controls; afterwards rank 0 solves the
         whole batch alone and the gathered controls must equal that bit for bit.
The timed region is bracketed by barrier + synchronize on both sides; the slowest rank's time counts.

Extra objects on the JSON line:
  roofline      the (only) kernel of the path: algorithmic flops per launch / average launch duration,
                measured with HIP events around every launch on the launch stream;
                two counts: SURVEY 8(d)'s dense condensed-ADMM formula (`achieved`) and the flops of the
                algorithm actually run, symmetric work counted once (`achieved_minimal`).
  cpu_baseline  the reference-style CPU path (dense np.kron assembly as REF:203-286 + a plain fp64 interior
                point solve, oracle/bmpc_oracle.py) on a bounded sample of the same workload: single-core
                latency split into assembly / solve, and the all-core rate (N = 1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MAX_CPU_WORKERS = 16
PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 vector = fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_FP64_TFLOPS = 78.6           # MI355X fp64 vector peak (AMD spec sheet: half the fp32 vector rate; the guide lists no fp64 figure)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # (0.08 s of timed region at the default configuration: box-to-box and
    ap.add_argument("--warmup", type=int, default=5)       #  run-to-run differences of a 17 ms region were +-2 %)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5, 6, 7),
                    help="BASELINE.json config number (1-based); 6 / 7: the long-horizon extensions h = 32 / 40")
    ap.add_argument("--path", choices=("auto", "dense", "stage", "best"), default=None,
                    help="kernel family: dense inverse, stage-structured, the library's choice, or `best` = time both "
                         "before the timed region and take the faster (default for config 5: its h = 20 has both)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--batch", type=int, default=None, help="weak: instances per GPU per step (default 4096)")
    ap.add_argument("--total", type=int, default=None, help="strong: instances in the one global batch (default 65536)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the results on their ranks")
    ap.add_argument("--no-strong-record", action="store_true",
                    help="N > 1, weak mode: skip the extra strong-scaling measurement of config 4 (the `strong` sub-record)")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="instances for the all-core CPU baseline and the in-run parity (0 = skip; default: the whole batch "
                         "at h = 10, i.e. ~4 s on 16 cores, 256 otherwise)")
    ap.add_argument("--skip-host-path", action="store_true",
                    help="do not measure the host-pointer (PCIe-inclusive) rate: under rocprofv3 its chunked launches would be "
                         "averaged into the kernel's statistics (tools/profile_round.sh)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal only: initialise torch.distributed and run the collectives even with one rank")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (1-GPU box, use with --backend gloo)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """Started without a launcher: run the N ranks as children (this parent never initialises the GPU,
    nothing is exec'ed) and relay rank 0's JSON line.  A failing rank takes the others down with it."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    out0 = b""
    deadline = time.monotonic() + float(os.environ.get("BMPC_BENCH_LAUNCH_TIMEOUT", "1500"))
    while pending:
        if time.monotonic() > deadline:               # a rank stuck in a rendezvous must not outlive the launcher
            for q in pending:
                procs[q].kill()
            rc = rc or 124
        for r in sorted(pending):
            if r == 0:
                try:
                    o, _ = procs[0].communicate(timeout=0.5)
                    out0 += o or b""
                except subprocess.TimeoutExpired:
                    continue
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                rc = rc or code
                for q in pending:                         # a dead rank leaves the others in a collective
                    procs[q].kill()
        time.sleep(0.2)
    for line in out0.decode("utf-8", "replace").splitlines():
        if line.startswith("{"):
            print(line, flush=True)
    return rc


# ------------------------------------------------------------------------------------------ flop counts
def flops_survey(h, iters):
    """dense condensed-ADMM algorithm, F(h, k) = F_setup(h) + k F_iter(h), n = 12h, nx = 13h."""
    n, nx = 12 * h, 13 * h
    f_setup = 4056 * h * (h - 1) / 2 + n * (n + 1) * nx + (2 * n * nx + 338 * h) + n ** 3 / 3 + 2 * nx * n
    f_iter = 2 * n * n + 640 * h
    return f_setup + iters * f_iter


def flops_run(h, iters, nfactor):
    """Flops of the algorithm that runs (1 MAC = 2 flops), DESIGN.md section 6: set-up of the wrench-space
    Hessian rows and gradient; per factorisation the 6x6 block algebra and the explicit symmetric inverse
    of the 6h x 6h matrix (n^3 flops: symmetric work counted once); per iteration the sparse constraint
    products, the G~ mat-vec and the application of K^-1."""
    n = 6 * h
    f_setup = 2.0 * (27 * h * (h - 1) / 2 + 9 * h * h * (h + 1) / 2 + 40 * h * h)
    f_factor = 1.0 * n ** 3 + 2.0 * 3500 * h
    f_iter = 2.0 * h * (690 + 48 * h)
    return f_setup + nfactor * f_factor + iters * f_iter, dict(setup=f_setup, factor=f_factor, iteration=f_iter)


def mfma_count(h):
    """v_mfma_f64_16x16x4_f64 instructions per solve of the dense family (bmpc_kernels.hip, phase B: the torque block of the
    wrench-space Hessian Gt = M' M, four rows of M per instruction): ceil(3 (h - 1) / 4) for the Euler rows + 1 for the angular-
    velocity rows (the same three rows at every state step: a rank-3 product and a count) per 16 x 16 tile, for the
    NTL (NTL + 1) / 2 tiles of the upper triangle, NTL = ceil(3 h / 16).  Each is 16 x 16 x 4 MACs = 2048 flops (f64)."""
    ntl = -(-3 * h // 16)
    return (ntl * (ntl + 1) // 2) * (-(-3 * (h - 1) // 4) + 1)


def stage_variant(h):
    """(steps a lane owns, waves per instance) of the stage-structured kernel that serves horizon h
    (bmpc_stage.hip: stage_steps_per_lane / stage_waves): one wave up to h = 24, two from h = 26."""
    nw = 1 if h <= 24 else 2
    return max(2, -(-h // (5 * nw))), nw


def flops_run_stage(h, iters, nfactor):
    """Flops of the stage-structured path (bmpc_stage.hip; 1 MAC = 2 flops): O(h) everywhere.  Set-up: references, step
    data, free response (~400 per step).  Per factorisation: the 6x6 block algebra (as the dense path, 7000 per step)
    and the backward Riccati recursion -- per step one 6x6 LDL' with 24 right-hand sides and the Schur complement /
    congruence products on 6x6 blocks (~2 x 2100 MACs).  Per iteration: constraint products and residual (690 per step, as
    the dense path), the adjoint of the tracking error (~60), two 12x12 mat-vecs of the stage solve and S^-1 g (2 x 324)."""
    f_setup = 2.0 * 400 * h
    f_factor = 2.0 * (3500 + 2100) * h
    f_iter = 2.0 * h * (690 + 60 + 324)
    return f_setup + nfactor * f_factor + iters * f_iter, dict(setup=f_setup, factor=f_factor, iteration=f_iter)


def hbm_bytes_per_solve(h, x_cmd, mu):
    """SURVEY 8(d) compulsory I/O: x_fb, foot, phase, contact (u8) [+ x_cmd] [+ mu] in; controls, states,
    iters / status / nfactor / residuals out."""
    return 4 * (12 + 6 + 1) + 2 * h + (48 if x_cmd else 0) + (8 * h if mu else 0) + 4 * 25 * h + 20


# ------------------------------------------------------------------------------------------ CPU baseline
def _cpu_init():
    """Worker start-up: ONE BLAS / OpenMP thread per worker process.  The environment variables must be set
    before numpy loads its BLAS, and threadpoolctl can only limit libraries that are already loaded -- so:
    environment, imports, then the limit (16 workers x a host-wide default thread count thrash the box)."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[k] = "1"
    import numpy  # noqa: F401
    import scipy.linalg  # noqa: F401
    import scipy.optimize  # noqa: F401
    from oracle import bmpc_oracle as orc          # noqa: F401  (import cost paid before the timed map)
    from threadpoolctl import threadpool_limits
    global _TP_LIMIT
    _TP_LIMIT = threadpool_limits(limits=1)


def _cpu_one(args):
    """One instance the way the reference does it: dense assembly of P, q, G, h, A, b (REF:203-286), then a
    plain fp64 interior point solve (what REF:297 asks cvxopt for).  `full` adds the oracle's polish."""
    x, t, f, c, xc, mu, h, half, full = args
    import numpy as np
    from oracle import bmpc_oracle as orc
    mpc, bp = orc.MPC(), orc.Biped()
    mpc.h = h
    if xc is not None:
        mpc.x_cmd = np.asarray(xc, float)
    t0 = time.perf_counter()
    sp = orc.build_sparse_qp(x, t, f, mpc, bp, c, half=half, mu_steps=mu)
    t1 = time.perf_counter()
    z, _, _, info = orc.solve_qp(sp["P"], sp["q"], sp["G"], sp["h"], sp["A"], sp["b"], 13 * h, polish=full)
    t2 = time.perf_counter()
    k = info["kkt"]
    certified = bool(info["polished"]) and max(k["stationarity"], k["primal_ineq"], k["complementarity"]) <= 1e-7
    return z[13 * h:].reshape(h, 12), t1 - t0, t2 - t1, certified


def _log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(s, h, dt, n_sample):
    """Reference-style CPU path on the first n_sample instances: single-core latency (ONE worker process
    busy, one BLAS thread) and all-core rate (one process per core); pool start-up and imports excluded.
    Everything runs in spawned workers: this process holds torch and the HIP runtime, and neither the BLAS
    thread limits nor a fork belong in it.  Every wait is bounded."""
    import multiprocessing as mp
    import numpy as np
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, MAX_CPU_WORKERS))      # the GPU box grants a 16-CPU share per GPU

    def arg(i, full):
        # a time in the middle of schedule step `phase`, so that int(t // dt) % h (REF:99-100) gives it back
        # the inputs cross the C ABI as fp32: the CPU path solves exactly the instance the GPU saw
        r32 = lambda a: np.asarray(a, np.float32).astype(float)
        return (r32(s["x_fb"][i]), (float(s["phase"][i]) + 0.5) * dt, r32(s["foot"][i]), s["contact"][i],
                r32(s["x_cmd"][i]) if s.get("use_x_cmd") else None, None if s["mu"] is None else r32(s["mu"][i]),
                h, s["half"], full)

    n1 = min(8, n_sample)
    t_end = time.monotonic() + 150.0                 # the whole baseline is bounded: a reported number, not the product

    def get(res):
        return res.get(max(5.0, t_end - time.monotonic()))
    with mp.get_context("spawn").Pool(cores, initializer=_cpu_init) as pool:
        _log(f"cpu_baseline: {cores} workers starting")
        get(pool.map_async(_cpu_one, [arg(i % n_sample, False) for i in range(2 * cores)], chunksize=1))   # start-up, imports
        _log("cpu_baseline: workers warm; single-core latency")
        single = get(pool.map_async(_cpu_one, [arg(i, False) for i in range(n1)], chunksize=n1))     # one worker, the others idle
        asm_ms = 1e3 * float(np.mean([o[1] for o in single]))
        sol_ms = 1e3 * float(np.mean([o[2] for o in single]))
        _log(f"cpu_baseline: {asm_ms:.1f} + {sol_ms:.1f} ms per solve on one core; all-core rate on {n_sample} instances")
        t0 = time.perf_counter()
        out = get(pool.map_async(_cpu_one, [arg(i, False) for i in range(n_sample)]))
        wall = time.perf_counter() - t0
        nfull = n_sample                              # the parity check covers every instance of the sample
        t0 = time.perf_counter()
        _log(f"cpu_baseline: {n_sample / wall:.0f} solves/s on {cores} cores; oracle with polish on {nfull} instances")
        outf = get(pool.map_async(_cpu_one, [arg(i, True) for i in range(nfull)]))
        wallf = time.perf_counter() - t0
    _log("cpu_baseline: done")
    ctrl_full = np.stack([o[0] for o in outf])
    cert_full = np.array([o[3] for o in outf], bool)
    ctrl_plain = np.stack([o[0] for o in out])
    cb = dict(value=n_sample / wall, unit="solves/s", cores=cores, kind="port",
              single_core_ms={"assembly": asm_ms, "solve": sol_ms, "total": asm_ms + sol_ms},
              sample=f"{n_sample} instances of the same batch; per instance the reference-style dense assembly "
                     f"(oracle.build_sparse_qp, REF:203-286) + a plain fp64 Mehrotra interior-point solve without "
                     f"polish (what REF:297 asks cvxopt for; cvxopt itself is absent from the image); one process "
                     f"per core, one BLAS thread each, pool start-up and imports excluded; single-core latency "
                     f"from {n1} instances on one worker with the others idle",
              oracle_with_polish={"value": nfull / wallf, "unit": "solves/s", "instances": nfull,
                                  "note": "oracle.solve_mpc as the parity tests use it (IPM + active-set polish + certificate)"})
    return ctrl_full, ctrl_plain, cb, cert_full


class _StdoutToStderr:
    """File descriptor 1 points at stderr inside the block (C-level output of a library included): stdout is
    reserved for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        with _StdoutToStderr():                     # RCCL prints its version banner on stdout when the communicator is made
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
            dist.barrier()
        # (rehearsal hook of tests/test_gpu_parity.py: this rank dies the hard way once the group is up -- the launcher has to
        #  take the others, who then sit in a collective, down with it)
        if os.environ.get("BMPC_BENCH_KILL_RANK") == str(rank):
            import signal
            os.kill(os.getpid(), signal.SIGKILL)

    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import sharding
    from biped_mpc_py_amd.synth import CONFIGS, synth_batch

    # what RCCL itself saw: the world size of the initialised process group, its backend, one device UUID per rank
    group = None
    if use_dist:
        uu = str(getattr(torch.cuda.get_device_properties(dev), "uuid", "unknown"))
        uuids = [None] * world
        dist.all_gather_object(uuids, f"rank {rank}: cuda:{dev_index} {torch.cuda.get_device_name(dev)} uuid {uu}")
        group = {"world_size_backend": int(dist.get_world_size()), "backend": str(dist.get_backend()),
                 "devices": uuids, "distinct_devices": len({u.split("uuid")[-1] for u in uuids})}

    def timed_run(config, strong, path_arg, steps, warmup, batch=None, total_arg=None, gather_on=True):
        """One measurement under the driver's contract -- `warmup` untimed steps, `steps` timed ones bracketed by barrier +
        synchronize on both sides, the slowest rank's time -- of one BASELINE config, weak (every rank its own batch) or
        strong (ONE batch sharded over the ranks: broadcast(params) + solve + all_gather per step, gathered controls checked
        bit for bit against rank 0's solve of the whole batch)."""
        cfg = CONFIGS[config]
        h = cfg["h"]
        mpc = bm.MPC()
        mpc.h = h
        if strong:
            total = total_arg or 65536
            s = synth_batch(total, h, cfg["seed"], gait=cfg["gait"], **cfg["kw"])       # ONE global batch
            lo, hi = sharding.shard_bounds(total, rank, world)
        else:
            B0 = batch or 4096
            total = world * B0
            s = synth_batch(B0, h, cfg["seed"] + 1000 * rank, gait=cfg["gait"], **cfg["kw"])
            lo, hi = 0, B0
        B = hi - lo
        use_x_cmd = bool(cfg["kw"].get("vx_cmd"))
        s["use_x_cmd"] = use_x_cmd
        cp = bm.pack_params(mpc, bm.Biped(), half=s["half"],
                            solver_options=dict(path={"auto": 0, "dense": 1, "stage": 2, "best": 0}[path_arg]))
        if use_dist:                                    # C0: one parameter block for every rank
            sharding.broadcast_params(cp, src=0, device=coll_dev)
        solver = bm.BatchSolver(cparams=cp, device=dev_index, max_batch=max(B, total if (strong and rank == 0) else B))

        def dev_inputs(a, b):
            return dict(x_fb=torch.from_numpy(s["x_fb"][a:b].astype(np.float32)).to(dev),
                        foot=torch.from_numpy(s["foot"][a:b].astype(np.float32)).to(dev),
                        contact=torch.from_numpy(np.ascontiguousarray(s["contact"][a:b])).to(dev),
                        phase=torch.from_numpy(np.ascontiguousarray(s["phase"][a:b])).to(dev),
                        x_cmd=torch.from_numpy(s["x_cmd"][a:b].astype(np.float32)).to(dev) if use_x_cmd else None,
                        mu=None if s["mu"] is None else torch.from_numpy(s["mu"][a:b].astype(np.float32)).to(dev))

        if rank == 0:
            _log(f"rank 0 of {world}: config {config} ({'strong' if strong else 'weak'}), h = {h}, {B} instances on this GPU")
        tin = dev_inputs(lo, hi)
        o_u2 = [torch.empty((B, h, 12), dtype=torch.float32, device=dev) for _ in range(2)]   # double buffered: step k's
        o_s = torch.empty((B, h, 13), dtype=torch.float32, device=dev)                          # gather overlaps step k + 1's solve
        o_it = torch.empty(B, dtype=torch.int32, device=dev)
        o_st = torch.empty(B, dtype=torch.int32, device=dev)
        o_nf = torch.empty(B, dtype=torch.int32, device=dev)
        o_rs = torch.empty((B, 2), dtype=torch.float32, device=dev)
        gather = use_dist and gather_on
        per = -(-total // world)                        # padded shard length of the all_gather
        st = dict(o_u=o_u2[0], g_out=None)
        if gather:
            g_in2 = [o_u2[i] if B == per else torch.zeros((per, h, 12), dtype=torch.float32, device=dev) for i in range(2)]
            g_in_c2 = [g_in2[i] if args.backend == "nccl" else torch.zeros((per, h, 12), dtype=torch.float32) for i in range(2)]
            g_out2 = [torch.empty((world * per, h, 12), dtype=torch.float32, device=coll_dev) for _ in range(2)]
            st["g_out"] = g_out2[0]
            pending = [None, None]

        # `best`: both kernel families (where both exist) on this batch before the timed region; rank 0's choice is
        # everybody's.  The families share the outer method, so the answer is the same either way.
        path_trial = None
        if path_arg == "best" and solver._lib.bmpc_supported_horizon_path(h, 1) and solver._lib.bmpc_supported_horizon_path(h, 2):
            path_trial = {}
            for name, code in (("dense", 1), ("stage", 2)):
                cpt = bm.pack_params(mpc, bm.Biped(), half=s["half"], solver_options=dict(path=code))
                solver.set_params(cpt)
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    solver.solve_device(tin["x_fb"], tin["foot"], tin["contact"], tin["phase"], x_cmd=tin["x_cmd"], mu=tin["mu"],
                                        controls=o_u2[0], iters=o_it, status=o_st, nfactor=o_nf)
                    e1.record()
                    torch.cuda.synchronize(dev)
                    path_trial[name] = e0.elapsed_time(e1) if rep else 1e30          # (first launch: code load)
            pick = torch.tensor([1 if path_trial["dense"] <= path_trial["stage"] else 2], dtype=torch.int32, device=coll_dev)
            if use_dist:
                dist.broadcast(pick, src=0)
            cp = bm.pack_params(mpc, bm.Biped(), half=s["half"], solver_options=dict(path=int(pick.item())))
            solver.set_params(cp)
        path_used = {1: "dense", 2: "stage"}[int(solver._lib.bmpc_solver_path(solver._h))]
        # the block the in-step broadcast carries: the FINAL one (after the path pick)
        pbuf = torch.from_numpy(np.frombuffer(bytes(cp), dtype=np.uint8).copy()).to(coll_dev) if (gather and strong) else None

        kev = []
        gev = []                                        # host time spent waiting for a step's gather (N > 1)
        nstep = [0]

        def drain():
            """Wait for the collectives still in flight (their buffers are about to be reused / read)."""
            if gather:
                for i in range(2):
                    if pending[i] is not None:
                        pending[i].wait()
                        pending[i] = None

        def step(timed):
            # Pipelined like a production loop: the all_gather of step k runs on RCCL's stream while step k + 1
            # solves into the other output buffer; a buffer is reused only after its gather has finished.
            b = nstep[0] & 1
            nstep[0] += 1
            if gather and pending[b] is not None:
                tw = time.perf_counter()
                pending[b].wait()                       # (NCCL: the current stream waits, not the host)
                pending[b] = None
                if timed:
                    gev.append(1e3 * (time.perf_counter() - tw))
            if gather and strong:                       # C0 inside the step: the block every rank solves with
                dist.broadcast(pbuf, src=0)
            st["o_u"] = o_u2[b]
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            solver.solve_device(tin["x_fb"], tin["foot"], tin["contact"], tin["phase"], x_cmd=tin["x_cmd"], mu=tin["mu"],
                                controls=st["o_u"], states=o_s, iters=o_it, residuals=o_rs, status=o_st, nfactor=o_nf)
            if timed:
                e1.record()
                kev.append((e0, e1))
            if gather:                                  # C2: every rank receives all controls
                st["g_out"] = g_out2[b]
                if g_in2[b] is not st["o_u"]:
                    g_in2[b][:B].copy_(st["o_u"])
                if args.backend == "nccl":
                    pending[b] = dist.all_gather_into_tensor(st["g_out"], g_in2[b], async_op=True)
                else:                                   # rehearsal backend: collectives on host tensors
                    g_in_c2[b].copy_(g_in2[b])
                    pending[b] = dist.all_gather_into_tensor(st["g_out"], g_in_c2[b], async_op=True)

        def fence():
            drain()
            torch.cuda.synchronize(dev)
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize(dev)

        # the kernel is launched on torch's CURRENT stream; a real (non-null) stream, so that the events are
        # recorded on exactly the stream the kernel runs on
        launch_stream = torch.cuda.Stream(dev)
        # A region of K steps at the headline configuration is 17 ms at the driver's K = 20, and one box differs from the next by
        # several per cent: below 100 steps the bracketed region of EXACTLY K steps is therefore run
        # `repeats` = 5 times back to back and the line is the MEDIAN repeat (its wall clock, its kernel events: `ms_per_step`
        # x `steps` is that region's own time); the spread goes on the line as value_min / value_max.
        repeats = 5 if steps < 100 else 1
        rep_elapsed, rep_kev, rep_gev = [], [], []
        with torch.cuda.stream(launch_stream):
            for _ in range(warmup):
                step(False)
            for _ in range(repeats):
                del kev[:], gev[:]
                fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step(True)
                fence()
                rep_elapsed.append(time.perf_counter() - t0)
                rep_kev.append(list(kev))
                rep_gev.append(list(gev))
        rep_t = torch.tensor(rep_elapsed, dtype=torch.float64, device=coll_dev)
        if use_dist:
            dist.all_reduce(rep_t, op=dist.ReduceOp.MAX)              # a repeat's time is its slowest rank's
        rep_all = [float(v) for v in rep_t.cpu()]
        mid = int(np.argsort(rep_all)[len(rep_all) // 2])             # (the same repeat on every rank)
        elapsed = rep_all[mid]
        gev = rep_gev[mid]
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in rep_kev[mid]]))     # average launch duration, the reported region
        if rank == 0:
            _log(f"timed region done: {1e3 * elapsed / steps:.3f} ms per step (median of {repeats}: "
                 f"{', '.join('%.3f' % (1e3 * t / steps) for t in rep_all)}), kernel {kernel_ms:.3f} ms")
        kernel_ms_ranks = [kernel_ms]
        if use_dist:
            own = torch.tensor([kernel_ms], dtype=torch.float64, device=coll_dev)
            allk = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
            dist.all_gather(allk, own)
            kernel_ms_ranks = [float(v.item()) for v in allk]
            t = torch.tensor([kernel_ms], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            kernel_ms = float(t[0].item())
        res = dict(cfg=cfg, config=config, h=h, mpc=mpc, s=s, lo=lo, hi=hi, B=B, total=total, strong=strong, use_x_cmd=use_x_cmd,
                   solver=solver, path_used=path_used, path_trial=path_trial, path_arg=path_arg, gather=gather, steps=steps,
                   warmup=warmup, elapsed=elapsed, rep_elapsed=rep_all, kernel_ms=kernel_ms, kernel_ms_ranks=kernel_ms_ranks, gev=gev,
                   iters=o_it.cpu().numpy(), nfac=o_nf.cpu().numpy(), status=o_st.cpu().numpy(), o_u=st["o_u"], gather_check=None)
        # strong scaling: the gathered controls against the single-GPU solve of the whole batch, bit for bit
        if gather and strong:
            fence()
            if rank == 0:
                tall = dev_inputs(0, total)
                u1 = torch.empty((total, h, 12), dtype=torch.float32, device=dev)
                solver.solve_device(tall["x_fb"], tall["foot"], tall["contact"], tall["phase"], x_cmd=tall["x_cmd"],
                                    mu=tall["mu"], controls=u1)
                torch.cuda.synchronize(dev)
                got = torch.cat([st["g_out"][r * per:r * per + (sharding.shard_bounds(total, r, world)[1] -
                                                                sharding.shard_bounds(total, r, world)[0])] for r in range(world)])
                same = bool(torch.equal(got.to(dev), u1))
                res["gather_check"] = ("gathered controls bit-identical to the N=1 solve of the whole batch"
                                       if same else "MISMATCH against the N=1 solve")
                if not same:
                    raise SystemExit("gathered controls differ from the single-GPU result")
        return res

    def two_in_flight():
        """The timed steps again with TWO batches in flight (two handles, two streams, alternating): a secondary record."""
        solver_b = bm.BatchSolver(cparams=solver.cparams, device=dev_index, max_batch=B)
        tin2 = {k: (None if s.get(k) is None or (k == "x_cmd" and not use_x_cmd) else
                    torch.from_numpy(np.ascontiguousarray(s[k][lo:hi].astype(np.float32) if s[k].dtype == np.float64 else s[k][lo:hi])).to(dev))
                for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
        pair = [(solver, torch.cuda.Stream(dev), torch.empty((B, h, 12), dtype=torch.float32, device=dev)),
                (solver_b, torch.cuda.Stream(dev), torch.empty((B, h, 12), dtype=torch.float32, device=dev))]

        def both(n):
            for k in range(n):
                sv, stq, out = pair[k & 1]
                with torch.cuda.stream(stq):
                    sv.solve_device(tin2["x_fb"], tin2["foot"], tin2["contact"], tin2["phase"], x_cmd=tin2["x_cmd"], mu=tin2["mu"],
                                    controls=out)
        both(max(2, args.warmup))
        torch.cuda.synchronize(dev)
        n2 = max(2, args.steps) & ~1
        t0 = time.perf_counter()
        both(n2)
        torch.cuda.synchronize(dev)
        t_two = (time.perf_counter() - t0) / n2
        rec = {
            "value": B / t_two, "unit": "solves/s", "ms_per_step": 1e3 * t_two, "steps": n2, "streams": 2,
            "ratio_to_value": (B / t_two) / (B * args.steps / elapsed),
            "bit_identical_to_one_at_a_time": bool(torch.equal(pair[0][2], o_u) and torch.equal(pair[1][2], o_u)),
            "what": "the same step with two batches in flight on two streams (two handles): the next launch's head fills the instance "
                    "slots the tail of this one leaves idle (4096 instances on 1024 slots: DESIGN 9); not `value`"}
        solver_b.close()
        return rec

    def ordered_dispatch():
        """The timed steps again with the instances dispatched longest first, by the iteration counts of the solve before (what a
        receding-horizon caller has from the previous control period: `bmpc_set_dispatch_order`): a secondary record that sizes the
        tail of a launch -- the results do not depend on the order."""
        tin2 = {k: (None if s.get(k) is None or (k == "x_cmd" and not use_x_cmd) else
                    torch.from_numpy(np.ascontiguousarray(s[k][lo:hi].astype(np.float32) if s[k].dtype == np.float64 else s[k][lo:hi])).to(dev))
                for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}
        out = torch.empty((B, h, 12), dtype=torch.float32, device=dev)
        it = torch.empty(B, dtype=torch.int32, device=dev)
        kw = dict(x_cmd=tin2["x_cmd"], mu=tin2["mu"], controls=out, iters=it)
        solver.solve_device(tin2["x_fb"], tin2["foot"], tin2["contact"], tin2["phase"], **kw)
        torch.cuda.synchronize(dev)
        order = torch.argsort(it, descending=True, stable=True).to(torch.int32).contiguous()
        solver.set_dispatch_order(order)
        try:
            for _ in range(max(1, args.warmup)):
                solver.solve_device(tin2["x_fb"], tin2["foot"], tin2["contact"], tin2["phase"], **kw)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                solver.solve_device(tin2["x_fb"], tin2["foot"], tin2["contact"], tin2["phase"], **kw)
            torch.cuda.synchronize(dev)
            t_ord = (time.perf_counter() - t0) / args.steps
        finally:
            solver.set_dispatch_order(None)
        return {"value": B / t_ord, "unit": "solves/s", "ms_per_step": 1e3 * t_ord, "steps": args.steps,
                "ratio_to_value": (B / t_ord) / (B * args.steps / elapsed),
                "bit_identical_to_batch_order": bool(torch.equal(out, o_u)),
                "what": "the same step with the workgroups dispatched by descending iteration count of the previous solve of the same batch "
                        "(bmpc_set_dispatch_order; roll-outs do it by themselves): what the unequal instances of a batch cost a launch "
                        "that does not know them (4096 instances on 1024 slots); not `value`"}

    strong = args.scaling == "strong"
    path_arg = args.path or ("best" if args.config == 5 else "auto")
    R = timed_run(args.config, strong, path_arg, args.steps, args.warmup, batch=args.batch, total_arg=args.total,
                  gather_on=not args.no_gather)
    cfg, h, mpc, s, lo, hi, B, total = R["cfg"], R["h"], R["mpc"], R["s"], R["lo"], R["hi"], R["B"], R["total"]
    use_x_cmd, solver, path_used, path_trial, gather = R["use_x_cmd"], R["solver"], R["path_used"], R["path_trial"], R["gather"]
    elapsed, kernel_ms, kernel_ms_ranks, gev, o_u = R["elapsed"], R["kernel_ms"], R["kernel_ms_ranks"], R["gev"], R["o_u"]
    rep_elapsed = R["rep_elapsed"]
    # N > 1, default (weak) mode: the north_star partition measured in the same run -- ONE 65536 batch of config 4 (mixed gait
    # schedules) sharded over the ranks, broadcast + solve + all_gather per step, gather checked bit for bit
    R2 = None
    if use_dist and not strong and not args.no_gather and not args.no_strong_record:      # (N > 1; one rank under --force-dist)
        R2 = timed_run(4, True, "auto", args.steps, args.warmup, total_arg=args.total)
        R2["solver"].close()

    iters, nfac, status = R["iters"], R["nfac"], R["status"]
    line = None
    if rank == 0:
        from biped_mpc_py_amd.synth import kernel_source_hash
        fl_s = flops_survey(h, float(iters.mean()))
        fl_r, parts = (flops_run if path_used == "dense" else flops_run_stage)(h, float(iters.mean()), float(nfac.mean()))
        ach_s = fl_s * B / (kernel_ms * 1e-3) / 1e12
        ach = fl_r * B / (kernel_ms * 1e-3) / 1e12
        # HBM bytes per launch and the MFMA share need rocprofv3 --pmc passes (tools/profile_round.sh -> profiles/); what
        # is replayed here was measured with THESE kernel sources and this path, or it is refused
        traffic, traffic_source, mfma_ops, hw_flops = None, None, None, None
        ksha = kernel_source_hash()
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as fh:
                for pm in json.load(fh):
                    if pm.get("batch") == B and pm.get("config") == args.config and pm.get("path", "dense") == path_used:
                        if pm.get("kernel_sha") != ksha:
                            traffic_source = (f"profiles/pmc_summary.json holds a figure for other kernel sources "
                                              f"({pm.get('kernel_sha')} != {ksha}): refused; rerun tools/profile_round.sh")
                            continue
                        traffic = pm["traffic_bytes_per_launch"]
                        sqc = pm.get("sq_per_launch", {})
                        mfma_ops = {k: sqc[k] for k in ("SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA") if k in sqc} or None
                        hw_flops = pm.get("valu_flops_per_launch")
                        traffic_source = "replayed from " + pm.get("source", "profiles/pmc_summary.json") + \
                                         " (rocprofv3 --pmc passes of this command; not measured in this run)"
        except (OSError, ValueError, KeyError, TypeError):
            pass
        what = (f"one batch of {total} sharded over {world} GPU(s)" if strong else f"batch={B} per GPU")
        line = {
            "metric": "MPC QP solves/sec (N=10, 2-contact)" if h == 10 else f"MPC QP solves/sec (N={h})",
            "value": total * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # spread of the repeated regions (one region = `steps` steps between barrier + synchronize brackets; `value` is the median one)
            "value_min": total * args.steps / max(rep_elapsed), "value_max": total * args.steps / min(rep_elapsed),
            "config": {"workload": f"{cfg['label']}; {what}, inputs resident in HBM",
                       "timed_regions": {"repeats": len(rep_elapsed), "reported": "median" if len(rep_elapsed) > 1 else "the one region",
                                         "ms_per_step_each": [1e3 * t / args.steps for t in rep_elapsed],
                                         "why": "below 100 steps the bracketed region of exactly `steps` steps is run 5 times back to back "
                                                "(a 20-step region is 17 ms; boxes differ by several per cent) and `value`, `ms_per_step`, "
                                                "`roofline.kernel_ms` are those of the median region"},
                       "baseline_config": args.config, "batch_per_gpu": B, "total": total, "horizon": h,
                       "path": path_used, "path_choice": (
                           {"how": "both kernel families timed on this batch before the timed region (ms per launch)", **path_trial}
                           if path_trial else {"how": f"--path {path_arg}"}),
                       "scaling": args.scaling,
                       "residual_dtype": "f64", "collectives_in_step": (["broadcast(params)"] if gather and strong else []) +
                       (["all_gather(controls)"] if gather else []),
                       "mean_iters": float(iters.mean()), "max_iters": int(iters.max()),
                       "mean_factorisations": float(nfac.mean()),
                       "not_converged": int((status != 0).sum())},
            # `achieved` / `frac`: SURVEY 8(d)'s per-solve flop figure x the solves of a launch / the launch duration (the
            # contract's definition; `frac_survey_formula` is the same number under the key rounds 2-3 used, kept stable);
            # `*_executed`: the flops of the algorithm that actually runs, symmetric work counted once
            "roofline": {"bound": "valu_f32", "achieved": ach_s, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_s / PEAK_FP32_TFLOPS,
                         "achieved_survey_formula": ach_s, "frac_survey_formula": ach_s / PEAK_FP32_TFLOPS,
                         "achieved_executed": ach, "frac_executed": ach / PEAK_FP32_TFLOPS,
                         # the matrix cores: the Hessian-block GEMM of the set-up (dense family); share of the f64 matrix peak
                         # = the instruction count of the code x 2048 flops / launch duration (the PMC counter, replayed, agrees)
                         "mfma_util": (mfma_count(h) * 2048.0 * B / (kernel_ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS) if path_used == "dense" else 0.0,
                         "mfma_peak": {"value": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "what": "f64 matrix (= f64 vector) peak: the GEMM runs on v_mfma_f64_16x16x4_f64"},
                         "mfma_instructions_per_solve": mfma_count(h) if path_used == "dense" else 0,
                         "mfma_ops_counter": mfma_ops,
                         "valu_flops_counter": (None if not hw_flops else {
                             "f32_per_launch": hw_flops["f32"], "f64_per_launch": hw_flops["f64"],
                             "tflops": (hw_flops["f32"] + hw_flops["f64"]) / (kernel_ms * 1e-3) / 1e12,
                             # time the counted flops would take at the two vector peaks, over the launch duration
                             "frac_of_vector_peak": (hw_flops["f32"] / (PEAK_FP32_TFLOPS * 1e12) + hw_flops["f64"] / (PEAK_FP64_TFLOPS * 1e12)) / (kernel_ms * 1e-3),
                             "what": "every vector flop the kernel issued (SQ_INSTS_VALU_FLOPS_FP32/_FP64 x 64 lanes, replayed from "
                                     "profiles/pmc_summary.json like `traffic`): redundant and masked-lane work included, so an upper "
                                     "bound of the useful flops that `achieved` counts"}),
                         "traffic": traffic, "traffic_source": traffic_source, "kernel_sha": ksha,
                         "kernel": (f"bmpc::solve_kernel<{h}>" if path_used == "dense" else f"bmpc::stage_kernel<{stage_variant(h)[0]}, {stage_variant(h)[1]}>"),
                         "kernel_ms": kernel_ms,
                         "flops_per_solve": fl_r, "flops_parts": parts, "flops_per_solve_survey_formula": fl_s,
                         "note": "`achieved` / `frac` (= `*_survey_formula`): SURVEY 8(d)'s dense condensed-ADMM flop count "
                                 "F(h, k) = F_setup + k F_iter per solve x the solves of a launch over the average launch duration, "
                                 "against the f32 vector peak (= the f32 matrix peak on CDNA4); `*_executed`: the flops of the "
                                 "algorithm that runs, symmetric work counted once (round 3 reported this one as `frac`).  "
                                 "mfma_util: the one dense horizon-block GEMM of the path -- the torque block of the wrench-space "
                                 "Hessian, Gt_tt = M' M -- runs on the matrix cores (v_mfma_f64_16x16x4_f64: 24 instructions per solve "
                                 "at h = 10, 160 at h = 20; f64 because the row it fills is also the operator of the carried gradient's "
                                 "increments: accumulated in f32 the at-scale error maxima rose tenfold); the wrench-space form leaves no other GEMM "
                                 "(the 12h x 12h Hessian is never formed), and two matrix-core sweeps of the factorisation were built "
                                 "and measured slower (DESIGN 9): the f32 matrix rate equals the f32 vector rate on CDNA4. "
                                 "The path is latency-bound: chains of dependent LDS exchanges, not a pipe",
                         "hbm_algorithmic_bytes_per_solve": hbm_bytes_per_solve(h, use_x_cmd, s["mu"] is not None)},
        }
        if world > 1 or use_dist:
            line["ranks"] = {"kernel_ms_min": min(kernel_ms_ranks), "kernel_ms_max": max(kernel_ms_ranks),
                             "kernel_ms_per_rank": kernel_ms_ranks,
                             "gather_wait_ms_per_step_rank0": (float(np.mean(gev)) if gev else None),
                             # what the process group itself reports (dist.get_world_size() after init_process_group): the
                             # proof of how many ranks the collectives ran over, and on which devices
                             **(group or {}),
                             "note": ("`value` is WEAK scaling: every rank solves its own batch, so it grows with N by construction "
                                      "unless the all_gather of the controls (2 MB per rank and step) hurts; the north_star "
                                      "partition -- ONE 65536 batch of config 4 sharded over the ranks, broadcast + solve + "
                                      "all_gather per step -- is measured in the same run: the `strong` record"
                                      if not strong else
                                      "ONE global batch sharded contiguously; per step broadcast(params) + solve + all_gather")}
        if R2 is not None:
            line["strong"] = {
                "what": "the north_star partition, measured in this same run after the weak region: ONE seeded batch of config 4 "
                        "(h = 10, mixed gait schedules) sharded contiguously over the ranks; per step an RCCL broadcast of the "
                        "parameter block, the solve of the rank's shard, the all_gather of the controls; same warmup / steps / "
                        "barrier + synchronize bracket, slowest rank's time",
                "baseline_config": 4, "total": R2["total"], "batch_per_gpu": R2["B"], "scaling": "strong",
                "value": R2["total"] * R2["steps"] / R2["elapsed"], "unit": "solves/s",
                "ms_per_step": 1e3 * R2["elapsed"] / R2["steps"], "steps": R2["steps"], "warmup": R2["warmup"],
                "kernel_ms_max": R2["kernel_ms"], "kernel_ms_per_rank": R2["kernel_ms_ranks"],
                "collectives_in_step": ["broadcast(params)", "all_gather(controls)"],
                "gather_check": R2["gather_check"], "path": R2["path_used"],
                "mean_iters_rank0": float(R2["iters"].mean()), "not_converged_rank0": int((R2["status"] != 0).sum())}
    if rank == 0 and R["gather_check"]:
        line["config"]["gather_check"] = R["gather_check"]
    if rank == 0 and args.skip_host_path:
        print(json.dumps(line), flush=True)
    elif rank == 0:
        if world == 1:
            # What the tail of a 4096-instance launch costs, measured: the same steps with TWO batches in flight (two handles, two
            # streams, alternating) -- the head of one launch fills the slots the tail of the other leaves idle.  Secondary: `value`
            # stays one batch at a time (a control loop's latency), this is what a caller with independent batches gets.
            _log("two batches in flight")
            try:
                line["two_batches_in_flight"] = two_in_flight()
            except Exception as e:             # a secondary record must not cost the line
                _log(f"two batches in flight failed: {type(e).__name__}: {e}")
                line["two_batches_in_flight"] = {"value": None, "unit": "solves/s", "what": f"unavailable in this run: {type(e).__name__}"}
            _log("dispatch ordered by the previous solve's iteration counts")
            try:
                line["ordered_dispatch"] = ordered_dispatch()
            except Exception as e:
                _log(f"ordered dispatch failed: {type(e).__name__}: {e}")
                line["ordered_dispatch"] = {"value": None, "unit": "solves/s", "what": f"unavailable in this run: {type(e).__name__}"}
        _log("host-pointer (PCIe-inclusive) rate")
        # whole-batch wall clock through the host-pointer entries (inputs from host arrays, results in host arrays, fp64 as REF:300-304)
        try:                                       # (a secondary record must not cost the line)
            xs = [s[k][lo:hi] for k in ("x_fb", "foot", "contact", "phase")]
            kw = dict(x_cmd=s["x_cmd"][lo:hi] if use_x_cmd else None, mu=None if s["mu"] is None else s["mu"][lo:hi])
            reps = 8
            # (i) the handle's page-locked I/O block: inputs converted straight into it, one copy in, one launch, the kernels store
            #     the fp64 results into its host arrays (bmpc_solve_batch_io) -- what a control loop that keeps its buffers calls
            st_i, u_i, _ = solver.solve_inplace(*xs, **kw)
            t0 = time.perf_counter()
            for _ in range(reps):
                solver.solve_inplace(*xs, **kw)
            t_io = (time.perf_counter() - t0) / reps
            same_io = bool(np.array_equal(u_i, o_u.cpu().numpy().astype(np.float64)))
            # (ii) pageable caller arrays (bmpc_solve_batch_f64): packed pinned staging, three prioritised chunks, the unpacking /
            #      widening of a chunk overlapped with the later chunks' solves
            st_h, u_h, _ = solver.solve(*xs, **kw)
            t0 = time.perf_counter()
            for _ in range(reps):
                solver.solve(*xs, out=(st_h, u_h), **kw)       # (output arrays reused)
            t_reuse = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                solver.solve(*xs, **kw)
            t_alloc = (time.perf_counter() - t0) / reps
            same = bool(np.array_equal(u_h, o_u.cpu().numpy().astype(np.float64)))
            dev_rate = (B * args.steps / elapsed) if world == 1 else None
            # `value` keeps the definition of rounds 3-4: the drop-in `BatchSolver.solve` into the caller's own arrays --
            # what a REF:487-style caller gets --, output arrays reused; the in-place path (round 5) is its own record
            frac = lambda t: (B / t) / dev_rate if dev_rate else None
            line["value_incl_pcie"] = {"value": B / t_reuse, "unit": "solves/s", "n_gpus": 1,
                                       "fraction_of_device_resident_rate": frac(t_reuse),
                                       "value_fresh_output_arrays": B / t_alloc,
                                       "bit_identical_to_device_path": same_io and same,
                                       "what": "BatchSolver.solve -> bmpc_solve_batch_f64, the caller's pageable fp64 arrays in and out (output arrays "
                                               "reused): per chunk the inputs packed into one pinned block and copied in, up to 3 chunked launches on "
                                               "prioritised streams, each followed on its stream by one packed device-to-host copy into pinned "
                                               "memory; the calling thread unpacks / widens chunk c while chunks c + 1 .. still solve; one GPU",
                                       "inplace": {"value": B / t_io, "fraction_of_device_resident_rate": frac(t_io),
                                                   "what": "BatchSolver.solve_inplace -> bmpc_solve_batch_io: inputs converted to fp32 straight into the "
                                                           "handle's page-locked I/O block, one copy in, up to 3 chunked launches; the kernels' epilogues "
                                                           "widen to fp64 and store controls + counters straight into the block's host arrays, states go to "
                                                           "HBM and follow by copy engine per chunk (the last chunk's states go the way of the controls); "
                                                           "no unpacking pass, the caller reads the results in place"}}
        except Exception as e:
            _log(f"host-pointer measurement failed: {type(e).__name__}: {e}")
            line["value_incl_pcie"] = {"value": None, "unit": "solves/s", "what": f"unavailable in this run: {type(e).__name__}: {e}"}
        cpu_sample = args.cpu_sample if args.cpu_sample is not None else (B if h == 10 else 256)
        if world == 1 and cpu_sample > 0:
            n = min(cpu_sample, B)
            got = o_u.cpu().numpy().astype(np.float64)
            try:
                ref_full, ref_plain, cb, cert = cpu_baseline(s, h, mpc.dt, n)
            except Exception as e:          # a stuck or failed worker pool must not cost the GPU measurement
                _log(f"cpu_baseline failed: {type(e).__name__}: {e}")
                line["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                        "sample": f"unavailable in this run: {type(e).__name__}"}
                print(json.dumps(line), flush=True)
                return

            def rel(a, b):
                k = len(b)
                return np.abs(a[:k] - b).reshape(k, -1).max(1) / np.maximum(1.0, np.abs(b).reshape(k, -1).max(1))
            r_full, r_plain = rel(got, ref_full), rel(got, ref_plain)
            n_unc = int((~cert).sum())               # references whose own KKT certificate is not tight are not a yardstick
            r_full = r_full[cert] if cert.any() else r_full
            line["cpu_baseline"] = cb
            line["parity"] = {"max_rel_err_vs_oracle": float(r_full.max()), "instances": int(len(r_full)),
                              "oracle_uncertified": n_unc,
                              "p99.9_rel_err_vs_oracle": float(np.quantile(r_full, 0.999)),
                              "p99.9_rel_err_vs_plain_ipm": float(np.quantile(r_plain, 0.999)),
                              "max_rel_err_vs_plain_ipm": float(r_plain.max()), "instances_plain_ipm": int(len(r_plain)),
                              "max_abs_err": float(np.abs(got[:len(ref_full)] - ref_full).max()),
                              "u0_max_rel_err": float((rel(got[:, :1], ref_full[:, :1])[cert] if cert.any() else
                                                       rel(got[:, :1], ref_full[:, :1])).max()),
                              "u0_p99.9_rel_err": float(np.quantile(rel(got[:, :1], ref_full[:, :1])[cert] if cert.any() else
                                                                    rel(got[:, :1], ref_full[:, :1]), 0.999)),
                              "tolerance": 1e-4}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus, argv))      # no torch, no HIP in this process
    run_rank(args)


if __name__ == "__main__":
    main()
