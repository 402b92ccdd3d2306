#!/usr/bin/env python3
"""bench.py -- throughput of the MPC hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `bmpc_solve_batch_device` on B = 4096
randomised CoM / stance states, horizon 10, double support (BASELINE.json configs[1]), inputs
already resident in HBM.  With --gpus N the driver launches one rank per GPU (torchrun env); every
rank solves its own 4096 instances (weak scaling, no data-path collective: instances are
independent); the timed region is bracketed by barrier + synchronize and the slowest
rank's time is used.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant (only) kernel: algorithmic flops per launch / average launch duration,
                measured with events on the launch stream over the timed region.
  cpu_baseline  the fp64 oracle (oracle/bmpc_oracle.py = CPU port of the reference path) on a bounded
                sample of the same workload over the host cores (N = 1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H = 10
BATCH = 4096
MAX_CPU_WORKERS = 16
PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 vector = fp32 matrix peak (MI355X_MICROARCH.md)


def synth(B, h, seed):
    """SURVEY 8(d) generator, config 2 (standing, double support, reference x_cmd)."""
    rng = np.random.default_rng(seed)
    x_fb = np.concatenate([
        rng.uniform(-0.2, 0.2, (B, 3)), rng.uniform(-0.5, 0.5, (B, 2)), rng.uniform(0.45, 0.60, (B, 1)),
        rng.uniform(-0.5, 0.5, (B, 3)), rng.uniform(-0.5, 0.5, (B, 2)), rng.uniform(-0.2, 0.2, (B, 1))], 1)
    foot = np.zeros((B, 6))
    for j, sgn in enumerate((1.0, -1.0)):
        foot[:, 3 * j + 0] = x_fb[:, 3] - 0.0195 + rng.uniform(-0.05, 0.05, B)
        foot[:, 3 * j + 1] = x_fb[:, 4] + sgn * (0.089 + rng.uniform(-0.03, 0.03, B))
    contact = np.ones((B, h, 2), np.uint8)
    phase = np.zeros(B, np.int32)
    return x_fb.astype(np.float32), foot.astype(np.float32), contact, phase


def flops_per_solve(h, iters, nfactor):
    """Algorithmic flops of one solve (1 MAC = 2 flops), DESIGN.md section 6.
    set-up: wrench-space Hessian rows + gradient; factor: 6x6 block algebra + the 6h x 6h sweep;
    iteration: sparse constraint products, Gt mat-vec, block-diagonal + dense K^-1 application."""
    n = 6 * h
    f_setup = 2.0 * (27 * h * (h - 1) / 2 + 9 * h * h * (h + 1) / 2 + 40 * h * h)
    f_factor = 2.0 * (n ** 3 + 3500 * h)
    f_iter = 2.0 * h * (690 + 48 * h)
    return f_setup + nfactor * f_factor + iters * f_iter, dict(setup=f_setup, factor=f_factor, iteration=f_iter)


def _oracle_one(args):
    x, f, c = args
    from threadpoolctl import threadpool_limits
    from oracle import bmpc_oracle as orc
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        _, ctrl = orc.solve_mpc(x, 0.0, f, orc.MPC(), orc.Biped(), c)
        return ctrl, time.perf_counter() - t0


def cpu_baseline(x_fb, foot, contact, n_sample):
    """Oracle (CPU port of REF:187-304) on the first n_sample instances over all host cores."""
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, MAX_CPU_WORKERS))      # the GPU box grants a 16-CPU share per GPU
    args = [(x_fb[i].astype(float), foot[i].astype(float), contact[i]) for i in range(n_sample)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        out = pool.map(_oracle_one, args)
    wall = time.perf_counter() - t0
    ctrl = np.stack([o[0] for o in out])
    per = float(np.mean([o[1] for o in out]))
    return ctrl, dict(value=n_sample / wall, unit="solves/s", cores=cores, kind="port",
                      sample=f"{n_sample} instances of the same batch, oracle/bmpc_oracle.solve_mpc (fp64 NumPy "
                             f"restatement of the reference + IPM/polish), 1 process per core; "
                             f"{per * 1e3:.0f} ms per solve per core, pool start-up included in the rate")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="instances per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=64, help="instances for the CPU baseline (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (1-GPU box, use with --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import biped_mpc_py_amd as bm

    B = args.batch
    mpc = bm.MPC()
    cp = bm.pack_params(mpc, bm.Biped())
    if world > 1:                                   # C0: one parameter block for every rank
        bm.sharding.broadcast_params(cp, src=0, device=coll_dev)
    solver = bm.BatchSolver(cparams=cp, device=dev_index, max_batch=B)
    x_fb, foot, contact, phase = synth(B, H, seed=1 + 1000 * rank)   # seed 1 = config 2
    t_x, t_f = torch.from_numpy(x_fb).to(dev), torch.from_numpy(foot).to(dev)
    t_c, t_p = torch.from_numpy(contact).to(dev), torch.from_numpy(phase).to(dev)
    o_u = torch.empty((B, H, 12), dtype=torch.float32, device=dev)
    o_s = torch.empty((B, H, 13), dtype=torch.float32, device=dev)
    o_it = torch.empty(B, dtype=torch.int32, device=dev)
    o_st = torch.empty(B, dtype=torch.int32, device=dev)
    o_nf = torch.empty(B, dtype=torch.int32, device=dev)
    o_rs = torch.empty((B, 2), dtype=torch.float32, device=dev)

    def step():
        solver.solve_device(t_x, t_f, t_c, t_p, controls=o_u, states=o_s, iters=o_it, residuals=o_rs,
                            status=o_st, nfactor=o_nf)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # the kernel is launched on torch's CURRENT stream; make that a real (non-null) stream so that
    # the events below are recorded on exactly the stream the kernel runs on
    launch_stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(launch_stream):
        for _ in range(args.warmup):
            step()
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        fence()
        elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch duration over the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    iters = o_it.cpu().numpy()
    nfac = o_nf.cpu().numpy()
    status = o_st.cpu().numpy()
    if rank == 0:
        fl, parts = flops_per_solve(H, float(iters.mean()), float(nfac.mean()))
        achieved = fl * B / (kernel_ms * 1e-3) / 1e12
        traffic = None                     # HBM bytes per launch from the committed PMC passes (profiles/)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as fh:
                pm = json.load(fh)
            if pm.get("batch") == B:
                traffic = pm["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "MPC QP solves/sec (N=10, 2-contact)",
            "value": world * B * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch=4096 randomised CoM/stance states per GPU, horizon 10, "
                                   "double support, inputs resident in HBM",
                       "batch_per_gpu": B, "horizon": H, "residual_dtype": "f64",
                       "mean_iters": float(iters.mean()), "max_iters": int(iters.max()),
                       "mean_factorisations": float(nfac.mean()),
                       "not_converged": int((status != 0).sum())},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
                         "kernel": "bmpc::solve_kernel<10,double>", "kernel_ms": kernel_ms,
                         "flops_per_solve": fl, "flops_parts": parts,
                         "hbm_algorithmic_bytes_per_solve": 4 * (12 + 6 + 1) + 2 * H + 4 * 25 * H + 20},
        }
        if world == 1 and args.cpu_sample > 0:
            n = min(args.cpu_sample, B)
            ref, cb = cpu_baseline(x_fb, foot, contact, n)
            got = o_u.cpu().numpy()[:n].astype(np.float64)
            rel = np.abs(got - ref).reshape(n, -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(n, -1).max(1))
            line["cpu_baseline"] = cb
            line["parity"] = {"max_rel_err_vs_oracle": float(rel.max()), "max_abs_err": float(np.abs(got - ref).max()),
                              "instances": n, "tolerance": 1e-4}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
