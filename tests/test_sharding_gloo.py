"""CPU, world_size 2, gloo: the N > 1 path -- parameter broadcast, disjoint shards, gather.  The HIP library
cannot compute here, so the kernel launch is the one thing replaced: every rank goes through
`sharding.solve_sharded` with the REAL host marshalling of `BatchSolver` (dtype / shape / contiguity checks of
the C-ABI call) on a slice of ONE seeded SURVEY 8(d) batch and hands the marshalled arrays to a stand-in
instance-wise map.  The same flow with the real kernel runs on the GPU box
(tests/test_gpu_parity.py::test_bench_two_ranks_strong_scaling: bench.py --gpus 2 --backend gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stand_in_kernel(B, x_fb, foot, contact, phase, x_cmd, mu, h):
    # any instance-wise map of everything the kernel reads: instance i must not depend on its neighbours
    assert x_fb.dtype == np.float32 and x_fb.shape == (B, 12) and x_fb.flags.c_contiguous
    assert foot.dtype == np.float32 and foot.shape == (B, 6)
    assert contact.dtype == np.uint8 and contact.shape == (B, h, 2) and phase.dtype == np.int32
    u = np.zeros((B, h, 12), np.float32)
    u += x_fb[:, None, :] * 2.0
    u[:, :, :6] += foot[:, None, :]
    u[:, :, 6:8] += contact
    u[:, :, 8] += phase[:, None]
    if x_cmd is not None:
        u[:, :, 9] += x_cmd[:, None, 9]
    if mu is not None:
        u[:, :, 10:12] += mu
    return u


def _solve_shard(marshal, h):
    def fn(**shard):
        B, x_fb, foot, contact, phase, x_cmd, mu = marshal(shard["x_fb"], shard["foot"], shard["contact"], shard["phase"],
                                                           shard["x_cmd"], shard["mu"])
        return _stand_in_kernel(B, x_fb, foot, contact, phase, x_cmd, mu, h)
    return fn


def _inputs(total, h):
    from biped_mpc_py_amd.synth import synth_batch
    s = synth_batch(total, h, 3, gait="mixed", vx_cmd=True, per_step_mu=True)
    return {k: s[k] for k in ("x_fb", "foot", "contact", "phase", "x_cmd", "mu")}


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    ge.build()
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import sharding
    # C0: rank 0 owns the (modified) parameter block, everyone ends up with it
    mpc = bm.MPC()
    if rank == 0:
        mpc.x_cmd[5] = 0.61
    cp = bm.pack_params(mpc, bm.Biped(), solver_options=dict(max_iter=77) if rank == 0 else None)
    sharding.broadcast_params(cp, src=0)
    assert cp.x_cmd[5] == 0.61 and cp.max_iter == 77
    h = 10
    host = object.__new__(bm.BatchSolver)                 # the marshalling half of a solver: no handle, no device
    host.h = h
    lo, hi, local = sharding.solve_sharded(_solve_shard(host._marshal, h), _inputs(total, h), total=total)
    assert (lo, hi) == sharding.shard_bounds(total, rank, world) and local.shape[0] == hi - lo
    full = sharding.gather_controls(torch.from_numpy(local), total)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_shard_and_gather(tmp_path, total):
    import biped_mpc_py_amd as bm
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    h = 10
    host = object.__new__(bm.BatchSolver)
    host.h = h
    expect = _solve_shard(host._marshal, h)(**_inputs(total, h))       # the N = 1 result of the whole batch
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == expect.shape
        assert np.array_equal(got, expect)


def test_eight_rank_shard_and_gather(tmp_path):
    """The driver's N = 8 partition (SCALE / MULTICHIP run bare `bench.py --gpus 8`): eight ranks over gloo, ONE seeded batch
    whose size is not a multiple of eight (65539: three shards of 8193 and five of 8192 -- `shard_bounds` hands the remainder to
    the first ranks), parameter broadcast from rank 0, disjoint contiguous shards through the real host marshalling, one
    all_gather of equal-sized padded shards, and on EVERY rank the gathered controls equal the N = 1 result of the whole
    batch, bit for bit."""
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import sharding
    world, total = 8, 65539
    bounds = [sharding.shard_bounds(total, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == total and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    sizes = [hi - lo for lo, hi in bounds]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    h = 10
    host = object.__new__(bm.BatchSolver)
    host.h = h
    expect = _solve_shard(host._marshal, h)(**_inputs(total, h))
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == expect.shape and np.array_equal(got, expect)
