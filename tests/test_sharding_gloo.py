"""CPU, world_size 2, gloo: the N > 1 path -- parameter broadcast, disjoint shards, gather -- with a
stand-in solver injected (the HIP library cannot run here; the sharding logic does not depend on it)."""
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


def _fake_solver(x_fb, foot):
    # any instance-wise map will do: the result of instance i must not depend on its neighbours
    return np.concatenate([x_fb * 2.0, foot + 1.0], axis=1).astype(np.float32)


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
    rng = np.random.default_rng(0)                        # same global batch on every rank
    x_fb = rng.normal(size=(total, 12)).astype(np.float32)
    foot = rng.normal(size=(total, 6)).astype(np.float32)
    lo, hi, local = sharding.solve_sharded(_fake_solver, dict(x_fb=x_fb, foot=foot), total=total)
    assert (lo, hi) == sharding.shard_bounds(total, rank, world)
    full = sharding.gather_controls(torch.from_numpy(local), total)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_shard_and_gather(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    x_fb = rng.normal(size=(total, 12)).astype(np.float32)
    foot = rng.normal(size=(total, 6)).astype(np.float32)
    expect = _fake_solver(x_fb, foot)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == expect.shape
        assert np.array_equal(got, expect)
