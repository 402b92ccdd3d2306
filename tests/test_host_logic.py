"""CPU: host-side logic of the drop-in (phase arithmetic, gait table, parameter packing, sharding maths)."""
import os

import numpy as np
import pytest

from tests import util


def test_phase_and_contact_sequence_match_reference():
    import biped_mpc_py_amd as bm
    d = util.load("unit_functions")
    mpc = bm.MPC()
    for t, c in zip(d["t_list"], d["contact_seq"]):
        assert np.array_equal(bm.get_contact_sequence(float(t), mpc), c)
        assert bm.phase_index(float(t), mpc) == int(float(t) // 0.04) % 10
    # floating floor division is part of the spec
    assert bm.phase_index(0.12, mpc) == int(0.12 // 0.04) % 10
    mpc.h = 16
    assert bm.get_contact_sequence(0.0, mpc, half=8).shape == (16, 2)
    assert bm.get_contact_sequence(0.0, mpc).shape == (10, 2)            # reference quirk: 10 rows


def test_vectorised_phase_is_the_reference_floor_division():
    """`phase_indices` (np.floor_divide, no Python loop over the batch) against REF:56-57's `int(t // dt) % h` on and
    next to 3000 step boundaries, for three step lengths: floating floor division at a boundary is part of the spec."""
    import biped_mpc_py_amd as bm
    for dt, h in ((0.04, 10), (0.02, 16), (0.05, 40)):
        k = np.arange(3000)
        base = k * dt
        t = np.concatenate([base, np.nextafter(base, np.inf), np.nextafter(base, -np.inf).clip(0), base + 0.5 * dt,
                            k.astype(float) * 0.04 * 1.0000001])
        want = np.array([int(float(v) // dt) % h for v in t], np.int32)
        got = bm.phase_indices(t, dt, h)
        assert got.dtype == np.int32 and np.array_equal(got, want)


def test_contact_table_validation_fast_path():
    from biped_mpc_py_amd.api import _contact_u8
    c = np.ones((3, 10, 2), np.uint8)
    assert _contact_u8(c, 3, 10).dtype == np.uint8
    assert np.array_equal(_contact_u8(c.astype(bool), 3, 10), c)
    assert np.array_equal(_contact_u8(c.astype(float), 3, 10), c)
    assert np.array_equal(_contact_u8(c.astype(np.int64).tolist(), 3, 10), c)
    for bad in (c * 2, c.astype(float) * 0.5, -c.astype(int)):
        with pytest.raises(ValueError):
            _contact_u8(bad, 3, 10)
    with pytest.raises(ValueError):
        _contact_u8(np.ones((3, 9, 2), np.uint8), 3, 10)


def test_pack_params_maps_reference_objects():
    import __graft_entry__ as ge
    ge.build()
    import biped_mpc_py_amd as bm
    mpc, biped = bm.MPC(), bm.Biped()
    mpc.x_cmd = np.arange(12) * 0.1
    biped.mu = 0.7
    biped.f_max = np.array([[400.0], [300.0], [200.0]])
    cp = bm.pack_params(mpc, biped, solver_options=dict(max_iter=123, rho=0.05))
    assert list(cp.x_cmd) == pytest.approx(list(np.arange(12) * 0.1))
    assert cp.mu == 0.7 and list(cp.f_max) == [400.0, 300.0, 200.0]
    assert cp.max_iter == 123 and cp.rho == 0.05 and cp.half == 5
    assert list(cp.I) == [0.932, 0, 0, 0, 0.942, 0, 0, 0, 0.0711]
    with pytest.raises(KeyError):
        bm.pack_params(mpc, biped, solver_options=dict(nope=1))
    # the reference's own attribute bags are accepted as they are (duck typing)
    class RefLikeMPC:
        h, dt, kv = 10, 0.04, 0.01
        x_cmd = [0] * 12
        Q = [1] * 13
        R = [1e-4] * 12
    assert bm.pack_params(RefLikeMPC(), biped).dt == 0.04


def test_shard_bounds_cover_batch_exactly():
    from biped_mpc_py_amd.sharding import shard_bounds
    for total in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_synthetic_generator_is_deterministic():
    a = util.synth_batch(8, 10, 5, gait="mixed", vx_cmd=True)
    b = util.synth_batch(8, 10, 5, gait="mixed", vx_cmd=True)
    assert all(np.array_equal(a[k], b[k]) for k in ("x_fb", "foot", "contact", "phase", "x_cmd"))
    assert set(np.unique(a["contact"])) <= {0, 1}


def test_bench_self_launch_propagates_rank_failure():
    """`python bench.py --gpus 2` with no launcher environment spawns its ranks itself (the parent never
    touches torch or HIP).  Without a GPU every rank exits with "needs a GPU": the parent must come back
    with a non-zero code and no JSON line instead of hanging in a collective."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, timeout=300)
    import torch
    if torch.cuda.is_available():                        # on a GPU box the second rank has no device of its own
        assert p.returncode != 0 or b'"n_gpus": 2' in p.stdout
    else:
        assert p.returncode != 0 and b"{" not in p.stdout
        assert b"needs a GPU" in p.stderr
    assert time.time() - t0 < 300


def test_bench_flop_counts_and_bytes():
    import bench
    # h = 10 -> F_setup = 2.712 MFLOP, F_iter = 35.2 kFLOP; h = 20 -> 20.67 MFLOP, 128.0 kFLOP
    assert abs(bench.flops_survey(10, 0) - 2.712e6) < 2e3 and abs(bench.flops_survey(10, 1) - bench.flops_survey(10, 0) - 35.2e3) < 1
    assert abs(bench.flops_survey(20, 0) - 20.67e6) < 1e4 and abs(bench.flops_survey(20, 1) - bench.flops_survey(20, 0) - 128.0e3) < 1
    assert bench.hbm_bytes_per_solve(10, False, False) == 1116
    fl, parts = bench.flops_run(10, 56.0, 6.0)
    assert fl == parts["setup"] + 6.0 * parts["factor"] + 56.0 * parts["iteration"]


def test_pack_params_half_defaults_follow_the_horizon():
    """`half` defaults to what bmpc_default_params chose: the reference's 5 at h = 10 (REF:101-105), h / 2 otherwise."""
    import biped_mpc_py_amd as bm
    for h, want in ((10, 5), (16, 8), (20, 10)):
        mpc = bm.MPC()
        mpc.h = h
        assert bm.pack_params(mpc, bm.Biped()).half == want
        assert bm.pack_params(mpc, bm.Biped(), half=3).half == 3
        mpc.half = 4
        assert bm.pack_params(mpc, bm.Biped()).half == 4


def test_solver_defaults_follow_the_horizon():
    """`bmpc_default_params(h)`: the re-classification schedule follows the cost of a factorisation relative to an iteration
    (DESIGN.md section 3).  h <= 12 (round 5): 5 apart for the first three (iterations 5, 10, 15), then 20 apart -- 10 after a
    re-classification that still found more than one row in another class --, confirmation (kappa_confirm 400) from the fourth
    on; h = 16: iterations 10, 20, then 20 apart, confirmation from the third on; h = 20: every 20 from iteration 20
    (rho0 = 0.045), one rate, no confirmation.  1500 instead
    of 1000 iterations beyond h = 12 and 60 factorisations allowed everywhere; rho_eq = rho x rho_eq_scale stays 30."""
    import biped_mpc_py_amd as bm
    want = {10: (5, 5, 3, 20, 10, 1000, 0.03), 16: (10, 10, 2, 20, 0, 1500, 0.03), 20: (20, 20, 0, 0, 0, 1500, 0.045)}
    for h, (every, start, early, late, busy, max_iter, rho) in want.items():
        mpc = bm.MPC()
        mpc.h = h
        cp = bm.pack_params(mpc, bm.Biped())
        assert (cp.adapt_every, cp.adapt_start, cp.adapt_early, cp.adapt_late, cp.adapt_busy, cp.max_iter) == (every, start, early, late, busy, max_iter), h
        assert (cp.confirm_from, cp.kappa_confirm) == {10: (3, 400.0), 16: (2, 400.0), 20: (0, 0.0)}[h] and cp.adapt_flips == 1, h
        assert abs(cp.rho - rho) < 1e-12 and abs(cp.rho * cp.rho_eq_scale - 30.0) < 1e-9, h
        assert cp.check_every == 5 and cp.warm_adapt_start == 5 and cp.kappa == 20.0 and cp.max_refactor == 60
    cp = bm.pack_params(bm.MPC(), bm.Biped(), solver_options=dict(adapt_every=15, rho=0.02))      # overrides still apply
    assert cp.adapt_every == 15 and cp.rho == 0.02


def test_rescue_mode_is_a_solver_option_with_default_auto():
    """bmpc_params.rescue (include/bmpc.h enum bmpc_rescue_mode): AUTO by default, settable through solver_options, and an
    unknown mode is refused by the library's parameter check (host arithmetic, no device)."""
    import ctypes as C
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd import _lib
    from biped_mpc_py_amd.params import RESCUE_AUTO, RESCUE_OFF, RESCUE_ON
    assert (RESCUE_AUTO, RESCUE_OFF, RESCUE_ON) == (-1, 0, 1)
    cp = bm.pack_params(bm.MPC(), bm.Biped())
    assert cp.rescue == RESCUE_AUTO and cp.accel == 1
    assert bm.pack_params(bm.MPC(), bm.Biped(), solver_options=dict(accel=0)).accel == 0
    assert bm.pack_params(bm.MPC(), bm.Biped(), solver_options=dict(rescue=RESCUE_ON)).rescue == RESCUE_ON
    out = (C.c_double * 5)()
    assert _lib.load().bmpc_effective_penalties(C.byref(cp), out) == 0
    cp.rescue = 3
    assert _lib.load().bmpc_effective_penalties(C.byref(cp), out) != 0
    assert b"rescue" in _lib.load().bmpc_last_error()


def test_kernel_source_hash_covers_code_not_commentary():
    """synth.kernel_source_hash strips comments before hashing; that is only sound while no string literal of the kernel
    sources contains a comment opener."""
    import re
    from biped_mpc_py_amd.synth import kernel_source_hash
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "biped_mpc_py_amd", "csrc")
    for name in ("bmpc_kernels.hip", "bmpc_stage.hip", "bmpc_capi.hip", "bmpc_lowlevel.hip", os.path.join("..", "..", "include", "bmpc.h")):
        text = open(os.path.join(here, name), encoding="utf-8").read()
        for m in re.finditer(r'"([^"\n]*)"', text):
            assert "//" not in m.group(1) and "/*" not in m.group(1), (name, m.group(0))
    assert re.fullmatch(r"[0-9a-f]{16}", kernel_source_hash())
    # the hash covers everything that decides the code object: the compile flags are part of it
    import __graft_entry__ as ge
    h0 = kernel_source_hash()
    saved = list(ge.KERNEL_FLAGS)
    try:
        ge.KERNEL_FLAGS.append("-DSOMETHING")
        assert kernel_source_hash() != h0
    finally:
        ge.KERNEL_FLAGS[:] = saved
    assert kernel_source_hash() == h0
