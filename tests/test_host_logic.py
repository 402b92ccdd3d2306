"""CPU: host-side logic of the drop-in (phase arithmetic, gait table, parameter packing, sharding maths)."""
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
