"""The solve kernel's SOURCE executed on the CPU (tests/emu: one thread per lane, barriers for __syncthreads and
for the cross-lane swaps) against the certified optima of the golden fixtures.  This is a logic check of the
kernel -- thread map, LDS exchanges, split of a row over two lanes -- that runs without a GPU; arithmetic
differs from the device in the last bits only (IEEE division instead of v_rcp + Newton).  The GPU parity
tests (-m gpu) remain the parity gate."""
import os
import shutil

import numpy as np
import pytest

from tests import util
from tests.emu import emu

pytestmark = pytest.mark.skipif(not (os.path.exists(emu.CLANG) or shutil.which(emu.CLANG)),
                                reason="host clang (ROCm) not available")


@pytest.mark.parametrize("name,idx", [("cfg4_walking_h10", [3, 17]), ("edge_cases_h10", [0, 4]), ("cfg2_standing_h10", [5]),
                                      ("cfg3_trot_h16", [2]), ("cfg5_mu_h20", [1]),
                                      # round 6: turning / attitude commands (standing, walking); bounds off their defaults (variant 3 of
                                      # gen_regimes: f_min < 0, m_x free, asymmetric tau_min -- one parameter block per call)
                                      ("cfg_cmd_h10", [5, 44]), ("cfg_bounds_h10", [3, 7])])
def test_kernel_source_on_cpu_matches_fixtures(name, idx):
    import __graft_entry__ as ge
    ge.build()
    import biped_mpc_py_amd as bm
    h, half = util.BATCH_FIXTURES[name]
    d = util.load(name)
    mpc = bm.MPC()
    mpc.h = h
    cp = bm.pack_params(mpc, util.biped_of(d, idx[0], bm), half=half)
    mu = d["mu_steps"][idx] if "mu_steps" in d.files and d["mu_steps"].size else None
    o = emu.solve(cp, d["x_fb"][idx], d["foot"][idx], d["contact"][idx], util.phases(d["t"][idx], mpc.dt, h),
                  x_cmd=d["x_cmd"][idx], mu=mu)
    assert (o["status"] == 0).all()
    assert util.rel_err(o["controls"].astype(float), d["controls"][idx]).max() <= util.REL_TOL
    assert util.rel_err(o["states"].astype(float), d["states"][idx]).max() <= util.REL_TOL
    assert o["iters"].max() <= 150
    if "x_ref" in d.files:                   # reference-captured references of these instances
        assert np.abs(o["x_ref"].transpose(0, 2, 1) - d["x_ref"][idx][:, :12]).max() < 1e-6
        assert np.abs(o["foot_ref"].transpose(0, 2, 1) - d["foot_ref"][idx]).max() < 1e-6


@pytest.mark.parametrize("name,key,idx", [("cfg4_walking_h10", None, [3]), ("edge_cases_h10", None, [2]), ("cfg_cmd_h10", None, [61]), ("cfg_bounds_h10", None, [11]), ("cfg_hgen", 14, [0]),
                                          ("cfg_hgen", 26, [0]), ("cfg_hodd", 5, [1]), ("cfg_hodd", 9, [2]), ("cfg_hodd", 1, [0]), ("cfg_hodd", 3, [3])])
def test_stage_kernel_source_on_cpu_matches_fixtures(name, key, idx):
    """The stage-structured kernel (bmpc_stage.hip: Riccati recursion, scans over the steps, phantom steps past the
    horizon at h = 14, the two-wave workgroup at h = 26, the DPP row broadcasts of the two passes emulated lane by lane) on
    the CPU against the fixtures; round 5: odd and short horizons (h = 5: most of the smallest variant's step slots are phantoms; h = 1: all but one)."""
    import __graft_entry__ as ge
    ge.build()
    import biped_mpc_py_amd as bm
    d0 = util.load(name)
    d = d0 if key is None else {k[len("h%d_" % key):]: d0[k] for k in d0.files if k.startswith("h%d_" % key)}
    h = 10 if key is None else key
    half = 5 if key is None else int(d["half"][0])
    mpc = bm.MPC()
    mpc.h = h
    cp = bm.pack_params(mpc, util.biped_of(d, idx[0], bm) if key is None else bm.Biped(), half=half, solver_options=dict(path=2))
    mu = d["mu_steps"][idx] if "mu_steps" in d and d["mu_steps"].size else None
    o = emu.solve(cp, d["x_fb"][idx], d["foot"][idx], d["contact"][idx], util.phases(d["t"][idx], mpc.dt, h),
                  x_cmd=d["x_cmd"][idx], mu=mu)
    assert (o["status"] == 0).all()
    assert util.rel_err(o["controls"].astype(float), d["controls"][idx]).max() <= util.REL_TOL
    assert util.rel_err(o["states"].astype(float), d["states"][idx]).max() <= util.REL_TOL
    assert o["iters"].max() <= 150


def test_warm_start_same_optimum_on_cpu():
    """Warm start (kernel source on the CPU): a second solve that starts from the state the first one left --
    after the state feedback of one control period -- reaches the oracle's optimum of the NEW problem (the saving
    in iterations is statistical: tests/test_gpu_parity.py measures it over a roll-out); a poisoned state buffer
    (NaNs) is ignored."""
    import __graft_entry__ as ge
    ge.build()
    import biped_mpc_py_amd as bm
    from oracle import bmpc_oracle as orc
    h = 10
    s = util.synth_batch(1, h, 77)
    mpc = bm.MPC()
    cp = bm.pack_params(mpc, bm.Biped(), half=5)
    warm = np.full((1, emu.threads(h), 6), np.nan)               # poisoned: the first solve must not read it
    x0 = s["x_fb"].astype(np.float32)
    o0 = emu.solve(cp, x0, s["foot"], s["contact"], s["phase"], warm=warm, warm_load=True, warm_theta=0.5)   # NaN state: cold
    cold0 = emu.solve(cp, x0, s["foot"], s["contact"], s["phase"])
    assert np.array_equal(o0["controls"], cold0["controls"]) and o0["iters"][0] == cold0["iters"][0]
    assert np.isfinite(warm).all()
    x1 = o0["states"][:, 0, :12].copy()                           # state feedback of one control period
    o1 = emu.solve(cp, x1, s["foot"], s["contact"], s["phase"], warm=warm, warm_load=True, warm_theta=0.5)
    cold1 = emu.solve(cp, x1, s["foot"], s["contact"], s["phase"])
    _, ref = orc.solve_mpc(x1[0].astype(float), 0.0, s["foot"][0].astype(np.float32).astype(float), orc.MPC(), orc.Biped(),
                           s["contact"][0])
    assert (o1["status"] == 0).all()
    assert util.rel_err(o1["controls"].astype(float), ref[None]).max() <= util.REL_TOL
    assert util.rel_err(cold1["controls"].astype(float), ref[None]).max() <= util.REL_TOL
    print("iterations: warm", o1["iters"][0], "cold", cold1["iters"][0])     # fewer on average, not for every instance


def test_no_lds_hand_over_without_a_barrier():
    """The emulation under ThreadSanitizer (tests/emu/tsan.py): lanes synchronise only where the GPU does (workgroup
    barrier, pair exchange, wave reduction), so an LDS value handed from one lane to another without an s_barrier in
    between is a data race on the shared-memory image.  The only reports allowed are write/write pairs from ONE source
    line: the lanes that clone the last row store the same value to the same address as the lane they clone."""
    from tests.emu import tsan
    if not os.path.exists(tsan.CLANG):
        pytest.skip("host clang not available")
    try:
        out, err, rc = tsan.run("cfg2_standing_h10", 1)
    except Exception as e:                      # the sanitizer runtime may be missing from a stripped-down image
        pytest.skip(f"sanitized build failed: {e}")
    if "FATAL: ThreadSanitizer" in err:
        pytest.skip("ThreadSanitizer cannot run in this environment")
    assert rc == 0 and "status 0" in out, (rc, out, err[-2000:])
    for acc in tsan.races(err):
        kinds, where = [a[0] for a in acc], {a[1] for a in acc}
        assert all("rite" in k for k in kinds) and len(where) == 1 and "?" not in where, acc
