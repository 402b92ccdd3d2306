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
                                      ("cfg3_trot_h16", [2]), ("cfg5_mu_h20", [1])])
def test_kernel_source_on_cpu_matches_fixtures(name, idx):
    import __graft_entry__ as ge
    ge.build()
    import biped_mpc_py_amd as bm
    h, half = util.BATCH_FIXTURES[name]
    d = util.load(name)
    mpc = bm.MPC()
    mpc.h = h
    cp = bm.pack_params(mpc, bm.Biped(), half=half)
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
