"""The kernels either side of the solve (FK, low-level control, gait scheduler, roll-out feedback)
at B = 65536, for a `rocprofv3 --kernel-trace --stats` line:   rocprofv3 --kernel-trace --stats -- python3 tools/lowlevel_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biped_mpc_py_amd as bm            # noqa: E402
from biped_mpc_py_amd import _lib        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
s = bm.BatchSolver(max_batch=B)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.uniform(-0.3, 0.3, (B, 12)).astype(np.float32)).to(dev)
q = torch.from_numpy(rng.uniform(-1, 1, (B, 10)).astype(np.float32)).to(dev)
qd = torch.from_numpy(rng.uniform(-1, 1, (B, 10)).astype(np.float32)).to(dev)
u0 = torch.from_numpy(rng.uniform(0, 50, (B, 12)).astype(np.float32)).to(dev)
t = torch.from_numpy(rng.uniform(0, 2, B)).to(dev)
c0 = torch.ones((B, 2), dtype=torch.uint8, device=dev)
pf = torch.empty((B, 6), dtype=torch.float32, device=dev)
tau = torch.empty((B, 10), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(10):
    _lib.check(s._lib.bmpc_foot_position_world_device(s._h, B, x.data_ptr(), q.data_ptr(), pf.data_ptr(), st))
    _lib.check(s._lib.bmpc_low_level_control_device(s._h, B, x.data_ptr(), t.data_ptr(), pf.data_ptr(), q.data_ptr(), qd.data_ptr(),
                                                    c0.data_ptr(), u0.data_ptr(), tau.data_ptr(), st))
    s.contact_sequence_device(t)
torch.cuda.synchronize()
print("ok", B)
