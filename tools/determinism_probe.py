"""Run-to-run determinism of the solve on the GPU: python tools/determinism_probe.py LIB.so B [B ...]
solves the config-2 batch of B instances 12 times with the given build of libbmpc.so and counts the instances whose
controls differ bitwise from the first run (must be 0; B >= 768 puts two waves on a SIMD, which is where a barrier
reached with an LDS store in flight showed -- DESIGN.md section 5)."""
import sys, numpy as np, shutil
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
so = sys.argv[1]
shutil.copy(so, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "biped_mpc_py_amd", "libbmpc.so"))
import torch
import biped_mpc_py_amd as bm
from biped_mpc_py_amd import synth
dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[2:]]:
    d = synth.synth_batch(B, 10, 1)
    t = {k: torch.from_numpy(np.ascontiguousarray(d[k].astype(np.float32) if d[k].dtype == np.float64 else d[k])).to(dev) for k in ("x_fb", "foot", "contact", "phase")}
    s = bm.BatchSolver(max_batch=B)
    outs = []
    for rep in range(12):
        ct, _ = s.solve_device(t["x_fb"], t["foot"], t["contact"], t["phase"])
        torch.cuda.synchronize()
        outs.append(ct.cpu().numpy().copy())
    nd = [int(((o != outs[0]) & ~(np.isnan(o) & np.isnan(outs[0]))).any(axis=(1, 2)).sum()) for o in outs[1:]]
    print(so, "B", B, "instances differing from rep 0 over 11 reps:", nd)
    s.close()
