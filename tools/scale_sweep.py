"""Large-batch robustness sweep on the GPU: convergence status, iteration / factorisation distribution and
constraint feasibility for the BASELINE config shapes (no oracle: too slow at these sizes)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import biped_mpc_py_amd as bm
from tests import util

def run(name, B, h, gait, seed, **kw):
    s = util.synth_batch(B, h, seed, gait=gait, **kw)
    mpc = bm.MPC(); mpc.h = h
    solver = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B)
    t0 = time.time()
    st, ct, info = solver.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"])
    dt = time.time() - t0
    it, nf, status = info["iters"], info["nfactor"], info["status"]
    mu = s["mu"] if s["mu"] is not None else np.full((B, h, 2), 0.5)
    f = ct.reshape(B, h, 4, 3)
    viol = 0.0
    for j in range(2):
        fx, fy, fz = f[:, :, j, 0], f[:, :, j, 1], f[:, :, j, 2]
        c = s["contact"][:, :, j].astype(float)
        viol = max(viol, (-fz).max(), (fz - 500 * c).max(), (np.abs(fx) - mu[:, :, j] * fz).max(), (np.abs(fy) - mu[:, :, j] * fz).max(),
                   np.abs(f[:, :, 2 + j, 0]).max(), (np.abs(f[:, :, 2 + j, 1]) - 67 * c).max(), (np.abs(f[:, :, 2 + j, 2]) - 33.5 * c).max())
    print(f"{name}: B={B} h={h} wall {dt*1e3:.0f} ms (incl. PCIe) kernel {solver.last_kernel_ms():.2f} ms | status!=0: {(status!=0).sum()} "
          f"| iters mean {it.mean():.1f} p99 {np.percentile(it,99):.0f} max {it.max()} | nfac mean {nf.mean():.2f} max {nf.max()} "
          f"| max constraint violation {viol:.2e} | nan {np.isnan(ct).sum()}", flush=True)

run("cfg2", 4096, 10, "standing", 1)
run("cfg2x16", 65536, 10, "standing", 21)
run("cfg3", 4096, 16, "walking", 2, vx_cmd=True)
run("cfg4", 65536, 10, "mixed", 3, vx_cmd=True)
run("cfg5(1/8)", 8192, 20, "walking", 4, vx_cmd=True, per_step_mu=True)
run("cfg5", 65536, 20, "walking", 44, vx_cmd=True, per_step_mu=True)
