"""Accuracy of the dense family against the stage family (which forms its gradient exactly in every iteration) on the turning
batches of round 6, for builds of the library given on the command line -- is what the dense family loses there the drift of
its carried gradient between exact rebuilds?  Each build in its own process.
    python tools/drift_probe.py build_tmp/a.so build_tmp/b.so ...        (ON THE GPU BOX)"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("turning_mixed_h10", 4096, 10, "mixed", 61, dict(vx_cmd=True, turn=True)),
         ("turning_trot_h16", 4096, 16, "walking", 62, dict(vx_cmd=True, turn=True)),
         ("turning_mu_h20", 4096, 20, "walking", 63, dict(vx_cmd=True, per_step_mu=True, turn=True)),
         ("config3_trot_h16", 4096, 16, "walking", 2, dict(vx_cmd=True)),
         ("config5_mu_h20", 4096, 20, "walking", 4, dict(vx_cmd=True, per_step_mu=True))]


def child(tag):
    import biped_mpc_py_amd as bm
    from biped_mpc_py_amd.synth import synth_batch
    for label, B, h, gait, seed, kw in CASES:
        s = synth_batch(B, h, seed, gait=gait, **kw)
        mpc = bm.MPC()
        mpc.h = h
        out = {}
        for path in (1, 2):
            sol = bm.BatchSolver(mpc=mpc, half=s["half"], max_batch=B, solver_options=dict(path=path))
            ms = []
            for _ in range(3):
                _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
                ms.append(sol.last_kernel_ms())
            sol.close()
            out[path] = (u, info, min(ms))
        ref = out[2][0]
        u, info, ms = out[1]
        e = np.abs(u - ref).reshape(B, -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(B, -1).max(1))
        e0 = np.abs(u[:, 0] - ref[:, 0]).max(1) / np.maximum(1.0, np.abs(ref[:, 0]).max(1))
        print("%-10s %-18s dense vs stage: all max %.2e p99.9 %.2e >5e-6: %3d | u0 max %.2e | it %.2f max %d nf %.2f | %.4f ms (stage %.4f)" % (
            tag, label, e.max(), np.quantile(e, 0.999), int((e > 5e-6).sum()), e0.max(), info["iters"].mean(), info["iters"].max(), info["nfactor"].mean(), ms, out[2][2]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for lib in sys.argv[1:]:
            shutil.copy(os.path.join(ROOT, lib), os.path.join(ROOT, "biped_mpc_py_amd", "libbmpc.so"))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", os.path.basename(lib)[:-3]])
