"""The instances of a BASELINE config batch where the two kernel families differ most, against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biped_mpc_py_amd as bm
from biped_mpc_py_amd.synth import CONFIGS, synth_batch
from tests import util
from tests.test_gpu_parity import _oracle_controls
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = CONFIGS[cfg]; h = c["h"]
s = synth_batch(B, h, c["seed"], gait=c["gait"], **c["kw"])
out = {}
for path in (1, 2):
    m = bm.MPC(); m.h = h
    sol = bm.BatchSolver(mpc=m, half=s["half"], max_batch=B, solver_options=dict(path=path))
    _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"] if c["kw"].get("vx_cmd") else None, mu=s["mu"], want_states=False)
    out[path] = (u, info)
d = util.rel_err(out[1][0], out[2][0])
idx = np.argsort(-d)[:6]
if not c["kw"].get("vx_cmd"):
    s["x_cmd"] = np.tile(np.array([0, 0, 0, 0, 0, 0.55, 0, 0, 0, 0, 0, 0.0]), (B, 1))
ref = _oracle_controls(s, idx, h)
for k, i in enumerate(idx):
    e1 = util.rel_err(out[1][0][i][None], ref[k][None])[0]; e2 = util.rel_err(out[2][0][i][None], ref[k][None])[0]
    print("inst %5d  dense-vs-stage %.2e | dense err %.2e (iters %d resid %s) | stage err %.2e (iters %d resid %s)" % (
        i, d[i], e1, out[1][1]["iters"][i], out[1][1]["residuals"][i], e2, out[2][1]["iters"][i], out[2][1]["residuals"][i]))
print("median dense-vs-stage %.2e  p99.9 %.2e" % (np.median(d), np.quantile(d, 0.999)))
