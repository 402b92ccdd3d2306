"""Parameter-range soak: 16384 instances per case and kernel family at h = 10 / 20, counting non-converged instances."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biped_mpc_py_amd as bm
from tests import util
cases = {"Q_x100": lambda m, b: setattr(m, "Q", np.asarray(m.Q, float) * 100), "R_div100": lambda m, b: setattr(m, "R", np.asarray(m.R, float) / 100),
         "R_x100": lambda m, b: setattr(m, "R", np.asarray(m.R, float) * 100), "Q_x10": lambda m, b: setattr(m, "Q", np.asarray(m.Q, float) * 10),
         "dt_0.05": lambda m, b: setattr(m, "dt", 0.05), "m_20": lambda m, b: setattr(b, "m", 20.0)}
B = 16384
for h, gait in ((10, "mixed"), (10, "standing"), (20, "standing")):
    s = util.synth_batch(B, h, 77 + h, gait=gait, vx_cmd=(gait != "standing"), per_step_mu=(h >= 20))
    for name, f in cases.items():
        for path in (1, 2):
            m, b = bm.MPC(), bm.Biped(); m.h = h; f(m, b)
            sol = bm.BatchSolver(mpc=m, biped=b, half=s["half"], max_batch=B, solver_options=dict(path=path))
            _, u, info = sol.solve(s["x_fb"], s["foot"], s["contact"], s["phase"], x_cmd=s["x_cmd"], mu=s["mu"], want_states=False)
            sol.close()
            st = info["status"]
            print("h %d %-8s %-8s path %d: not converged %d (NaN %d) iters mean %.1f max %d" % (h, gait, name, path, int((st != 0).sum()), int((st == 2).sum()),
                                                                                      info["iters"].mean(), info["iters"].max()), flush=True)
