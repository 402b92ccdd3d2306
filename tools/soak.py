"""Convergence soak (GPU box): random batches of the four BASELINE config shapes over a range of seeds; prints the
instances that did not converge and the worst iteration count.  python tools/soak.py [seed_lo seed_hi [long]]
("stage": six shapes of the long horizons h = 24 .. 40 on the stage-structured kernels, 8192 instances per seed each;
"long": four more shapes of the long horizons -- h = 16 walking / mixed, h = 20 walking / standing with per-step
friction -- 16384 instances per seed each)"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import biped_mpc_py_amd as bm
from tests import util
tot=0; bad=0; worst=0
SHAPES = ((10,'mixed',dict(vx_cmd=True)), (10,'standing',{}), (16,'walking',dict(vx_cmd=True)), (20,'walking',dict(vx_cmd=True, per_step_mu=True)))
if len(sys.argv) > 3 and sys.argv[3] == "long":
    SHAPES = ((16,'walking',dict(vx_cmd=True)), (20,'walking',dict(vx_cmd=True, per_step_mu=True)), (16,'mixed',dict(vx_cmd=True)), (20,'standing',dict(per_step_mu=True)))
if len(sys.argv) > 3 and sys.argv[3] == "stage":       # the stage-structured family: every NP / NW variant, long horizons twice
    SHAPES = ((24,'walking',dict(vx_cmd=True, per_step_mu=True)), (28,'mixed',dict(vx_cmd=True)), (32,'walking',dict(vx_cmd=True, per_step_mu=True)),
              (40,'walking',dict(vx_cmd=True, per_step_mu=True)), (40,'mixed',dict(vx_cmd=True)), (36,'standing',dict(per_step_mu=True)))
for h, gait, kw in SHAPES:
    mpc=bm.MPC(); mpc.h=h
    B=65536 if h==10 else (16384 if h <= 20 else 8192)
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 540):
        s=util.synth_batch(B,h,seed,gait=gait,**kw)
        sol=bm.BatchSolver(mpc=mpc, half=s['half'], max_batch=B)
        _,u,info=sol.solve(s['x_fb'],s['foot'],s['contact'],s['phase'],x_cmd=s['x_cmd'],mu=s['mu'],want_states=False)
        sol.close()
        nb=int((info['status']!=0).sum()); tot+=B; bad+=nb; worst=max(worst,int(info['iters'].max()))
        print(h,gait,seed,'not converged',nb,'iters mean %.1f max %d'%(info['iters'].mean(), info['iters'].max()), 'nan', int(np.isnan(u).any()), flush=True)
print('TOTAL', tot, 'instances, not converged', bad, 'worst iterations', worst)
