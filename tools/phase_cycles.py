import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import biped_mpc_py_amd as bm
from biped_mpc_py_amd import _lib
from biped_mpc_py_amd.synth import synth_batch
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
cfg=int(sys.argv[2]) if len(sys.argv)>2 else 2        # BASELINE config: 2 (h=10), 3 (h=16), 5 (h=20)
from biped_mpc_py_amd.synth import CONFIGS
_c=CONFIGS[cfg]
h=_c["h"]
_s=synth_batch(B,h,_c["seed"],gait=_c["gait"],**_c["kw"])
_m=bm.MPC(); _m.h=h
s=bm.BatchSolver(mpc=_m,half=_s["half"],max_batch=B)
x,f,c,p=_s["x_fb"].astype(np.float32),_s["foot"].astype(np.float32),_s["contact"],_s["phase"]
xc=torch.from_numpy(_s["x_cmd"].astype(np.float32)).cuda()
mu=None if _s["mu"] is None else torch.from_numpy(_s["mu"].astype(np.float32)).cuda()
dev=torch.device('cuda',0)
prof=torch.zeros((B,16),dtype=torch.int64,device=dev)
_lib.check(s._lib.bmpc_debug_set_profile(s._h, prof.data_ptr()))
tx,tf,tc,tp=[torch.from_numpy(a).to(dev) for a in (x,f,c,p)]
st=torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3):
        s.solve_device(tx,tf,tc,tp,xc,mu)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); s.solve_device(tx,tf,tc,tp,xc,mu); e1.record()
torch.cuda.synchronize()
pr=prof.cpu().numpy().astype(float)
print('B',B,'kernel ms',e0.elapsed_time(e1))
print('cycles mean: setup %.0f blocks %.0f sweeps %.0f total %.0f | iters %.1f nfac %.2f'%tuple(pr[:,:6].mean(0)))
it=pr[:,3]-pr[:,0]-pr[:,1]-pr[:,2]
print('iteration cycles per iter %.0f ; blocks per factor %.0f ; sweep per factor %.0f'%((it/pr[:,4]).mean(),(pr[:,1]/pr[:,5]).mean(),(pr[:,2]/pr[:,5]).mean()))
if '--blocks' in sys.argv:          # (a library built with -DBMPC_PROF_BLOCKS: the phase slots hold the stages of the block algebra)
    print('block algebra, cycles per factorisation: ' + ' '.join('%s %.0f'%(n,v) for n,v in zip(['rvg+sync','D rows','Ka/B/U','inv6+KaB','w0','F/L1/TKa','G images'], (pr[:,8:15]/pr[:,5:6]).mean(0))))
    sys.exit(0)
print('iteration phases, cycles per iteration: ' + ' '.join('%s %.0f'%(n,v) for n,v in zip(['P0','P1','P2','P3','P4','P5','tail'], (pr[:,8:15]/pr[:,4:5]).mean(0))))
nred=np.floor(pr[:,15]/1000); nreb=pr[:,15]-1000*nred
print('tail per solve: %.0f cycles = reductions %.0f (%.1f of them, %.0f each) + rebuilds / last refresh %.0f (%.2f rebuilds) + rest %.0f'%(
    pr[:,14].mean(), pr[:,6].mean(), nred.mean(), (pr[:,6]/np.maximum(nred,1)).mean(), pr[:,7].mean(), nreb.mean(), (pr[:,14]-pr[:,6]-pr[:,7]).mean()))
