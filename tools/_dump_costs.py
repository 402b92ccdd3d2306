import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biped_mpc_py_amd as bm
from biped_mpc_py_amd import _lib
from bench import synth
B=4096; h=10
s=bm.BatchSolver(max_batch=B)
x,f,c,p=synth(B,h,1)
dev=torch.device('cuda',0)
prof=torch.zeros((B,8),dtype=torch.int64,device=dev)
_lib.check(s._lib.bmpc_debug_set_profile(s._h, prof.data_ptr()))
tx,tf,tc,tp=[torch.from_numpy(a).to(dev) for a in (x,f,c,p)]
st=torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(2): s.solve_device(tx,tf,tc,tp)
torch.cuda.synchronize()
os.makedirs('gpurun_out',exist_ok=True)
np.savez('gpurun_out/costs.npz',prof=prof.cpu().numpy(),x=x,f=f)
print('saved')
