#!/usr/bin/env python3
"""Initial row classifications for the adaptive-penalty ADMM, on the NumPy model of the product's algorithm (oracle/ws_model.py)
-- VERDICT r5 item 3, CPU only:  python tools/classify_explore.py [stand10|mixed10|walk16|walk20]
How right a guess of the final active set has to be to pay (final classes with a share of the rows flipped at random), what the
candidates reach (rows violated by the equality-constrained minimiser; a re-classification after three iterations at that
guess's limits; the batch's most frequent class per row), per-class starting penalties, and the default run's own classes over
the iterations.  Cost = iterations + 10.1 x factorisations (h = 10).  Findings: docs/history_r06.md.  Test infrastructure /
tuning aid: the product never imports this."""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ws_model as wm
from tools.schedule_explore import load_sets
sets=load_sets()
name=sys.argv[1] if len(sys.argv)>1 else 'stand10'
z=sets[name]
n=256
h=int(z['h'])
def mk():
    P=wm.Params(h=h,half=int(z['half']))
    P.adapt_start,P.adapt_every,P.adapt_early,P.adapt_late,P.adapt_busy,P.adapt_flips,P.kappa_confirm,P.confirm_from=(5,5,3,20,10,1,400.0,3)
    P.rho=0.03; P.rho_eq_scale=30/P.rho; P.slow_guard=1e-6; P.max_iter=400
    return P
def run(P,tag):
    mu=z['mu'][:n] if z['mu'].size else None
    st,u,info=wm.solve_batch(P,z['x_fb'][:n],z['foot'][:n],z['contact'][:n],z['phase'][:n],x_cmd=z['x_cmd'][:n],mu=mu,dtype=np.float32,res_dtype=np.float64,return_debug=True)
    ref=z['ref'][:n]
    rel=np.abs(u-ref).reshape(n,-1).max(1)/np.maximum(1,np.abs(ref).reshape(n,-1).max(1))
    c=info['iters']+10.1*info['n_factor']
    print('%-44s it %5.1f nf %4.2f cost %6.1f p95 %6.1f max %6.1f maxit %d err %.1e'%(tag,info['iters'].mean(),info['n_factor'].mean(),c.mean(),np.percentile(c,95),c.max(),info['iters'].max(),rel.max()),flush=True)
    return info
info=run(mk(),'default')
# rows: 0-2 force box, 3-5 moment box, 6-9 friction, 10-11 line foot
def rvi(fb,mb,fr,lf):
    v=np.empty(12); v[0:3]=fb; v[3:6]=mb; v[6:10]=fr; v[10:12]=lf
    return np.broadcast_to(v,(n,h,2,12)).copy()
for fb,mb,fr,lf in [(0.03,0.003,0.03,0.03),(0.03,3e-4,0.03,0.03),(0.03,0.0015,0.03,0.03),(0.1,0.003,0.03,0.03),(0.1,0.003,0.1,0.1),(0.06,0.003,0.06,0.3),(0.03,0.003,0.03,0.3),(0.03,0.003,0.03,1.0),(0.2,0.003,0.03,0.03)]:
    P=mk(); P.rv_init=rvi(fb,mb,fr,lf); run(P,'rho0 fbox %g mbox %g fric %g lf %g'%(fb,mb,fr,lf))
print('--- guesses from the equality-constrained minimiser')
zz,y,l,uu=info['z'],info['y'],info['l'],info['u']
eq=l==uu
act_true=((zz<=l)|(zz>=uu))&(y!=0)
hi=np.empty(12); hi[[0,1,2,6,7,8,9]]=1.0; hi[[3,4,5,10,11]]=100.0
P1=mk(); P1.rv_init=np.full((n,h,2,12),3e-4); P1.accel=False
mu=z['mu'][:n] if z['mu'].size else None
_,_,i1=wm.solve_batch(P1,z['x_fb'][:n],z['foot'][:n],z['contact'][:n],z['phase'][:n],x_cmd=z['x_cmd'][:n],mu=mu,dtype=np.float32,res_dtype=np.float64,return_debug=True,iters=1)
xt=i1['x']/1.6
Ax=np.einsum('bhfri,bhfi->bhfr',i1['A'],xt)
g=(Ax<l)|(Ax>uu)
g=g&~eq
at=act_true&~eq
print('guess A: violated by the unconstrained minimiser: predicted active %.1f true active %.1f  wrong rows per instance %.1f (false pos %.1f, false neg %.1f) of %d'%(g.reshape(n,-1).sum(1).mean(),at.reshape(n,-1).sum(1).mean(),(g^at).reshape(n,-1).sum(1).mean(),(g&~at).reshape(n,-1).sum(1).mean(),(~g&at).reshape(n,-1).sum(1).mean(), (~eq).reshape(n,-1).sum(1).mean()))
for k in (3.,20.):
    P=mk(); P.rv_init=np.where(g,np.minimum(0.03*k,hi),np.maximum(0.03/k,3e-4)); run(P,'guess A, rho0 */ %g'%k)
# guess B: after the first 5 iterations of the default run -> what the default already does at iteration 5 (reference)
# guess C: two-stage: unconstrained -> fix violated rows at bounds (rho hi), resolve once, re-classify by violation or multiplier sign
P2=mk(); P2.rv_init=np.where(g,hi,3e-4); P2.accel=False
_,_,i2=wm.solve_batch(P2,z['x_fb'][:n],z['foot'][:n],z['contact'][:n],z['phase'][:n],x_cmd=z['x_cmd'][:n],mu=mu,dtype=np.float32,res_dtype=np.float64,return_debug=True,iters=3)
z2,y2=i2['z'],i2['y']
g2=((z2<=l)|(z2>=uu))&(y2!=0)&~eq
print('guess C (3 iterations at the limits of guess A): wrong rows per instance %.1f (fp %.1f fn %.1f)'%((g2^at).reshape(n,-1).sum(1).mean(),(g2&~at).reshape(n,-1).sum(1).mean(),(~g2&at).reshape(n,-1).sum(1).mean()))
# what the default run's own classes look like over time
for its in (5,10,15,20,30):
    Pd=mk()
    _,_,idb=wm.solve_batch(Pd,z['x_fb'][:n],z['foot'][:n],z['contact'][:n],z['phase'][:n],x_cmd=z['x_cmd'][:n],mu=mu,dtype=np.float32,res_dtype=np.float64,return_debug=True,iters=its)
    gd=((idb['z']<=l)|(idb['z']>=uu))&(idb['y']!=0)&~eq
    print('default run, classes at iteration %2d: wrong rows per instance %.1f (fp %.1f fn %.1f)'%(its,(gd^at).reshape(n,-1).sum(1).mean(),(gd&~at).reshape(n,-1).sum(1).mean(),(~gd&at).reshape(n,-1).sum(1).mean()))
mode=(at.mean(0)>0.5)
gm=np.broadcast_to(mode,at.shape)
print('guess S (the batch\'s most frequent class per row, i.e. a nominal standing solution): wrong rows per instance %.1f'%((gm^at).reshape(n,-1).sum(1).mean()))
for k in (3.,):
    P=mk(); P.rv_init=np.where(gm,np.minimum(0.03*k,hi),np.maximum(0.03/k,3e-4)); run(P,'guess S, rho0 */ %g'%k)
