"""Conditioning of a pinned scenario's Gauss-Newton systems, on the CPU (oracle only): python tools/cond_scenario.py <scenario name>
Per frame: iterations, n_valid, cond(H) of the last system, smallest / largest LDL^T pivot, and the distance of three solvers from the restated Eigen
full-pivot QR (the one-lane LDL^T of rounds 2-6, the rows-in-lanes LDL^T of the end of round 6 in numpy with exact reciprocals, numpy's LU); then frame 0
iteration by iteration (fresh oracles with optimization_iter_num = 1, 2, ...).  Written for fuzz355 (tools/gpu_vs_ref_fuzz.py: 7.3e-8 m from the compiled
reference in frame 0, inside the 1e-4 contract, outside that tool's 1e-8): 13 valid points, cond(H) = 3e8, a step of 4.7 -- the summation order of H alone
(1e-16 relative) moves such a solution by up to 1e-7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import refpin, util
from oracle import oracle as O
name=sys.argv[1] if len(sys.argv)>1 else 'fuzz355'
sc=refpin.make_scenario(name)
mode,y,loc=sc["mode"],sc["y"],bool(sc.get("loc",False))
print(mode, {k:y[k] for k in y if 'iter' in k or 'thres' in k}, 'cap', sc.get('ivox_capacity'), 'frames', len(sc['frames']), 'pts', [f['scan'].shape[0] for f in sc['frames']])
o=util.oracle_for(mode,y,loc)
if sc.get('ivox_capacity') is not None: o.set_ivox_capacity(sc['ivox_capacity'])
o.AddCloudToLocalMap(*sc["init_clouds"])
Tprev=np.eye(4)
def ldlt_old(H,g):
    L=np.zeros((6,6));D=np.zeros(6)
    for j in range(6):
        d=H[j,j]
        for k in range(j): d-= (L[j,k]*L[j,k])*D[k]
        D[j]=d; inv=1.0/d
        for i in range(j+1,6):
            s=H[i,j]
            for k in range(j): s-=(L[i,k]*L[j,k])*D[k]
            L[i,j]=s*inv
    y=np.zeros(6)
    for i in range(6):
        s=g[i]
        for k in range(i): s-=L[i,k]*y[k]
        y[i]=s
    y=y/D
    for i in range(5,-1,-1):
        s=y[i]
        for k in range(i+1,6): s-=L[k,i]*y[k]
        y[i]=s
    return y,D
def ldlt_new(H,g):
    A=H.copy();U=np.zeros((6,6));inv=np.zeros(6);D=np.zeros(6)
    for j in range(6):
        p=A[j,:].copy(); D[j]=p[j]; iv=1.0/p[j]; inv[j]=iv
        for i in range(6):
            l=A[i,j]*iv if i>j else 0.0
            for k in range(j+1,6): A[i,k]-=l*p[k]
        for k in range(j+1,6): U[j,k]=p[k]*iv
    y=np.zeros(6)
    for i in range(6):
        s=g[i]
        for k in range(i): s-=U[k,i]*y[k]
        y[i]=s
    y=y*inv
    for i in range(5,-1,-1):
        s=y[i]
        for k in range(i+1,6): s-=U[i,k]*y[k]
        y[i]=s
    return y,D
for k,f in enumerate(sc["frames"]):
    guess=f["absolute_guess"] if "absolute_guess" in f else Tprev@f["guess_step"]
    ok,T=o.Match(f["scan"],np.array(guess,dtype=np.float64),src1=f["corner"],update_map=True)
    H,g=o.last_system()
    c=np.linalg.cond(H)
    xo,Do=ldlt_old(H,g); xn,Dn=ldlt_new(H,g); xr=np.linalg.solve(H,g); xq=O.fullpiv_qr_solve_6(H,g)
    print(k,'ok',ok,'iters',o.stats.iterations,'nvalid',o.stats.n_valid,'cond %.2e'%c,'dmin/dmax %.2e'%(Do.min()/Do.max()),'|x| %.2e'%np.abs(xr).max(),
          'old-qr %.1e new-qr %.1e np-qr %.1e'%(np.abs(xo-xq).max(),np.abs(xn-xq).max(),np.abs(xr-xq).max()))
    Tprev=T
print('--- frame 0, iteration by iteration')
f=sc["frames"][0]
guess=f["absolute_guess"] if "absolute_guess" in f else np.eye(4)@f["guess_step"]
for it in range(1,5):
    y2=dict(y,optimization_iter_num=it)
    o2=util.oracle_for(mode,y2,loc)
    if sc.get('ivox_capacity') is not None: o2.set_ivox_capacity(sc['ivox_capacity'])
    o2.AddCloudToLocalMap(*sc["init_clouds"])
    ok,T=o2.Match(f["scan"],np.array(guess,dtype=np.float64),src1=f["corner"],update_map=False)
    H,g=o2.last_system()
    xo,Do=ldlt_old(H,g); xn,Dn=ldlt_new(H,g); xq=O.fullpiv_qr_solve_6(H,g)
    print(it,'nvalid',o2.stats.n_valid,'cond %.2e'%np.linalg.cond(H),'dmin/dmax %.2e'%(Do.min()/Do.max()),'D>0',bool((Do>0).all()),'|x| %.2e'%np.abs(xq).max(),'old-qr %.1e new-qr %.1e'%(np.abs(xo-xq).max(),np.abs(xn-xq).max()))
    o2.close()
