import os
import sys
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_oracle
from sparse_gslam_amd import synth
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fsa_lib.py')).read())
V=int(sys.argv[1]); E=int(sys.argv[2])
g=synth.manhattan(V,E,seed=4,init="odom")
free=np.flatnonzero(~g.fixed)
poses=g.poses.copy()
H0,b0,_,_=np_oracle.linearize(poses,g.fixed,g.ei,g.ej,g.meas,g.info,g.phi)
hier0=build(H0,poses[free,:2])
it0,x0=solve(hier0,b0)
print("GN0: fresh hierarchy:",it0,"iterations")
for step in range(1,4):
    poses=np_oracle.oplus(poses,g.fixed,x0)
    H1,b1,_,rc=np_oracle.linearize(poses,g.fixed,g.ei,g.ej,g.meas,g.info,g.phi)
    pos1=poses[free,:2]
    fresh=build(H1,pos1)
    itf,xf=solve(fresh,b1)
    stale=build(H1,pos1,reuse=hier0,dynamic_mask=False)
    its,_=solve(stale,b1)
    dyn=build(H1,pos1,reuse=hier0,dynamic_mask=True)
    itd,_=solve(dyn,b1)
    anded=build(H1,pos1,reuse=hier0,dynamic_mask="and")
    ita,_=solve(anded,b1)
    print(f"   stale aggregates + (stale AND fresh) mask {ita}")
    nl=build(H1,pos1,reuse=hier0,dynamic_mask="nolump")
    itn,_=solve(nl,b1)
    print(f"   stale aggregates + stale mask, newly strong connections not lumped {itn}")
    print(f"GN{step}: robust chi2 {rc:.4g}: fresh {itf}; stale aggregates + stale mask {its}; stale aggregates + fresh mask {itd}",flush=True)
    hier0=fresh; x0=xf
