"""C4 against its golden for a few PCG tolerances / tolerance caps: iteration counts, time, chi2 error per iterate (round 3).
python scripts/tol_sweep.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_gslam_amd import capi, synth
f=np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'C4_direct.npz'))
g=synth.config("C4")
for tol in (1e-8,1e-7,1e-6,1e-5):
    with capi.Optimizer(0,pcg_tol=tol) as o:
        o.set_graph(*g.arrays()); done,st=o.optimize(20); P=o.get_poses()
    err=max(abs(a-b)/b for a,b in zip(st['chi2'],f['chi2']))
    rerr=max(abs(a-b)/b for a,b in zip(st['robust_chi2'],f['robust_chi2']))
    print("tol %g: pcg avg %.1f  gn_ms %.2f  max rel chi2 err over iterations %.2e (final %.2e) robust %.2e  pose err %.2e"%(tol,np.mean(st['pcg_iters']),1e3*np.median(st['seconds']),err,abs(st['chi2'][-1]-f['chi2'][-1])/f['chi2'][-1],rerr,np.abs(P[::50]-f['poses_stride50']).max()))
