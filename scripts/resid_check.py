import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
from oracle import c_oracle
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
g = synth.manhattan(V, V - 1 + 45, seed=1, info_mode="full", init="incremental", phi=10.0)
free = ~g.fixed
with capi.Optimizer(0, direct_rows=0, pcg_tol=1e-8) as o:
    o.set_graph(*g.arrays())
    b, _, _, _ = o.linearize()
    xp, it, rr = o.solve()
    def resid(x): return np.linalg.norm(b - o.hessian_apply(x)) / np.linalg.norm(b)
    print("pcg: iters", it, "true relres", resid(xp))
    with capi.Optimizer(0) as d:
        d.set_graph(*g.arrays()); print(d.solver_description()[:60])
        d.optimize(1); Pd = d.get_poses()
    xd = Pd[free] - g.poses[free]; xd[:, 2] = (xd[:, 2] + np.pi) % (2 * np.pi) - np.pi
    print("direct: true relres", resid(xd))
    oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=1)
    xo = oP[free] - g.poses[free]; xo[:, 2] = (xo[:, 2] + np.pi) % (2 * np.pi) - np.pi
    print("oracle: true relres", resid(xo))
    nrm = np.abs(xo).max()
    print("max |xd-xo|", np.abs(xd - xo).max(), "max |xp-xo|", np.abs(xp - xo).max(), "max |xd-xp|", np.abs(xd - xp).max(), "|x|max", nrm)
    # energy-norm differences
    def en(x): return float(np.sum(x * o.hessian_apply(x)))
    print("energy: ||xd-xo||_H^2", en(xd - xo), "||xp-xo||_H^2", en(xp - xo), "||xo||_H^2", en(xo))
