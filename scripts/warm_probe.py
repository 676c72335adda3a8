import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
g = synth.config(name)
free = ~g.fixed
with capi.Optimizer(0) as o:
    o.set_graph(*g.arrays())
    P_prev = g.poses.copy()
    x_prev = None
    hist = []
    for k in range(12):
        if x_prev is not None:
            b, _, _, _ = o.linearize()
            q = o.hessian_apply(x_prev)
            gam = float((b * x_prev).sum() / (x_prev * q).sum())
            red = np.linalg.norm(b - gam * q) / np.linalg.norm(b)
            line = f"k={k}: gamma {gam:.3f}  ||b - gamma H x_prev|| / ||b|| = {red:.3f}"
            for m in (2, 3):
                if len(hist) >= m:
                    X = np.stack([h.ravel() for h in hist[-m:]], 1)
                    Q = np.stack([o.hessian_apply(h).ravel() for h in hist[-m:]], 1)
                    # Galerkin (energy-optimal) combination: (X^T H X) c = X^T b
                    c = np.linalg.solve(X.T @ Q, X.T @ b.ravel())
                    line += f" | {m} steps: {np.linalg.norm(b.ravel() - Q @ c) / np.linalg.norm(b):.3f}"
            print(line, flush=True)
        done, st = o.optimize(1)
        P = o.get_poses()
        d = P[free] - P_prev[free]; d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
        # hessian order = ascending free id == order of P[free]
        x_prev = d
        hist.append(d)
        P_prev = P
