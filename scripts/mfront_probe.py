"""The multifrontal path against the multigrid PCG on the same graph (one GPU): chi2 per iteration, final poses, time.

    python scripts/mfront_probe.py [C3s] [iters=20] [reps=5]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402


def run(g, iters, reps, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        opt = capi.Optimizer(0)
        t0 = time.perf_counter()
        opt.set_graph(*g.arrays())
        t_set = time.perf_counter() - t0
        desc = opt.solver_description()
        times = []
        st = None
        poses = None
        for r in range(reps):
            opt.set_poses(g.poses)
            t0 = time.perf_counter()
            rc, st = opt.optimize(iters)
            times.append(time.perf_counter() - t0)
            poses = opt.get_poses()
        return dict(desc=desc, t_set=t_set, times=times, rc=rc, st=st, poses=poses)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3s"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    g = synth.config(name)
    a = run(g, iters, reps, {"SGO_MFRONT": "1"})
    b = run(g, iters, reps, {"SGO_MFRONT": "0"})
    print("A:", a["desc"])
    print("B:", b["desc"])
    print(f"set_graph: A {1e3 * a['t_set']:.1f} ms   B {1e3 * b['t_set']:.1f} ms")
    print("optimize ms: A", " ".join(f"{1e3 * t:.2f}" for t in a["times"]), "  B", " ".join(f"{1e3 * t:.2f}" for t in b["times"]))
    print(f"per GN iteration: A {1e3 * min(a['times']) / iters:.3f} ms   B {1e3 * min(b['times']) / iters:.3f} ms")
    print("rc:", a["rc"], b["rc"], "iters_done:", a["st"]["iters_done"], b["st"]["iters_done"])
    ca, cb = np.array(a["st"]["chi2"]), np.array(b["st"]["chi2"])
    k = min(len(ca), len(cb))
    rel = np.abs(ca[:k] - cb[:k]) / np.maximum(np.abs(cb[:k]), 1e-300)
    print("chi2 A:", " ".join(f"{v:.6e}" for v in ca[:6]), "...", f"{ca[-1]:.9e}")
    print("chi2 B:", " ".join(f"{v:.6e}" for v in cb[:6]), "...", f"{cb[-1]:.9e}")
    print(f"max rel chi2 diff over iterates: {rel.max():.3e}")
    print(f"max pose diff: {np.abs(a['poses'] - b['poses']).max():.3e}")
    print("A seconds[0..3]:", a["st"]["seconds"][:4])


if __name__ == "__main__":
    main()
