"""optimize(iters) x reps on one config through whichever path sgo_set_graph_se2 picks (profiling target)."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sparse_gslam_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3s"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = synth.config(name)
opt = capi.Optimizer(0)
opt.set_graph(*g.arrays())
print(opt.solver_description())
for r in range(reps):
    opt.set_poses(g.poses)
    t0 = time.perf_counter()
    rc, st = opt.optimize(iters)
    print(f"rep {r}: {1e3 * (time.perf_counter() - t0):.2f} ms, rc {rc}, chi2 {st['chi2'][-1]:.9e}")
