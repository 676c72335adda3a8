import sys, numpy as np
sys.path.insert(0, "/root/repo")
from sparse_gslam_amd import capi, synth
from oracle import c_oracle, np_oracle
rng = np.random.default_rng(0)
# replay the generator of scripts/stress.py up to case 25
for case in range(26):
    V = int(rng.choice([60, 300, 450, 1200, 3000, 8000, 20000]))
    dens = float(rng.choice([1.0, 1.05, 1.5, 3.0, 6.0]))
    E = max(V - 1, int(dens * V))
    kw = dict(V=V, E=E, seed=int(rng.integers(1, 10**6)), p_random=float(rng.choice([0.0, 0.0, 0.05, 0.3])),
              info_mode=str(rng.choice(["diag", "full"])), phi=float(rng.choice([1.0, 10.0])),
              init=str(rng.choice(["incremental", "incremental", "odom"])) if V <= 1200 else "incremental")
    g = synth.manhattan(**kw)
    if rng.random() < 0.3:
        g.fixed[rng.integers(0, V, 3)] = True
print(kw, g.fixed.sum())
res = {}
for name, rows in (("direct", 1 << 20), ("amg", 0)):
    with capi.Optimizer(0, direct_rows=rows) as o:
        o.set_graph(*g.arrays()); print(o.solver_description()[:80])
        done, st = o.optimize(6); res[name] = st["chi2"]
oP, ost = c_oracle.gauss_newton(*g.arrays(), iters=6)
nP, nst = np_oracle.gauss_newton(*g.arrays(), iters=6)
for k in range(7):
    print(k, "oracleC %.12e" % ost["chi2"][k], "oracleNP %.12e" % nst["chi2"][k], "direct %.12e" % res["direct"][k], "amg %.12e" % res["amg"][k])
