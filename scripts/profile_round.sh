#!/bin/bash
# Regenerates the rocprofv3 artifacts under profiles/ on the GPU box (run through gpurun from the
# repo root: `gpurun -- bash scripts/profile_round.sh r01`); outputs land in gpurun_out/.
# rocprofv3 wants cwd and TMPDIR under /tmp; --pmc passes are separate from the --stats pass.
set -u
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
# Plain stream launches (no hipGraph) for the profiled passes, as in bench.py's own roofline leg: graph
# replay keeps up to two 4-iteration chunks in flight past convergence, whose early-exit launches
# (0.6 us each) would pull the per-kernel averages of the summary 10-20 % below the working launches'.
export SGO_USE_GRAPH=0
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_amg_c4 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_amg_c4.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --iters 4 --no-cpu-baseline --no-roofline > $O/pmc_$c.log 2>&1
done
python3 $R/scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/${tag}_pmc_traffic_amg_c4.json
cp $(find $O/prof_amg_c4 -name "*kernel_stats.csv" | head -1) $O/${tag}_amg_c4_kernel_stats.csv
# per-dispatch traces are large: keep only the summaries
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
# bench lines (the roofline leg reads the PMC summary: make it visible under profiles/ first)
unset SGO_USE_GRAPH
cp $O/${tag}_pmc_traffic_amg_c4.json $R/profiles/${tag}_pmc_traffic_amg_c4.json
cd $R
python3 bench.py 2>/dev/null | tail -1 > $O/${tag}_bench_c4.json
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${tag}_bench_c4_steps20.json
python3 bench.py --config C2 2>/dev/null | tail -1 > $O/${tag}_bench_c2.json
python3 bench.py --config C3s 2>/dev/null | tail -1 > $O/${tag}_bench_c3s.json
for c in c4 c4_steps20 c2 c3s; do python3 - $O/${tag}_bench_$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s;", round(d["gn_iter_ms_median"], 3), "ms per GN iteration;", d["pcg_iters_per_gn_iter"],
      "PCG its; set_graph", round(d["set_graph_ms"], 1), "ms;", r.get("kernel"), "frac", r.get("frac"), "traffic", r.get("traffic"),
      "sharded", r.get("row_owner_sharded_frac"), "odom", d.get("value_init_odom"))
PY
done
python3 scripts/robustness.py > $O/${tag}_robustness_raw.txt 2>/dev/null; cat $O/${tag}_robustness_raw.txt
