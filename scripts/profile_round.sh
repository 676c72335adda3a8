#!/bin/bash
# Regenerates the rocprofv3 artifacts under profiles/ on the GPU box (through gpurun, from the repo root:
#   gpurun -- bash scripts/profile_round.sh r04 C4      or  ... r04 C5 ); outputs land in gpurun_out/.
# rocprofv3 wants cwd and TMPDIR under /tmp; --pmc passes are separate from the --stats pass.
# Per config three passes + the bench lines:
#   plain  --kernel-trace --stats, plain stream launches with the stop flag read after EVERY PCG iteration
#          (SGO_USE_GRAPH=0 SGO_PCG_CHUNK=1): no early-exit launch past convergence, every dispatch is a working one
#   graph  --kernel-trace of the default (hipGraph replay) run: the kernels back to back inside a solve
#   pmc    --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass
# The per-dispatch traces are large and deleted; scripts/trace_summary.py keeps per-kernel median / p10 / p90 of the
# working dispatches first, which is what bench.py's roofline (isolated and in_solve) is recomputed from.
set -u
tag=${1:-r04}
cfg=${2:-C4}
lc=$(echo $cfg | tr 'A-Z' 'a-z')
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out
B="--config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
SGO_USE_GRAPH=0 SGO_PCG_CHUNK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plain_$lc -- python3 $R/bench.py $B > $O/prof_plain_$lc.log 2>&1
python3 $R/scripts/trace_summary.py $O/prof_plain_$lc $O/${tag}_trace_summary_${lc}_plain.json $O/${tag}_trace_summary_${lc}_plain.csv
cp $(find $O/prof_plain_$lc -name "*kernel_stats.csv" | head -1) $O/${tag}_amg_${lc}_kernel_stats.csv
find $O/prof_plain_$lc -name "*kernel_trace.csv" -delete
echo "[profile_round] plain pass done"
rocprofv3 --kernel-trace --output-format csv -d $O/prof_graph_$lc -- python3 $R/bench.py $B > $O/prof_graph_$lc.log 2>&1
python3 $R/scripts/trace_summary.py $O/prof_graph_$lc $O/${tag}_trace_summary_${lc}_graph.json $O/${tag}_trace_summary_${lc}_graph.csv
find $O/prof_graph_$lc -name "*kernel_trace.csv" -delete
echo "[profile_round] graph pass done"
for c in FETCH_SIZE WRITE_SIZE; do
  SGO_USE_GRAPH=0 SGO_PCG_CHUNK=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$lc -- python3 $R/bench.py --config $cfg --steps 1 --warmup 0 --iters 3 --no-cpu-baseline --no-roofline > $O/pmc_${c}_$lc.log 2>&1
  echo "[profile_round] pmc $c done"
done
python3 $R/scripts/pmc_summary.py $O/pmc_FETCH_SIZE_$lc $O/pmc_WRITE_SIZE_$lc > $O/${tag}_pmc_traffic_amg_$lc.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
# bench lines (the roofline leg reads the PMC and in-solve summaries: make them visible under profiles/ first)
cp $O/${tag}_pmc_traffic_amg_$lc.json $O/${tag}_trace_summary_${lc}_graph.json $O/${tag}_trace_summary_${lc}_plain.json $R/profiles/
cd $R
if [ "$cfg" = "C4" ]; then
  python3 bench.py 2>/dev/null | tail -1 > $O/${tag}_bench_c4.json
  python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${tag}_bench_c4_steps20.json
  python3 bench.py --config C2 2>/dev/null | tail -1 > $O/${tag}_bench_c2.json
  python3 bench.py --config C3s 2>/dev/null | tail -1 > $O/${tag}_bench_c3s.json
  list="c4 c4_steps20 c2 c3s"
else
  python3 bench.py --config $cfg --no-cpu-baseline 2> $O/${tag}_bench_$lc.err | tail -1 > $O/${tag}_bench_$lc.json
  list="$lc"
fi
for c in $list; do python3 - $O/${tag}_bench_$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], round(d["value"] / 1e6, 1), "M/s;", round(d["gn_iter_ms_median"], 3), "ms per GN iteration;", d["pcg_iters_per_gn_iter"],
      "PCG its; set_graph", round(d["set_graph_ms"], 1), "ms;", r.get("kernel"), "frac", r.get("frac"), "in_solve", (r.get("in_solve") or {}).get("frac"),
      "traffic", r.get("traffic"), "sharded", r.get("row_owner_sharded_frac"), "odom", d.get("value_init_odom"))
PY
done
