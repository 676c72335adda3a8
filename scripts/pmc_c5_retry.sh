#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE) of the C5 bench on their own, for when scripts/profile_round.sh lost one of them to a profiler crash.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_${c}_c5
  SGO_USE_GRAPH=0 SGO_PCG_CHUNK=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_c5 -- python3 $R/bench.py --config C5 --steps 1 --warmup 0 --iters 3 --no-cpu-baseline --no-roofline > $O/pmc_${c}_c5.log 2>&1
  echo "pmc $c rc=$?"
done
python3 $R/scripts/pmc_summary.py $O/pmc_FETCH_SIZE_c5 $O/pmc_WRITE_SIZE_c5 > $O/r05_pmc_traffic_amg_c5.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cp $O/r05_pmc_traffic_amg_c5.json $R/profiles/
cd $R
python3 bench.py --config C5 --no-cpu-baseline 2> $O/r05_bench_c5.err | tail -1 > $O/r05_bench_c5.json
python3 -c "
import json; d=json.load(open('$O/r05_bench_c5.json')); r=d['roofline']; print(d['value'], r['kernel'], r['frac'], r['traffic'], r['traffic_over_algorithmic'] if 'traffic_over_algorithmic' in r else None, r['level0_product']['frac'], r['level0_product']['traffic_over_algorithmic'])"
