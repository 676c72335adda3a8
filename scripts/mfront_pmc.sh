#!/bin/bash
# HBM traffic of the multifrontal path's kernels (rocprofv3 --pmc, one counter per pass): scripts/mfront_pmc.sh <config> <tag>
cfg=${1:-C3s}; tag=${2:-r04_mfront}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/mfpmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/mfpmc_$c -- python3 $R/scripts/mfront_only.py $cfg 20 1 > $O/mfpmc_$c.log 2>&1
done
cd $R
python3 scripts/pmc_summary.py $O/mfpmc_FETCH_SIZE $O/mfpmc_WRITE_SIZE > $O/${tag}_pmc_traffic.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python3 -c "
import json
d=json.load(open('$O/${tag}_pmc_traffic.json'))['kernels']
for k,v in d.items():
    if k.startswith('k_mf'): print(k, json.dumps(v))
"
