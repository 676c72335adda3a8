#!/bin/bash
# kernel table of the multifrontal path: scripts/mfront_prof.sh <config> <tag>
cfg=${1:-C3s}; tag=${2:-mf}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
SGO_MFRONT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 scripts/mfront_only.py $cfg 20 3 > $out/run.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' > gpurun_out/${tag}_kernel_stats.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f} {r['Percentage']}%")
PY
cat gpurun_out/${tag}_kernel_stats.txt
tail -5 $out/run.log
