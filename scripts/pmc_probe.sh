#!/bin/bash
# PMC passes over scripts/spmv0_modes.py (a few counters per pass); prints per-kernel means.
# usage: gpurun -- bash scripts/pmc_probe.sh "CNT1 CNT2" "CNT3 ..." ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
export SGO_PROBE_REPS=10 SGO_PROBE_VARIANTS=${SGO_PROBE_VARIANTS:-0,16}
i=0
for set in "$@"; do
  i=$((i+1))
  O=$R/gpurun_out/pmcp_$i
  rm -rf $O
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -- python3 $R/scripts/spmv0_modes.py C4 > $O.log 2>&1
  python3 - "$O" <<'PY'
import sys, glob, csv, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void ", "").replace("sgo::(anonymous namespace)::", ""))
        if "spmv0" in n:
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
for n in sorted(acc):
    print(n, {c: round(sum(v) / len(v), 1) for c, v in acc[n].items()}, "launches", len(next(iter(acc[n].values()))))
PY
  find $O -name "*.csv" -delete
done
