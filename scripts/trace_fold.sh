# rocprofv3 kernel timeline of the replayed PCG hipGraph near the end of a solve (CFG=C2 WIN=330 for a window further back):
# start, duration and gap of every kernel -- what a node of the graph costs back to back.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/trace_fold
rocprofv3 --kernel-trace --output-format csv -d $O/trace_fold -- python3 $R/bench.py --config ${CFG:-C4} --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/trace_fold.log 2>&1
f=$(find $O/trace_fold -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 400 kernels: print a window in the middle of the last optimize
names = [r["Kernel_Name"] for r in rows]
# find the last k_pose_update, take the 60 kernels before it
idx = max(i for i, n in enumerate(names) if "k_pose_update" in n)
W = int(__import__("os").environ.get("WIN", "70")); t0 = int(rows[idx - W]["Start_Timestamp"])
prev_end = None
for r in rows[idx - W: idx - W + 71]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("sgo::(anonymous namespace)::", "").replace("void ", "")[:44]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {n}")
    prev_end = e
PY
find $O/trace_fold -name "*.csv" -delete
