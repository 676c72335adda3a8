#!/bin/bash
# dead-reckoned start: optimize(20) twice in one process (the second pass at busy-chip clocks) under environment variants
cfg=${1:-C4}
run() { echo "== $*"; env "$@" python scripts/odom_probe.py $cfg 20 2 2>&1 | grep "^pcg \[\|^done" | tail -2 | sed 's/relres.*ms \[/ms [/' | cut -c1-700; }
run SGO_AMG_FILTER=0
run SGO_AMG_FILTER=1
run SGO_AMG_FILTER=1 SGO_AMG_NU=1
run SGO_AMG_FILTER=1 SGO_AMG_THETA=0.05
run SGO_AMG_FILTER=1 SGO_AMG_THETA=0.1
run SGO_AMG_FILTER=1 SGO_AMG_KDEPTH=10
