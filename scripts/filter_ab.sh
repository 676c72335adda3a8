#!/bin/bash
# A/B of the filtered smoothing (SGO_AMG_FILTER) on the dead-reckoned start: scripts/filter_ab.sh [config] [iters]
cfg=${1:-C4}; it=${2:-6}
for f in 0 1; do
  echo "== SGO_AMG_FILTER=$f $cfg init=odom"
  SGO_AMG_FILTER=$f python scripts/odom_probe.py $cfg $it 2>&1 | cut -c1-900
done
