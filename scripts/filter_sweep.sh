#!/bin/bash
# sweep of the filtered smoothing's threshold (SGO_AMG_THETA_FILTER) over the shapes it matters for
for tf in 1e-4 1e-3 1e-2; do
  for cfg in "C4 init=odom" "C4r" "C2 init=odom"; do
    echo "== theta_filter=$tf $cfg"
    SGO_AMG_THETA_FILTER=$tf python scripts/cfg_probe.py ${cfg%% *} 20 2 $( [[ "$cfg" == *" "* ]] && echo ${cfg#* } ) 2>&1 | tail -3 | cut -c1-420
  done
done
