# A/B of the folded multigrid cycle against SGO_AMG_FOLD=0 on the bench configs (gpurun -- 'CONFIGS="C4 C2" bash scripts/fold_ab.sh').
set -e
cd $GRAFT_REPO_ROOT
pr() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], round(d['value']/1e6,1),'M/s', round(d['gn_iter_ms_median'],3),'ms/GN', d['pcg_iters_per_gn_iter'],'its', d['final_chi2_rel_err_vs_oracle'].get('value'), 'set_graph', round(d['set_graph_ms'],1))
" $1 "$2"; }
for c in ${CONFIGS:-C4 C2 C3s}; do
  timeout -k 10 300 python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/fold_$c.json; pr gpurun_out/fold_$c.json "$c fold"
  SGO_AMG_FOLD=0 timeout -k 10 300 python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/nofold_$c.json; pr gpurun_out/nofold_$c.json "$c nofold"
done
