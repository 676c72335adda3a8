for cfg in "64000 512 512" "64000 512 1024" "40000 1024 512" "100000 256 512" "100000 256 1024" "150000 256 1024" "150000 128 1024" "30000 2048 512"; do
  set -- $cfg
  echo "== LDS $1 DIV $2 THREADS $3"
  SGO_TILE_LDS=$1 SGO_TILE_DIV=$2 SGO_TILE_THREADS=$3 SGO_VERBOSE=1 python scripts/spmv0_probe.py C4 2>&1 | grep "tiles:\|variant"
done
