for cfg in "78000 512 512" "150000 256 1024" "150000 256 512" "110000 384 1024" "110000 384 512"; do
  set -- $cfg
  echo "== LDS $1 DIV $2 THREADS $3"
  SGO_TILE_LDS=$1 SGO_TILE_DIV=$2 SGO_TILE_THREADS=$3 SGO_VERBOSE=1 python scripts/spmv0_probe.py C4 2>&1 | grep "tiles:\|variant"
done
