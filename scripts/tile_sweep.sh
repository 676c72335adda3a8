for cfg in "78000 512" "52000 768" "39000 1024" "52000 700" "60000 640"; do
  set -- $cfg
  echo "== LDS $1 DIV $2"
  SGO_TILE_LDS=$1 SGO_TILE_DIV=$2 SGO_VERBOSE=1 python scripts/spmv0_probe.py C4 2>&1 | grep "tiles:\|variant"
done
