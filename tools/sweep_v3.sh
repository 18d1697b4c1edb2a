#!/bin/bash
# lane-per-block (variant 2) vs lane-group-per-block (variant 1) LZ4 decoders, same box, same run
for data in fragments wordmix corpus; do
  for cfg in "1 0" "2 0" "2 1"; do
    set -- $cfg
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --blocks 131072 --workload lz4_decompress --data $data --variant $1 --ring-class $2 2>&1 | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])" 2>&1 | tail -1)
    echo "lz4_decompress $data variant=$1 ring_class=$2 $v"
  done
done
