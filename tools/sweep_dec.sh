#!/bin/bash
# default decoders over the three data kinds (LZ4 and Snappy), 131072 blocks
for wl in lz4_decompress snappy_decompress; do
  for data in fragments wordmix corpus; do
    v=$(timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --blocks 131072 --workload $wl --data $data $* 2>&1 | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])" 2>&1 | tail -1)
    echo "$wl $data $* $v"
  done
done
