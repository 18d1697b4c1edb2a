#!/bin/bash
# A/B of the LDS slot padding of the ring decoders (same box, same run)
for wl in lz4_decompress snappy_decompress; do
  for data in fragments wordmix; do
    for pad in 0 16 48 0 16; do
      v=$(timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 --blocks 131072 --workload $wl --data $data --ring-pad $pad 2>&1 | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
      echo "$wl $data pad=$pad $v"
    done
  done
done
