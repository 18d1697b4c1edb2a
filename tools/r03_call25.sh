#!/bin/bash
# Round 3, call 25: the ring decoders' LDS slot padding re-swept with this round's 256 + 256-byte rings (it was chosen for 128 + 256), headline config, same box
export TMPDIR=/tmp
O=gpurun_out/r03c25
rm -rf $O; mkdir -p $O
for wl in lz4_decompress snappy_decompress; do
  for pad in 80 16 48 64 96 112 144 80; do
    v=$(timeout 200 python bench.py --no-cpu-baseline --no-extra --no-sweep --steps 5 --warmup 2 --workload $wl --ring-pad $pad 2>&1 | grep '^{' | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'])")
    echo "$wl pad=$pad $v" | tee -a $O/pad.txt
  done
done
