#!/bin/bash
# Round 3, call 26: run-to-run spread of the headline line on one box (five runs of `bench.py --no-extra --no-cpu-baseline`, LZ4 and Snappy)
export TMPDIR=/tmp
O=gpurun_out/r03c26
rm -rf $O; mkdir -p $O
for wl in lz4_decompress snappy_decompress; do
  for i in 1 2 3 4 5; do
    timeout 200 python bench.py --no-cpu-baseline --no-extra --no-sweep --workload $wl 2>&1 | grep '^{' | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$wl', r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_avg'], r['roofline']['traffic'])" | tee -a $O/spread.txt
  done
done
