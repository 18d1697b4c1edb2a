#!/bin/bash
# Round 3, GPU call 7: the LZ4 "many matches per window" encoder (variant 4) against the batch-probe encoder (variant 1); two LZ4 ring placements.
export TMPDIR=/tmp
O=gpurun_out/r03c7
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-sweep --no-extra"
echo skip-tests
for d in corpus wordmix fragments; do
  for v in 1 4; do
    echo "## lz4_compress $d variant $v" >> $O/enc.txt
    timeout 200 $B --workload lz4_compress --data $d --blocks 65536 --steps 3 --warmup 1 --compress-variant $v 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/enc.txt 2>&1
  done
done
echo "## lz4_compress corpus variant 4, 262144 blocks" >> $O/enc.txt
timeout 200 $B --workload lz4_compress --data corpus --steps 3 --warmup 1 --compress-variant 4 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/enc.txt 2>&1
cat $O/enc.txt
