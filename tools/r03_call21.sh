#!/bin/bash
# Round 3, call 21: differential fuzz of every DECODER variant against the oracle (status, error offset, plaintext): the ring decoders changed this
# round (rings topped up per sequence, Snappy's literal-then-copy trips), the readers' auto choice, the Hadoop readers
export TMPDIR=/tmp
O=gpurun_out/r03c21
rm -rf $O; mkdir -p $O
for seed in 41 42; do
  timeout 1500 python tools/fuzz_decoders.py 20000 $seed 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a $O/fuzz_decoders.txt | tail -20
done
timeout 1500 python tools/fuzz_decoders.py 10000 43 lz4frame,snappyframed 2>&1 | grep -v "^$\|amdgpu.ids" | tee -a $O/fuzz_decoders.txt | tail -6
grep "TOTAL" $O/fuzz_decoders.txt
