#!/bin/bash
# Round 3, call 20: per-kernel times of the two-pass decoders on the corpus batch (LZ4 and Snappy)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c20
for w in lz4_decompress snappy_decompress; do
  timeout 300 bash tools/kstats.sh c20_$w --workload $w --data corpus --steps 5 --warmup 2
  cp gpurun_out/kstats_c20_$w.txt gpurun_out/r03c20/
done
cat gpurun_out/r03c20/*.txt
