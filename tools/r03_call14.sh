#!/bin/bash
# Round 3, call 14: LZ4 / Snappy encoders after wave_count() returns a scalar (uniform positions for the compiler) -- corpus, wordmix, fragments
export TMPDIR=/tmp
O=gpurun_out/r03c15
rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-sweep --no-extra --blocks 65536 --steps 5 --warmup 2"
for w in lz4_compress snappy_compress; do
  for d in corpus wordmix fragments; do
    echo "## $w $d" >> $O/enc.txt
    timeout 200 $B --workload $w --data $d 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/enc.txt 2>&1
  done
done
cat $O/enc.txt
timeout 600 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_hadoop.py tests/test_gpu_snappy_framed.py tests/test_gpu_lz4_frame.py -m gpu -x -q 2>&1 | tail -2
