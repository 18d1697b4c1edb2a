#!/bin/bash
# Round 3, call 24: the Snappy encoder kernels capped at 96 VGPRs (five wavefronts per SIMD: what their LDS tables allow; they sat at 97 -> four)
export TMPDIR=/tmp
O=gpurun_out/r03c24
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_hadoop.py tests/test_gpu_snappy_framed.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-sweep --no-extra --blocks 65536 --steps 5 --warmup 2"
for d in corpus wordmix fragments; do
  echo "## snappy_compress $d" >> $O/enc.txt
  timeout 200 $B --workload snappy_compress --data $d 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/enc.txt 2>&1
done
timeout 300 python bench.py --section lz4frame --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read())
for k,v in r.items():
    if 'snappy' in k: print(k, v['compress_GiBps'], v['decompress_GiBps'])" >> $O/enc.txt
cat $O/enc.txt
