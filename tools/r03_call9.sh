#!/bin/bash
# Round 3, GPU call 9: the Snappy "many matches per window" encoder (variant 4, two tiers) against variant 3 / 2.
export TMPDIR=/tmp
O=gpurun_out/r03c9
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-sweep --no-extra"
timeout 300 python -m pytest tests/test_gpu_lz4_snappy.py -m gpu -x -q -k "compress" > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for d in corpus wordmix fragments; do
  for v in 2 3 4; do
    echo "## snappy_compress $d variant $v" >> $O/enc.txt
    timeout 250 $B --workload snappy_compress --data $d --blocks 65536 --steps 3 --warmup 1 --compress-variant $v 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/enc.txt 2>&1
  done
done
echo "## snappy_compress corpus variant 4, 262144 blocks" >> $O/enc.txt
timeout 250 $B --workload snappy_compress --data corpus --steps 2 --warmup 1 --compress-variant 4 2>&1 | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['kernel_ms_avg'])" >> $O/enc.txt 2>&1
cat $O/enc.txt
# instruction / wait profile of the two mw kernels on corpus
for w in lz4_compress snappy_compress; do
  timeout 250 bash tools/pmc.sh c9_$w "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH" --workload $w --data corpus --blocks 65536 --steps 2 --warmup 1 --compress-variant 4 > /dev/null 2>&1
  echo "## $w" >> $O/profile.txt; cat gpurun_out/pmc_c9_$w.txt >> $O/profile.txt; rm -f gpurun_out/pmc_c9_$w.txt
done
cat $O/profile.txt
