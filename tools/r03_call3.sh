#!/bin/bash
# Round 3, GPU call 3: the PHASED ring decoders (one memory phase per sequence, achip_rings.h) against round 2's rings (ring class 2), same build.
export TMPDIR=/tmp
O=gpurun_out/r03c3
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-sweep --no-extra"
line() { grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_avg'], r['config']['decoder'][:20])"; }
timeout 500 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_corpus.py tests/test_gpu_fuzz.py tests/test_gpu_snappy_framed.py tests/test_gpu_hadoop.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for w in lz4_decompress snappy_decompress; do
  for rc in 0 2 3 4 5; do
    for d in fragments; do
      echo "## $w ring_class $rc $d" >> $O/ab.txt
      timeout 150 $B --workload $w --data $d --variant 1 --ring-class $rc --steps 5 --warmup 2 2>&1 | line >> $O/ab.txt 2>&1
    done
  done
done
cat $O/ab.txt
for w in lz4_decompress snappy_decompress; do
  set="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM"
  timeout 200 bash tools/pmc.sh c3_$w "$set" --workload $w --steps 3 --warmup 1 > /dev/null 2>&1
  echo "## $w" >> $O/issue_profile.txt; grep "rings_kernel" gpurun_out/pmc_c3_$w.txt | grep -v "true, " >> $O/issue_profile.txt; rm -f gpurun_out/pmc_c3_$w.txt
done
cat $O/issue_profile.txt
