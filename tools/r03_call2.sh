#!/bin/bash
# Round 3, GPU call 2: the suite on the cleaned-up library (variants deleted, readers' auto choice, Snappy ring decoder with literal-then-copy
# trips, LZ4 Hadoop reader's negative lengths), the headline pair and the containers re-measured, and an instruction-issue profile of the two
# ring kernels (what bounds them: VALU issue, scalar issue, LDS, or waiting).  Everything lands in gpurun_out/r03c2/.
export TMPDIR=/tmp
O=gpurun_out/r03c2
mkdir -p $O
T0=$(date +%s)
stamp() { echo "== $1 at +$(( $(date +%s) - T0 )) s" | tee -a $O/timeline.txt; }
B="python bench.py --no-cpu-baseline --no-sweep"
line() { grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['kernel_ms_avg'], r['config']['decoder'][:30])"; }

stamp "gpu tests"
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log | tee -a $O/timeline.txt

stamp "headline pair (fragments): lz4, snappy; corpus: snappy rings forced vs auto"
for w in lz4_decompress snappy_decompress; do
  echo "## $w fragments" >> $O/headline.txt
  timeout 150 $B --no-extra --workload $w 2>&1 | line >> $O/headline.txt 2>&1
done
echo "## snappy corpus auto" >> $O/headline.txt
timeout 150 $B --no-extra --workload snappy_decompress --data corpus --steps 5 --warmup 2 2>&1 | line >> $O/headline.txt 2>&1
echo "## snappy corpus rings" >> $O/headline.txt
timeout 150 $B --no-extra --workload snappy_decompress --data corpus --steps 5 --warmup 2 --variant 1 2>&1 | line >> $O/headline.txt 2>&1
echo "## snappy wordmix rings" >> $O/headline.txt
timeout 150 $B --no-extra --workload snappy_decompress --data wordmix --steps 5 --warmup 2 --variant 1 2>&1 | line >> $O/headline.txt 2>&1
cat $O/headline.txt | tee -a $O/timeline.txt

stamp "containers with the readers' auto choice"
timeout 200 $B --section lz4frame > $O/containers_auto.json 2> $O/containers_auto.err

stamp "counters available"
rocprofv3 --list-avail > $O/counters_avail.txt 2>&1
grep -c . $O/counters_avail.txt | tee -a $O/timeline.txt

stamp "issue profile of the ring kernels"
for w in lz4_decompress snappy_decompress; do
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
    tag=${w}_$(echo $set | md5sum | cut -c1-6)
    timeout 200 bash tools/pmc.sh $tag "$set" --workload $w --steps 3 --warmup 1 > /dev/null 2>&1
    echo "## $w : $set" >> $O/issue_profile.txt
    grep "rings_kernel" gpurun_out/pmc_$tag.txt >> $O/issue_profile.txt 2>/dev/null
    rm -f gpurun_out/pmc_$tag.txt
  done
done
stamp "done"
