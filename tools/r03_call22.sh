#!/bin/bash
# Round 3, call 22: the record executor's LDS window with aligned accesses only (achip_seqexec2.h Window): tests, corpus decode of LZ4 / Snappy with
# per-kernel times, the Zstd sections
export TMPDIR=/tmp
O=gpurun_out/r03c22
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lz4_snappy.py tests/test_gpu_zstd.py tests/test_gpu_corpus.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt
for w in lz4_decompress snappy_decompress; do
  timeout 300 bash tools/kstats.sh c22_$w --workload $w --data corpus --steps 5 --warmup 2
  grep -E "execute2|parse2|value" gpurun_out/kstats_c22_$w.txt | cut -c1-140 | tee -a $O/summary.txt
done
timeout 300 python bench.py --section zstd --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read())
for k,v in r.items(): print(k, v['decompress_GiBps'], v['java_frames_decompress_GiBps'])" | tee -a $O/summary.txt
timeout 300 bash tools/pmc.sh c22 "SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES" --workload lz4_decompress --data corpus --steps 3 --warmup 1 > /dev/null 2>&1
grep -E "execute2" gpurun_out/pmc_c22.txt | cut -c1-160 | tee -a $O/summary.txt
