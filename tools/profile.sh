#!/bin/bash
# rocprofv3 capture used for profiles/: kernel-trace stats in one run, PMC counters in separate runs.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- python bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM -d $OUT/pmc_sq -o pmc -- python bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc_tcp -o pmc -- python bench.py $ARGS > $OUT/pmc_tcp.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small artefacts (gpurun copies back <= 64 MiB)
mkdir -p $OUT/keep; find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/keep/ \; ; find $OUT -name '*.log' -size -200k -exec cp {} $OUT/keep/ \; ; cp $OUT/summary.txt $OUT/keep/; cp $OUT/traffic.json $OUT/keep/ 2>/dev/null; rm -rf $OUT/stats $OUT/pmc_*; find gpurun_out -size +20M -delete
