#!/bin/bash
# extra SQ / LDS counters for one bench configuration: tools/profile_sq.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/a -o pmc -- python bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/b -o pmc -- python bench.py $ARGS > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT/c -o pmc -- python bench.py $ARGS > $OUT/c.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("a","b","c"):
    acc=collections.defaultdict(float); cnt=collections.defaultdict(int)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "decompress" in row["Kernel_Name"] and "compress_kernel" not in row["Kernel_Name"]:
                acc[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
    for k in sorted(acc): print("%-28s %.6g per dispatch (n=%d)"%(k, acc[k]/cnt[k], cnt[k]))
    import os
    for l in open("$OUT/%s.log"%d).read().splitlines():
        if "rror" in l: print("  log:", l[:160])
PY
rm -rf $OUT/a $OUT/b $OUT/c
