#!/bin/bash
# instruction-mix counters of the decode kernel for one bench configuration
# usage: tools/profile_sq2.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/sq2_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra --steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $OUT/p1 -o pmc -- python bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $OUT/p2 -o pmc -- python bench.py $ARGS > $OUT/p2.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decompress" in r["Kernel_Name"] and "compress_" not in r["Kernel_Name"].replace("decompress_", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as w:
    for k in sorted(acc):
        w.write("%-28s per_dispatch=%.6g (n=%d)\n" % (k, acc[k] / cnt[k], cnt[k]))
print(open(out + "/summary.txt").read())
PY
grep -h "rror" $OUT/p1.log $OUT/p2.log | head -3
rm -rf $OUT/p1 $OUT/p2
