#!/bin/bash
# PMC counters per achip kernel of any command: tools/pmc_any.sh <tag> "<counters>" <command...>  -> gpurun_out/pmc_<tag>.txt
TAG=$1; shift
CTRS=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pm_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc $CTRS -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
python - $OUT > gpurun_out/pmc_$TAG.txt <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:70]
        if "achip" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in acc:
    for n, v in sorted(acc[k].items()):
        print("%-70s %-28s %.6g (n=%d)" % (k, n, v / cnt[k][n], cnt[k][n]))
PY
rm -rf $OUT
