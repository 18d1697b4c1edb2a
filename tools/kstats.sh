#!/bin/bash
# per-kernel time of one bench.py run: tools/kstats.sh <tag> [bench args...]  -> gpurun_out/kstats_<tag>.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python bench.py --no-extra --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
F=$(find $OUT -name '*kernel_stats.csv' | head -1)
python - "$F" > gpurun_out/kstats_$TAG.txt <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "achip" in row["Name"]:
        print("%-90s calls=%s avg_ms=%.3f" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e6))
PY
grep '^{' $OUT/log.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('value', r['value'], 'frac', r['roofline']['frac'])" >> gpurun_out/kstats_$TAG.txt
rm -rf $OUT
