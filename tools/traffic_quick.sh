#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one bench configuration (two rocprofv3 PMC passes).  usage: tools/traffic_quick.sh <tag> [bench args...]
TAG=$1; shift
OUT=gpurun_out/tq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra --steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/f -o pmc -- python bench.py $ARGS > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/w -o pmc -- python bench.py $ARGS > $OUT/w.log 2>&1
python - <<PY
import csv, glob, collections
for tag, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name and "decompress" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[1:] if len(v) > 1 else v
        print("$TAG", name, k, "per dispatch GB: %.1f" % (sum(v) / len(v) * 1024 * (2 if name == "FETCH_SIZE" else 1) / 1e9), "n=%d" % len(v))
PY
tail -1 $OUT/f.log | cut -c95-140
rm -rf $OUT/f $OUT/w
