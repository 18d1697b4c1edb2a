#!/bin/bash
# kernel-trace of the Zstd section of bench.py: per-dispatch time of every pipeline stage and of the encoder
# usage: tools/profile_zstd.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/keep
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --section zstd $* > $OUT/keep/stats.log 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/keep/ \;
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "achip::" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("(")[0].replace("achip::", "").replace("void ", ""), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                         r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
rows.sort()
with open(out + "/keep/dispatches.txt", "w") as w:
    w.write("kernel us grid vgpr lds\n")
    for _, n, d, g, v, l in rows:
        w.write("%-48s %10.1f %9s %4s %6s\n" % (n, d / 1e3, g, v, l))
PY
rm -rf $OUT/stats
grep -o '{"zstd_fragments.*' $OUT/keep/stats.log | tail -1
