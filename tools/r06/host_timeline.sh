#!/bin/bash
# the host-pointer pipeline's device-side timeline: kernels and memory copies with their start / end times -> gpurun_out/r06/host_timeline_<tag>.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out/r06
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python tools/host_path_rate.py 0 96 "$@" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - $OUT > gpurun_out/r06/host_timeline_$TAG.txt <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if True:  # (every kernel: the runtime's own copy kernels too)
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("achip::", "").replace("void ", "")[:40]))
for f in glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + " ".join("%s=%s" % (k, v) for k, v in r.items() if k not in ("Start_Timestamp", "End_Timestamp", "Correlation_Id", "Kind"))))
ev.sort()
# the last call of the run: everything after the last gap of more than 50 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 50_000_000:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
for s, e, n in ev:
    if e - s > 20_000:
        print("%9.3f .. %9.3f ms  (%7.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
rm -rf $OUT
head -120 gpurun_out/r06/host_timeline_$TAG.txt
