#!/bin/bash
# Per-kernel time, HBM traffic and issue counters of any command: tools/r06/counters.sh <tag> <command...>  -> gpurun_out/r06/counters_<tag>.txt
# Separate rocprofv3 passes (kernel-trace only beside --pmc, as MI355X_MICROARCH.md prescribes): --stats; FETCH_SIZE; WRITE_SIZE (+ TCC hit / miss);
# SQ instruction counts; SQ busy / wait cycles.  FETCH_SIZE is doubled (gfx950 counts half the bytes of wide reads; calibrated in round 3), WRITE_SIZE as counted.
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/cn_$TAG
rm -rf $OUT; mkdir -p $OUT gpurun_out/r06
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- "$@" > $OUT/stats.log 2>&1
pass() { rocprofv3 --kernel-trace --output-format csv --pmc $2 -d $OUT/$1 -o p -- "${@:3}" > $OUT/$1.log 2>&1; }
pass fetch "FETCH_SIZE" "$@"
pass write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "$@"
pass insts "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "$@"
pass cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "$@"
python - $OUT "$TAG" "$*" > gpurun_out/r06/counters_$TAG.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict, OrderedDict
out, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
def short(n):
    return n.split("(")[0].replace("achip::", "").replace("void ", "")[:78]
t = OrderedDict()
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "achip" in r["Name"]:
            t[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "achip" in r.get("Kernel_Name", ""):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("# %s\n# %s\n# per dispatch (averages over the run's dispatches of each kernel); HBM read = FETCH_SIZE KiB x 1024 x 2, written = WRITE_SIZE KiB x 1024" % (tag, cmd))
for k, (calls, ms) in sorted(t.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if calls * ms < 0.02:
        continue
    a = {n: acc[k][n] / cnt[k][n] for n in acc[k]}
    rd, wr = a.get("FETCH_SIZE", 0) * 2048, a.get("WRITE_SIZE", 0) * 1024
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    print("%s\n    calls %d  avg %.3f ms | HBM read %.3f GB  written %.3f GB  (%.0f GB/s) | L2 hit %.0f %%" % (
        k, calls, ms, rd / 1e9, wr / 1e9, (rd + wr) / 1e9 / (ms / 1e3) if ms else 0, 100.0 * a.get("TCC_HIT_sum", 0) / max(a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0), 1)))
    print("    waves %.0f | wave instructions: VALU %.4g  SALU %.4g  LDS %.4g  VMEM rd %.4g wr %.4g  branch %.4g  SMEM %.4g" % (
        a.get("SQ_WAVES", 0), a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_SALU", 0), a.get("SQ_INSTS_LDS", 0), a.get("SQ_INSTS_VMEM_RD", 0), a.get("SQ_INSTS_VMEM_WR", 0),
        a.get("SQ_INSTS_BRANCH", 0), a.get("SQ_INSTS_SMEM", 0)))
    print("    of the wave-cycles: waiting (s_waitcnt / barrier) %.0f %%  issue stall %.0f %%  issuing %.0f %%  (VALU %.0f %%, scalar %.0f %%, LDS %.0f %%)" % (
        100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * a.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_SCA", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_LDS", 0) / wc))
PY
rm -rf $OUT
cat gpurun_out/r06/counters_$TAG.txt
