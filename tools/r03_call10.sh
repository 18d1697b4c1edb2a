#!/bin/bash
# Round 3, call 10: the rest of the GPU suite behind the stale options test (fixed), and the Zstd match finder in window form
# (zstd.compress.variant 3: zstd_dfast_mw.h) beside the batch-probe one, with per-kernel times and an issue profile.  -> gpurun_out/r03c10/
export TMPDIR=/tmp
O=gpurun_out/r03c10
rm -rf $O; mkdir -p $O
T0=$(date +%s)
stamp() { echo "== $1 at +$(( $(date +%s) - T0 )) s" | tee -a $O/timeline.txt; }
stamp "gpu tests"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log | tee -a $O/timeline.txt
for v in 0 3; do
  stamp "zstd section, compress variant $v"
  timeout 300 python bench.py --section zstd --no-cpu-baseline --zstd-compress-variant $v > $O/zstd_v$v.json 2> $O/zstd_v$v.err
  python - $O/zstd_v$v.json <<'PY' | tee -a gpurun_out/r03c10/timeline.txt
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for k, v in r.items():
    print(k, "compress", v["compress_GiBps"], "decode of its frames", v["java_frames_decompress_GiBps"])
PY
done
stamp "per-kernel times, variant 3"
timeout 300 bash tools/profile_zstd.sh r03c10 --no-cpu-baseline --zstd-compress-variant 3 > /dev/null 2>&1
cp gpurun_out/prof_r03c10/keep/*kernel_stats.csv $O/zstd_v3_kernel_stats.csv 2>/dev/null
grep -E "zstd_match|zstd_compress_kernel" gpurun_out/prof_r03c10/keep/dispatches.txt | tee -a $O/timeline.txt
stamp "issue profile of the match kernel, variant 3 (one pass per counter set)"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  D=$O/pmc_tmp; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $D -o pmc -- python bench.py --section zstd --no-cpu-baseline --zstd-compress-variant 3 > /dev/null 2>&1
  python - $D "$set" <<'PY' >> $O/match_profile.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zstd_match_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in acc.items():
    v = list(d.values())
    print(c, "dispatches", len(v), "per dispatch: last third avg %.4g (corpus)" % (sum(v[-len(v)//3:]) / max(1, len(v)//3)), "first third avg %.4g (fragments)" % (sum(v[:len(v)//3]) / max(1, len(v)//3)))
PY
  rm -rf $D
done
cat $O/match_profile.txt | tee -a $O/timeline.txt
stamp "done"
