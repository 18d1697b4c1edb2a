#!/bin/bash
# Round 3, call 13: issue profile of the Zstd match kernel (window form, v2).  -> gpurun_out/r03c13/
export TMPDIR=/tmp
O=gpurun_out/r03c13
rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_FLAT SQ_WAVES_EQ_64 SQ_INSTS_EXP_GDS"; do
  D=$O/pmc_tmp; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $D -o pmc -- python bench.py --section zstd --no-cpu-baseline > /dev/null 2>&1
  python - $D "$set" <<'PY' >> $O/match_profile.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "zstd_match_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in acc.items():
    v = [d[k] for k in sorted(d, key=int)]
    n = len(v) // 3
    print(c, "dispatches", len(v), "corpus %.4g" % (sum(v[-n:]) / max(1, n)), "fragments %.4g" % (sum(v[:n]) / max(1, n)))
PY
  rm -rf $D
done
cat $O/match_profile.txt
