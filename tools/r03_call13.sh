#!/bin/bash
# Round 3: instruction-cache, scalar-cache and memory-level counters of the Zstd match kernel (window form).  -> gpurun_out/r03c13/
export TMPDIR=/tmp
O=gpurun_out/r03c13
rm -rf $O; mkdir -p $O
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_IFETCH" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  D=$O/pmc_tmp; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $D -o pmc -- python bench.py --section zstd --no-cpu-baseline > /dev/null 2>&1
  python - $D <<'PY' >> $O/match_profile.txt
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
