#!/bin/bash
# Round 3, call 23: FETCH_SIZE / WRITE_SIZE per kernel -- the Zstd stages and encoder kernels on the corpus batch, the LZ4 / Snappy encoders on corpus
# (separate --pmc passes; FETCH_SIZE x 2 on gfx950, WRITE_SIZE as counted: profiles/r03_counter_calibration.txt)
export TMPDIR=/tmp
O=gpurun_out/r03c23
rm -rf $O; mkdir -p $O
run() {  # tag, counter, bench args...
  TAG=$1; CTR=$2; shift; shift
  D=$O/tmp; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $CTR -d $D -o p -- python bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
  python - $D $TAG $CTR <<'PY' >> $O/traffic.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    per = collections.defaultdict(float)
    order = []
    for r in rows:
        if "achip::" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]:
            k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("achip::", "")[:44], r.get("Grid_Size", ""))
            if k not in per: order.append(k)
            per[k] += float(r["Counter_Value"])
    for k in order:
        acc[(k[1], k[2])].append(per[k])
for (name, grid), v in acc.items():
    scale = 2048 if sys.argv[3] == "FETCH_SIZE" else 1024   # KiB -> bytes (FETCH_SIZE counts half)
    third = v[-max(1, len(v) // 3):] if sys.argv[2] == "zstd" else v[1:] or v
    print("%-10s %-12s %-46s grid %9s  dispatches %3d  GB per dispatch (corpus) %8.3f" % (sys.argv[2], sys.argv[3], name, grid, len(v), sum(third) / len(third) * scale / 1e9))
PY
  rm -rf $D
}
for c in FETCH_SIZE WRITE_SIZE; do
  run zstd $c --section zstd
  run lz4c $c --no-extra --no-sweep --workload lz4_compress --data corpus --blocks 65536 --steps 3 --warmup 1
  run snappyc $c --no-extra --no-sweep --workload snappy_compress --data corpus --blocks 65536 --steps 3 --warmup 1
done
sort $O/traffic.txt | grep -v " 0.000$" | cut -c1-170
