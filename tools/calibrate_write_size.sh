#!/bin/bash
# calibrates rocprofv3's WRITE_SIZE / FETCH_SIZE on a plain device copy of known size (MI355X_MICROARCH.md: "calibrate on a known byte count")
export TMPDIR=/tmp
OUT=gpurun_out/calib
mkdir -p $OUT
cat > /tmp/calib.py <<'PY'
import torch
a = torch.empty(1 << 32, dtype=torch.uint8, device="cuda")   # 4 GiB
b = torch.empty_like(a)
a.fill_(7)
torch.cuda.synchronize()
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
PY
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/$c -o pmc -- python /tmp/calib.py > $OUT/$c.log 2>&1
done
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and float(r["Counter_Value"]) > 1e5:
                print(c, r["Kernel_Name"][:70], "value=%s (x1024 = %.3f GiB; copy = 4 GiB)" % (r["Counter_Value"], float(r["Counter_Value"]) * 1024 / 2**30))
PY
rm -rf $OUT/WRITE_SIZE $OUT/FETCH_SIZE
