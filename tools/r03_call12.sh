#!/bin/bash
# Round 3, call 12: the Zstd entropy stage fed from registers (sequence loop) and by 16-byte loads one ahead (Huffman streams), wide
# flushes; per-kernel times with the window match finder.  -> gpurun_out/r03c12/
export TMPDIR=/tmp
O=gpurun_out/r03c12
rm -rf $O; mkdir -p $O
T0=$(date +%s)
stamp() { echo "== $1 at +$(( $(date +%s) - T0 )) s" | tee -a $O/timeline.txt; }
stamp "zstd tests"
timeout 600 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstd_stream.py -m gpu -x -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log | tee -a $O/timeline.txt
for v in ${VARIANTS:-0 3}; do
  stamp "zstd section, compress variant $v"
  timeout 300 python bench.py --section zstd --no-cpu-baseline --zstd-compress-variant $v > $O/zstd_v$v.json 2> $O/zstd_v$v.err
  python - $O/zstd_v$v.json <<'PY' | tee -a gpurun_out/r03c12/timeline.txt
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for k, v in r.items():
    print(k, "compress", v["compress_GiBps"], "decode of its frames", v["java_frames_decompress_GiBps"])
PY
done
stamp "per-kernel times, variant 3"
timeout 300 bash tools/profile_zstd.sh r03c12 --no-cpu-baseline --zstd-compress-variant 3 > /dev/null 2>&1
grep -E "zstd_match|zstd_compress_kernel" gpurun_out/prof_r03c12/keep/dispatches.txt | awk '{print $1, $2}' | sort | uniq -c | sort -k2,2 -k3,3n | awk '{print $2, $3}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | cut -c1-600 | tee -a $O/timeline.txt
stamp "done"
