#!/bin/bash
# Round 3, call 19: the Zstd sequences stage (K3) with its bit stream through an LDS ring: Zstd tests + fuzz, the Zstd sections, per-dispatch times
export TMPDIR=/tmp
O=gpurun_out/r03c19
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zstd.py tests/test_gpu_zstd_stream.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt
timeout 300 python bench.py --section zstd --no-cpu-baseline > $O/zstd.json 2> $O/zstd.err
timeout 300 python bench.py --section zstdstream --no-cpu-baseline > $O/zstdstream.json 2> $O/zstdstream.err
python - <<'PY' | tee gpurun_out/r03c19/summary.txt
import json
for f in ("zstd", "zstdstream"):
    r = json.loads([l for l in open("gpurun_out/r03c19/%s.json" % f) if l.startswith("{")][-1])
    for k, v in r.items():
        if isinstance(v, dict):
            print(k, {a: b for a, b in v.items() if "GiBps" in a or "fallback" in a})
PY
timeout 300 bash tools/profile_zstd.sh r03c19 --no-cpu-baseline > /dev/null 2>&1
grep -E "sequences|literals_kernel|parse_kernel|execute2" gpurun_out/prof_r03c19/keep/dispatches.txt | awk '{print $1, $2}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | cut -c1-400 | tee -a $O/summary.txt
timeout 600 python tools/fuzz_decoders.py 20000 31 zstd 2>&1 | tail -4 | tee -a $O/summary.txt
