#!/bin/bash
# Round 3, call 18: the Hadoop Snappy writer in two tiers (and the LZ4 writer at ten wavefronts per CU): tests, then the containers section
export TMPDIR=/tmp
O=gpurun_out/r03c18
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hadoop.py tests/test_gpu_snappy_framed.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt
timeout 300 python bench.py --section lz4frame --no-cpu-baseline > $O/containers.json 2> $O/containers.err
python - <<'PY' | tee gpurun_out/r03c18/containers.txt
import json
r = json.loads([l for l in open("gpurun_out/r03c18/containers.json") if l.startswith("{")][-1])
for k, v in r.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if "GiBps" in a})
PY
