#!/usr/bin/env python3
"""Variant sweep on one GPU: decoder group size x data kind -> GiB/s table (used to pick defaults; results in profiles/)."""
import json
import subprocess
import sys

blocks = sys.argv[1] if len(sys.argv) > 1 else "32768"
rows = []
for wl in ("lz4_decompress", "snappy_decompress"):
    for data in ("fragments", "wordmix"):
        for group in (1, 2, 4, 8, 16, 32, 64):
            cmd = [sys.executable, "bench.py", "--blocks", blocks, "--pool", "2048", "--steps", "5", "--warmup", "2", "--workload", wl, "--data", data,
                   "--group", str(group), "--no-cpu-baseline", "--no-extra"]
            p = subprocess.run(cmd, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                rows.append((wl, data, group, "FAILED", p.stderr[-400:]))
                continue
            r = json.loads(line[-1])
            rows.append((wl, data, group, r["value"], r["roofline"]["frac"], r["config"]["compression_ratio"], r["roofline"]["kernel_ms_avg"]))
            print(rows[-1], flush=True)
print(json.dumps(rows))
