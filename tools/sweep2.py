#!/usr/bin/env python3
"""Decoder sweep: variant x group x ring class x data kind (full-size batch by default)."""
import json
import subprocess
import sys

blocks = sys.argv[1] if len(sys.argv) > 1 else "131072"
wls = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lz4_decompress"]
rows = []
for wl in wls:
    for data in ("fragments", "wordmix"):
        for variant, group, ring in [(1, 1, 0), (1, 1, 1), (1, 2, 0), (1, 2, 1), (1, 4, 0), (1, 4, 1), (1, 8, 0)]:
            cmd = [sys.executable, "bench.py", "--blocks", blocks, "--pool", "2048", "--steps", "4", "--warmup", "1", "--workload", wl, "--data", data,
                   "--group", str(group), "--variant", str(variant), "--ring-class", str(ring), "--no-cpu-baseline", "--no-extra"]
            p = subprocess.run(cmd, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print((wl, data, variant, group, ring, "FAILED", p.stderr[-300:]), flush=True)
                continue
            r = json.loads(line[-1])
            print((wl, data, "v%d" % variant, "gs%d" % group, "ring%d" % ring, r["value"], r["roofline"]["frac"], r["roofline"]["kernel_ms_avg"]), flush=True)
