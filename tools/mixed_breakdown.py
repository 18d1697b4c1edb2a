"""Where the mixed corpus batch (bench.py's configs[4] leg) spends its time: every (codec, direction) bucket of one copy of the job timed on its own,
with its longest item.   python tools/mixed_breakdown.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
import aircompressor_amd as A
from tests.gpu_harness import GpuBatch

rows, items, weights = bench.mixed_job(1)
files = bench.corpus_files()
gb = GpuBatch(0)
lib = gb.codec.lib
bound = {"lz4": lib.achip_lz4_max_compressed_length, "snappy": lib.achip_snappy_max_compressed_length, "zstd": lib.achip_zstd_max_compressed_length}
plain = lambda r: files[rows[r][0]][rows[r][1]:rows[r][1] + rows[r][2]]
for codec in ("lz4", "snappy", "zstd"):
    cop, dop = bench.MIXED_OPS[codec]
    rs = [r for r in range(len(rows)) if rows[r][3] == codec]
    blocks = [plain(r) for r in rs]
    caps = [bound[codec](len(b)) for b in blocks]
    gb.run(cop, blocks[:4], caps[:4])
    t0 = time.perf_counter(); outs, st, _ = gb.run(cop, blocks, caps); tc = time.perf_counter() - t0
    assert all(s == 0 for s in st)
    big = [(b, c) for b, c in zip(blocks, caps) if len(b) > 1 << 20]
    t0 = time.perf_counter(); gb.run(cop, [b for b, _ in big], [c for _, c in big]); tcb = time.perf_counter() - t0
    t0 = time.perf_counter(); back, st, _ = gb.run(dop, outs, [max(len(b), 1) for b in blocks]); td = time.perf_counter() - t0
    assert all(s == 0 for s in st)
    bigd = [(o_, len(b)) for o_, b in zip(outs, blocks) if len(b) > 1 << 20]
    t0 = time.perf_counter(); gb.run(dop, [o_ for o_, _ in bigd], [n for _, n in bigd]); tdb = time.perf_counter() - t0
    print("%-6s %4d items %6.1f MB: compress %6.0f ms (the %d items above 1 MiB alone: %6.0f ms), decompress %6.0f ms (those alone: %6.0f ms)   [wall incl. copies]" % (
        codec, len(rs), sum(len(b) for b in blocks) / 1e6, tc * 1e3, len(big), tcb * 1e3, td * 1e3, tdb * 1e3), flush=True)
