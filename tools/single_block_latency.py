"""Latency of ONE block per call through the host-pointer entry points (what Lz4HipDecompressor.decompress(MemorySegment, MemorySegment) costs), by
ring-decoder lane-group size and data kind.   python tools/single_block_latency.py"""
import ctypes, statistics, sys, time
import numpy as np
sys.path.insert(0, ".")
import aircompressor_amd as A
from tests import common

bs = 65536
rng = np.random.default_rng(1)
frag = rng.integers(0, 256, size=(bs // 100 + 1, 50), dtype=np.uint8)
kinds = {"fragments r=0.5": np.tile(frag, (1, 2)).reshape(-1)[:bs].copy(),
         "text": np.frombuffer(b"".join(d for _, d, _ in common.corpus_sample())[:bs], dtype=np.uint8).copy()}
for name, plain in kinds.items():
    for codec, cfn, dfn, bound in (("lz4", "achip_lz4_compress", "achip_lz4_decompress", "achip_lz4_max_compressed_length"), ("snappy", "achip_snappy_compress", "achip_snappy_decompress", "achip_snappy_max_compressed_length")):
        for group in (4, 64, 0):  # 0: the default -- the latency class (a wavefront and 128 KiB of LDS history per block) for batches of <= 256 blocks
            nat = A.HipNative(0)
            nat.set_option("%s.decompress.group" % codec, group if group else 4)
            if group:
                nat.set_option("%s.decompress.variant" % codec, 1)
            nat.set_option("decompress.latency_max_blocks", 0 if group else 256)
            lib = nat.lib
            cap = getattr(lib, bound)(bs)
            comp = np.zeros(cap, dtype=np.uint8)
            back = np.zeros(bs, dtype=np.uint8)
            eo = ctypes.c_int64()
            n = getattr(lib, cfn)(nat.ctx, plain.ctypes.data, comp.ctypes.data, bs, cap, ctypes.byref(eo))
            assert n > 0
            td = []
            for it in range(120):
                t0 = time.perf_counter()
                r = getattr(lib, dfn)(nat.ctx, comp.ctypes.data, back.ctypes.data, n, bs, ctypes.byref(eo))
                td.append(time.perf_counter() - t0)
                assert r == bs
            assert (back == plain).all()
            print("%-16s %-6s %s: decompress %7.1f us (ratio %.2f)" % (name, codec, ("compact rings, %2d lanes per block" % group) if group else "default (few blocks: by a look at the tokens)", statistics.median(td[20:]) * 1e6, bs / n), flush=True)
            nat.close()
