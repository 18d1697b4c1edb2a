import ctypes, statistics, sys, time
import numpy as np
sys.path.insert(0, ".")
import aircompressor_amd as A
from tests import common
bs = 65536
rng = np.random.default_rng(1)
frag = rng.integers(0, 256, size=(bs // 100 + 1, 50), dtype=np.uint8)
kinds = {"fragments": np.tile(frag, (1, 2)).reshape(-1)[:bs].copy(), "text": np.frombuffer(b"".join(d for _, d, _ in common.corpus_sample())[:bs], dtype=np.uint8).copy()}
for name, plain in kinds.items():
    for parse in (1, 2):
        nat = A.HipNative(0)
        nat.set_option("lz4.decompress.variant", 7)
        nat.set_option("lz4.decompress.parse", parse)
        lib = nat.lib
        cap = lib.achip_lz4_max_compressed_length(bs)
        comp = np.zeros(cap, dtype=np.uint8); back = np.zeros(bs, dtype=np.uint8); eo = ctypes.c_int64()
        n = lib.achip_lz4_compress(nat.ctx, plain.ctypes.data, comp.ctypes.data, bs, cap, ctypes.byref(eo))
        td = []
        for it in range(60):
            t0 = time.perf_counter()
            r = lib.achip_lz4_decompress(nat.ctx, comp.ctypes.data, back.ctypes.data, n, bs, ctypes.byref(eo))
            td.append(time.perf_counter() - t0)
            assert r == bs
        assert (back == plain).all()
        print("%-10s lz4 two passes, parse %d: %7.1f us" % (name, parse, statistics.median(td[10:]) * 1e6), flush=True)
