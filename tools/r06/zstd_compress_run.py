"""N frames of 128 KiB corpus text through the Zstd level-3 encoder, device-resident (a driver for the profilers): python tools/r06/zstd_compress_run.py [frames]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import aircompressor_amd as A
from tests import common

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
fs = 131072
text = b"".join(d for _, d, _ in common.corpus_sample())
plain = [text[(i * 40000) % (len(text) - fs):][:fs] for i in range(64)]
codec = A.HipBatchCodec(0)
dev = torch.device("cuda", 0)
tile = torch.from_numpy(np.frombuffer(b"".join(plain), dtype=np.uint8).copy()).to(dev)
d_src = tile.repeat(n // 64)
cap = A.ZstdHipCompressor().max_compressed_length(fs)
d_dst = torch.zeros(n * cap + 64, dtype=torch.uint8, device=dev)
a_so = torch.arange(n, dtype=torch.int64, device=dev) * fs
a_sl = torch.full((n,), fs, dtype=torch.int32, device=dev)
a_do = torch.arange(n, dtype=torch.int64, device=dev) * cap
a_dc = torch.full((n,), cap, dtype=torch.int32, device=dev)
o_len, st, eo = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
launch = lambda: codec.launch(A.OP_ZSTD_COMPRESS, d_src, a_so, a_sl, d_dst, a_do, a_dc, o_len, st, eo, n)  # noqa: E731
launch(); codec.synchronize()
assert int(st.abs().sum().item()) == 0
t0 = time.perf_counter()
for _ in range(3):
    launch()
codec.synchronize()
t = (time.perf_counter() - t0) / 3
print("zstd compress, %d frames of 128 KiB (corpus text): %.2f ms per call, %.2f GiB/s, ratio %.3f" % (n, t * 1e3, n * fs / t / 2**30, n * fs / float(o_len.sum().item())))
