"""Development aid (needs a library built with EXTRA=-DACHIP_K3_PROBE): wave-steps of the Zstd sequence stage, how many of them took the second refill,
clocks per step.  python tools/r06/k3_probe.py [frames]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import pyarrow as pa
import aircompressor_amd as A
from tests import common

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fs = 131072
text = b"".join(d for _, d, _ in common.corpus_sample())
z = pa.Codec("zstd", compression_level=3)
rng = np.random.default_rng(3)
sets = {"corpus text": [text[(i * 40000) % (len(text) - fs):][:fs] for i in range(64)],
        "fragments": [np.tile(rng.integers(0, 256, size=(fs // 100 + 1, 50), dtype=np.uint8), (1, 2)).reshape(-1)[:fs].tobytes() for _ in range(64)]}
codec = A.HipBatchCodec(0)
dev = torch.device("cuda", 0)
for kind, plain in sets.items():
    base = [z.compress(b, asbytes=True) for b in plain]
    comp = [base[i % 64] for i in range(n)]
    lens = np.array([len(c) for c in comp], dtype=np.int32)
    pad = (lens.astype(np.int64) + 63) // 64 * 64
    s_off = np.cumsum(pad) - pad
    buf = np.zeros(int(pad.sum()) + 64, dtype=np.uint8)
    for o_, c in zip(s_off, comp):
        buf[o_:o_ + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d_src = torch.from_numpy(buf).to(dev)
    d_dst = torch.zeros(n * fs + 64, dtype=torch.uint8, device=dev)
    a_so, a_sl = torch.from_numpy(s_off).to(dev), torch.from_numpy(lens).to(dev)
    a_do = torch.arange(n, dtype=torch.int64, device=dev) * fs
    a_dc = torch.full((n,), fs, dtype=torch.int32, device=dev)
    o_len, st, eo = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
    for _ in range(2):
        codec.launch(A.OP_ZSTD_DECOMPRESS, d_src, a_so, a_sl, d_dst, a_do, a_dc, o_len, st, eo, n); codec.synchronize()
    assert int(st.abs().sum().item()) == 0
    steps, r2, kc = (codec.native.get_stat("zstd.decompress.fallback_stage%d" % k) for k in (4, 5, 6))
    print("%s, %d frames: %d wave-steps, %d with the second refill (%.1f %%), %.0f clock ticks per step (s_memtime: 100 MHz on this part? ticks x 1024 / steps)" % (
        kind, n, steps, r2, 100.0 * r2 / max(steps, 1), kc * 1024.0 / max(steps, 1)))
