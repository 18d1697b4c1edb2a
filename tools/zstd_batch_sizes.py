"""Zstd level-3 decode against the batch size: N frames of 128 KiB (corpus text, libzstd's frames), device-resident, time per call.
   python tools/zstd_batch_sizes.py [sizes...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import pyarrow as pa
import aircompressor_amd as A
from tests import common

args = [x for x in sys.argv[1:] if not x.startswith("--")]
sizes = [int(x) for x in args] or [64, 256, 1024, 4096, 8192, 16384, 65536]
kind = "fragments" if "--fragments" in sys.argv else "corpus text"
text = b"".join(d for _, d, _ in common.corpus_sample())
fs = 131072
z = pa.Codec("zstd", compression_level=3)
if kind == "fragments":  # (RandomGenerator-like: 50-byte random fragments, each twice)
    rng = np.random.default_rng(3)
    plain = [np.tile(rng.integers(0, 256, size=(fs // 100 + 1, 50), dtype=np.uint8), (1, 2)).reshape(-1)[:fs].tobytes() for _ in range(64)]
else:
    plain = [text[(i * 40000) % (len(text) - fs):][:fs] for i in range(64)]
base = [z.compress(b, asbytes=True) for b in plain]
codec = A.HipBatchCodec(0)
for a_ in sys.argv[1:]:
    if a_.startswith("--exec="):  # the execute stage: 2 chosen per item, 1 record executor, 0 rings (process-wide)
        codec.native.set_option("zstd.decompress.exec", int(a_[7:]))
dev = torch.device("cuda", 0)
for n in sizes:
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
    torch.cuda.synchronize()
    launch = lambda: codec.launch(A.OP_ZSTD_DECOMPRESS, d_src, a_so, a_sl, d_dst, a_do, a_dc, o_len, st, eo, n)  # noqa: E731
    launch(); codec.synchronize()
    assert int(st.abs().sum().item()) == 0 and int(o_len.sum().item()) == n * fs
    t0 = time.perf_counter()
    for _ in range(5):
        launch()
    codec.synchronize()
    t = (time.perf_counter() - t0) / 5
    print("zstd decompress, %6d frames of 128 KiB (%s): %8.2f ms per call, %7.1f GiB/s" % (n, kind, t * 1e3, n * fs / t / 2**30), flush=True)
