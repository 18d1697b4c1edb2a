"""One large buffer per achip_batch call through the Snappy encoder (device-resident): the sub-blocks side by side (snappy.compress.fan = 1, the default)
against one wavefront per buffer (0).  python tools/r06/big_snappy.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import aircompressor_amd as A
from tests import common, oracle_lib

o = oracle_lib.load()
corpus = common.corpus_full()
files = sorted(corpus.items(), key=lambda kv: -len(kv[1]))[:3]
dev = torch.device("cuda", 0)
for fan in (1, 0):
    codec = A.HipBatchCodec(0)
    codec.native.set_option("snappy.compress.fan", fan)
    for name, data in files + [("16 x " + files[0][0], None)]:
        items = [bytes(files[0][1])] * 16 if data is None else [bytes(data)]
        n = len(items)
        lens = np.array([len(b) for b in items], dtype=np.int32)
        caps = np.array([o.max_compressed_length("snappy", len(b)) for b in items], dtype=np.int32)
        s_off = np.cumsum(lens.astype(np.int64)) - lens
        d_off = np.cumsum(caps.astype(np.int64)) - caps
        d_src = torch.from_numpy(np.frombuffer(b"".join(items), dtype=np.uint8).copy()).to(dev)
        d_dst = torch.zeros(int(caps.sum()) + 64, dtype=torch.uint8, device=dev)
        a_so, a_sl = torch.from_numpy(s_off).to(dev), torch.from_numpy(lens).to(dev)
        a_do, a_dc = torch.from_numpy(d_off).to(dev), torch.from_numpy(caps).to(dev)
        o_len, st, eo = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
        launch = lambda: codec.launch(A.OP_SNAPPY_COMPRESS, d_src, a_so, a_sl, d_dst, a_do, a_dc, o_len, st, eo, n)  # noqa: E731
        launch(); codec.synchronize()
        want = o.compress("snappy", items[0])
        got = d_dst[:int(o_len[0].item())].cpu().numpy().tobytes()
        t0 = time.perf_counter()
        for _ in range(3):
            launch()
        codec.synchronize()
        t = (time.perf_counter() - t0) / 3
        print("snappy.compress.fan = %d, %-28s %9d bytes x %2d: %8.2f ms per call, %6.3f GiB/s, status %d, bytes %s" % (
            fan, name, len(items[0]), n, t * 1e3, sum(len(b) for b in items) / t / 2**30, int(st.abs().sum().item()), "identical to the oracle's" if got == want else "DIFFER"), flush=True)
