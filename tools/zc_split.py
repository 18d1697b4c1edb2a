"""DEBUG: time the Zstd encoder with and without the entropy stages (variant 100 stops after the match finder)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import aircompressor_amd as A
import bench
class Args: ratio = 0.5
dev = torch.device("cuda", 0)
codec = A.HipBatchCodec(0)
for fs, n, kind in ((131072, 8192, "fragments"), (131072, 8192, "wordmix"), (32768, 32768, "wordmix"), (65536, 16384, "corpus")):
    plain = bench.gen_data(torch, dev, kind, n, fs, 0.5, 4242)
    torch.cuda.synchronize()  # the codec runs on its own stream
    max_c = codec.lib.achip_zstd_max_compressed_length(fs)
    cs = (max_c + 15) // 16 * 16
    i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
    so = torch.arange(n, **i64) * fs; sl = torch.full((n,), fs, **i32)
    dst = torch.empty(n * cs + 64, dtype=torch.uint8, device=dev); do = torch.arange(n, **i64) * cs; dc = torch.full((n,), max_c, **i32)
    ol = torch.zeros(n, **i32); st = torch.zeros(n, **i32); eo = torch.zeros(n, **i64)
    for variant in (0, 2, 100):
        codec.native.set_option("zstd.compress.variant", variant)
        f = lambda: codec.launch(A.OP_ZSTD_COMPRESS, plain, so, sl, dst, do, dc, ol, st, eo, n)
        f(); codec.synchronize()
        e0, e1 = codec.event(), codec.event()
        codec.record(e0); f(); codec.record(e1)
        t = codec.elapsed_ms(e0, e1)
        print(kind, fs, "variant", variant, "%.1f ms" % t, "%.2f GiB/s" % (n * fs / t / 1e-3 / 2**30), flush=True)
