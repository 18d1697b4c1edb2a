"""PCIe-inclusive rate of the host-buffer boundary (achip_batch_host) on ordinary (pageable) memory, swept over the pipeline's two knobs --
copy threads per pool and staging bytes per chunk -- with the pipeline's own stage times (achip_ctx_get_stat host.*).
bench.py reports the default configuration as `end_to_end`; never used for `value`.
    python tools/host_path_rate.py [threads,threads,...] [chunk_MiB,chunk_MiB,...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import aircompressor_amd as A

threads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4,8,16,32").split(",")]
chunks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "192").split(",")]
bs, n, pool_n = 65536, 16384, 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
frags = torch.randint(0, 256, (pool_n * bs // 100 + 1, 50), dtype=torch.uint8, device=dev, generator=g)
plain = frags.repeat(1, 2).reshape(-1)[:pool_n * bs].contiguous().cpu().numpy()
base = A.HipBatchCodec(0)
cap = base.lib.achip_lz4_max_compressed_length(bs)
comp = np.zeros(pool_n * cap, dtype=np.uint8)
ol, st, _ = base.run_host(A.OP_LZ4_COMPRESS, plain, np.arange(pool_n, dtype=np.int64) * bs, np.full(pool_n, bs, dtype=np.int32), comp, np.arange(pool_n, dtype=np.int64) * cap, np.full(pool_n, cap, dtype=np.int32))
assert (st == 0).all()
reps = n // pool_n
src = np.tile(comp, reps)
so = np.tile(np.arange(pool_n, dtype=np.int64) * cap, reps) + np.repeat(np.arange(reps, dtype=np.int64) * comp.size, pool_n)
sl = np.tile(ol, reps)
dst = np.zeros(n * bs, dtype=np.uint8); do = np.arange(n, dtype=np.int64) * bs; dc = np.full(n, bs, dtype=np.int32)
for t in threads:
    for ch in chunks:
        codec = A.HipBatchCodec(0)
        codec.native.set_option("host.copy_threads", t)
        codec.native.set_option("host.chunk_bytes", ch << 20)
        for a_ in sys.argv[3:]:  # e.g. lz4.decompress.variant=7: further context options
            k_, v_ = a_.split("=")
            codec.native.set_option(k_, int(v_))
        best = None
        for it in range(4):
            t0 = time.perf_counter()
            o2, s2, _ = codec.run_host(A.OP_LZ4_DECOMPRESS, src, so, sl, dst, do, dc)
            el = time.perf_counter() - t0
            assert (s2 == 0).all() and (dst[:pool_n * bs] == plain).all()
            if it and (best is None or el < best[0]):
                best = (el, {k: codec.native.get_stat("host." + k) for k in ("chunks", "total_us", "gather_us", "scatter_us", "wait_slot_us", "wait_download_us")})
        print("threads/pool %2d chunk %4d MiB: %6.2f GiB/s decompressed  %s" % (t, ch, n * bs / best[0] / 2**30, best[1]), flush=True)
        codec.native.close()
