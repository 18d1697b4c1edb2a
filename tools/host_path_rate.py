"""PCIe-inclusive rate of the host-buffer boundary (achip_batch_host): host numpy in, host numpy out, staged through the
context's pinned buffer.  Reported in DESIGN.md; never used for bench.py's `value`."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import aircompressor_amd as A
from tests import oracle_lib
o = oracle_lib.load()
codec = A.HipBatchCodec(0)
rng = np.random.default_rng(1)
bs, n = 65536, 16384
frags = rng.integers(0, 256, size=(bs // 100 + 1, 50), dtype=np.uint8)
block = np.tile(frags, (1, 2)).reshape(-1)[:bs].tobytes()
comp = o.compress("lz4", block)
src = np.frombuffer(comp * n, dtype=np.uint8)
so = np.arange(n, dtype=np.int64) * len(comp); sl = np.full(n, len(comp), dtype=np.int32)
dst = np.zeros(n * bs, dtype=np.uint8); do = np.arange(n, dtype=np.int64) * bs; dc = np.full(n, bs, dtype=np.int32)
for it in range(3):
    t0 = time.perf_counter()
    ol, st, eo = codec.run_host(A.OP_LZ4_DECOMPRESS, src, so, sl, dst, do, dc)
    t = time.perf_counter() - t0
    assert (st == 0).all() and bytes(dst[:bs]) == block and bytes(dst[-bs:]) == block
    print("host-buffer LZ4 decompress: %d x %d B, %.1f ms, %.2f GiB/s decompressed (H2D %.2f GB + D2H %.2f GB)" % (n, bs, t * 1e3, n * bs / t / 2**30, src.size / 1e9, dst.size / 1e9), flush=True)

# the same batch from PINNED host segments (achip_host_alloc_pinned + explicit H2D / D2H): what a Java caller that keeps
# its MemorySegments in pinned memory gets
import ctypes
lib = codec.lib
ctx = codec.native.ctx
meta = np.concatenate([so.view(np.uint8), sl.view(np.uint8), do.view(np.uint8), dc.view(np.uint8)])
h_src = lib.achip_host_alloc_pinned(src.size); h_dst = lib.achip_host_alloc_pinned(dst.size)
ctypes.memmove(h_src, src.ctypes.data, src.size)
d_src = lib.achip_device_alloc(ctx, src.size); d_dst = lib.achip_device_alloc(ctx, dst.size + 64)
d_meta = lib.achip_device_alloc(ctx, meta.size + n * 16 + 64)
lib.achip_memcpy_h2d(ctx, d_meta, meta.ctypes.data, meta.size)
o_so, o_sl, o_do, o_dc = 0, n * 8, n * 12, n * 20
o_ol, o_st, o_eo = n * 24, n * 28, n * 32
for it in range(3):
    t0 = time.perf_counter()
    lib.achip_memcpy_h2d(ctx, d_src, h_src, src.size)
    r = lib.achip_lz4_decompress_batch(ctx, d_src, d_meta + o_so, d_meta + o_sl, d_dst, d_meta + o_do, d_meta + o_dc, d_meta + o_ol, d_meta + o_st, d_meta + o_eo, n)
    assert r == 0, r
    lib.achip_memcpy_d2h(ctx, h_dst, d_dst, dst.size)
    lib.achip_ctx_synchronize(ctx)
    t = time.perf_counter() - t0
    print("pinned segments: %.1f ms, %.2f GiB/s decompressed, %.1f GB/s over PCIe" % (t * 1e3, n * bs / t / 2**30, (src.size + dst.size) / t / 1e9), flush=True)
out = (ctypes.c_uint8 * bs).from_address(h_dst + (n - 1) * bs)
assert bytes(out) == block
