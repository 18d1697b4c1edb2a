"""GPU parity tests for the batched XXH64 / XXH32 kernels through the C ABI: equal to the oracle
(which is pinned by the reference's KATs, T/zstd/TestXxHash64.java:39-61 and T/xxhash/TestXxHash32.java:45-46) for every
length around the stripe / tail boundaries, several seeds, unaligned buffers, and at full size."""
import ctypes

import numpy as np
import pytest

from tests import common, oracle_lib

pytestmark = pytest.mark.gpu
SEEDS = [0, 1, 0x9E3779B1, -1, 2**31 - 1, -2**31]  # T/xxhash/TestXxHash32.java:30


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


@pytest.fixture(scope="module")
def gb():
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0)


def run_batch(gb, wide, buffers, seed, misalign=3):
    lib, ctx = gb.codec.lib, gb.codec.native.ctx
    n = len(buffers)
    offs, pos = [], misalign
    for b in buffers:
        offs.append(pos)
        pos += len(b) + (len(b) % 7) + 1
    src = np.zeros(pos + 64, dtype=np.uint8)
    for b, so in zip(buffers, offs):
        src[so:so + len(b)] = np.frombuffer(b, dtype=np.uint8)
    so = np.array(offs, dtype=np.int64)
    sl = np.array([len(b) for b in buffers], dtype=np.int32)
    out = np.zeros(n, dtype=np.int64 if wide else np.int32)
    d = [lib.achip_device_alloc(ctx, a.nbytes + 64) for a in (src, so, sl, out)]
    for dp, a in zip(d[:3], (src, so, sl)):
        assert lib.achip_memcpy_h2d(ctx, dp, a.ctypes.data, a.nbytes) == 0
    if wide:
        r = lib.achip_xxhash64_batch(ctx, d[0], d[1], d[2], ctypes.c_int64(seed), d[3], n)
    else:
        r = lib.achip_xxhash32_batch(ctx, d[0], d[1], d[2], ctypes.c_int32(seed), d[3], n)
    assert r == 0
    assert lib.achip_memcpy_d2h(ctx, out.ctypes.data, d[3], out.nbytes) == 0
    assert lib.achip_ctx_synchronize(ctx) == 0
    for dp in d:
        lib.achip_device_free(ctx, dp)
    return out


@pytest.mark.parametrize("wide", [True, False], ids=["xxh64", "xxh32"])
def test_every_length_and_seed(gb, o, wide):
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()
    lengths = list(range(0, 200)) + [255, 256, 257, 1000, 4095, 4096, 4097, 65535, 65536, 65537]
    buffers = [base[i % 13:i % 13 + n] for i, n in enumerate(lengths)]
    buffers += [b"", b"abc", b"a" * 1000] + [d for _, d in common.HAND_CASES]
    for seed in SEEDS:
        got = run_batch(gb, wide, buffers, seed)
        for b, g in zip(buffers, got):
            want = o.xxh64(b, seed & (2**64 - 1)) if wide else o.xxh32(b, seed & 0xFFFFFFFF)
            assert int(g) & ((1 << (64 if wide else 32)) - 1) == want, (len(b), seed)


def test_reference_kats_through_the_abi(gb):
    # T/xxhash/TestXxHash32.java:45-46 and T/zstd/TestXxHash64.java:42 (empty input, seed 0)
    assert [int(v) & 0xFFFFFFFF for v in run_batch(gb, False, [b"", b"abc"], 0)] == [0x02CC5D05, 0x32D153FF]
    assert int(run_batch(gb, True, [b""], 0)[0]) & (2**64 - 1) == 0xEF46DB3751D8E999


def test_one_shot_host_api_mirrors_reference(o):
    import aircompressor_amd as A
    h64, h32 = A.XxHash64HipHasher(), A.XxHash32HipHasher()
    data = common.corpus_sample()[0][1]
    assert h64.hash(data) & (2**64 - 1) == o.xxh64(data)
    assert h32.hash(data) & 0xFFFFFFFF == o.xxh32(data)
    assert h64.hash(data, 5, 1000, seed=42) & (2**64 - 1) == o.xxh64(data[5:1005], 42)
    assert h32.hash(data, 7, 33, seed=-1) & 0xFFFFFFFF == o.xxh32(data[7:40], 0xFFFFFFFF)
    assert h32.hash(b"") == 0x02CC5D05
    with pytest.raises(IndexError):
        h64.hash(b"abc", 2, 5)


def test_full_size_property(gb, o):
    """65536 x 64 KiB buffers: every copy of a block hashes to the oracle's value"""
    blocks = [d for _, d, _ in common.corpus_sample()][:8]
    got = run_batch(gb, True, blocks * 512, 7, misalign=0)
    want = [o.xxh64(b, 7) for b in blocks] * 512
    assert [int(v) & (2**64 - 1) for v in got] == want
