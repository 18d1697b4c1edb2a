"""Pins the CPU oracle against every golden vector / known-answer test the reference's own tests
hold for the hot path.  Runs without a GPU and without /root/reference."""
import numpy as np
import pytest

from tests import common, oracle_lib
from tests.oracle_lib import OracleError

D = lambda s: s  # noqa: E731


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


# ---- Zstd decoder golden fixtures: T/zstd/AbstractTestZstd.java:41-78,175-184 ----
def test_zstd_with_checksum(o):
    z, p = common.golden_zstd("with-checksum.zst"), common.golden_zstd("with-checksum")
    assert o.decompress("zstd", z, len(p)) == p
    assert o.decompress("zstd", z, len(p) + 37) == p


def test_zstd_multiple_frames(o):
    z, p = common.golden_zstd("multiple-frames.zst"), common.golden_zstd("multiple-frames")
    assert o.decompress("zstd", z, len(p)) == p


def test_zstd_offset_before_start(o):
    with pytest.raises(OracleError) as e:
        o.decompress("zstd", common.golden_zstd("offset-before-start.zst"), 1 << 20)
    assert e.value.cls == 1 and e.value.detail == 34  # "Input is corrupted"


def test_zstd_bad_second_frame(o):
    with pytest.raises(OracleError) as e:
        o.decompress("zstd", common.golden_zstd("bad-second-frame.zst"), 1 << 20)
    assert e.value.cls == 1 and e.value.detail == 35  # "Invalid magic prefix"


def test_zstd_output_too_small(o):
    z, p = common.golden_zstd("with-checksum.zst"), common.golden_zstd("with-checksum")
    with pytest.raises(OracleError) as e:
        o.decompress("zstd", z, len(p) - 1)
    assert e.value.detail == 33  # "Output buffer too small"


def test_zstd_bad_checksum(o):
    z = bytearray(common.golden_zstd("with-checksum.zst"))
    z[-1] ^= 0x55
    with pytest.raises(OracleError) as e:
        o.decompress("zstd", bytes(z), 1 << 20)
    assert e.value.detail == 37


def test_zstd_truncated_and_garbage(o):
    z = common.golden_zstd("with-checksum.zst")
    for cut in (0, 1, 3, 4, 5, 6, 9, 100, len(z) - 5, len(z) - 1):
        if cut == 0:
            assert o.decompress("zstd", b"", 100) == b""  # loop never runs: 0 bytes
            continue
        with pytest.raises(OracleError):
            o.decompress("zstd", z[:cut], 1 << 20)


# T/zstd/AbstractTestZstd.java:140-147
def test_zstd_max_compressed_length(o):
    assert o.max_compressed_length("zstd", 0) == 64
    assert o.max_compressed_length("zstd", 64 * 1024) == 65_824
    assert o.max_compressed_length("zstd", 128 * 1024) == 131_584
    assert o.max_compressed_length("zstd", 128 * 1024 + 1) == 131_585


def test_zstd_decompressed_size(o):
    import ctypes
    z = np.frombuffer(common.golden_zstd("with-checksum.zst"), dtype=np.uint8)
    eo = ctypes.c_int64()
    assert o.lib.orc_zstd_decompressed_size(z.ctypes.data, len(z), ctypes.byref(eo)) == -1  # no content size in this frame


# ---- XXH64: T/zstd/TestXxHash64.java:30-61 ----
def test_xxh64_kats(o):
    prime = 2654435761
    value = prime
    buf = bytearray(101)
    for i in range(101):
        buf[i] = (value >> 24) & 0xFF
        value = (value * value) & 0xFFFFFFFFFFFFFFFF
        # Java long arithmetic: `value >> 24` on a signed long, then (byte) cast: low 8 bits identical
    kats = [
        (0, 0, 0xEF46DB3751D8E999), (0, 1, 0x4FCE394CC88952D8), (prime, 1, 0x739840CB819FA723),
        (0, 4, 0x9256E58AA397AEF1), (prime, 4, 0x9D5FFDFB928AB4B), (0, 8, 0xF74CB1451B32B8CF), (prime, 8, 0x9C44B77FBCC302C5),
        (0, 14, 0xCFFA8DB881BC3A3D), (prime, 14, 0x5B9611585EFCC9CB), (0, 32, 0xAF5753D39159EDEE), (prime, 32, 0xDCAB9233B8CA7B0F),
        (0, 101, 0x0EAB543384F878AD), (prime, 101, 0xCAA65939306F1E21),
    ]
    for seed, n, expected in kats:
        assert o.xxh64(bytes(buf[:n]), seed) == expected, (seed, n)


def test_xxh32_kats_and_python_xxhash(o):
    """T/xxhash/TestXxHash32.java:45-46 pins the two values; the third-party `xxhash` module checks every length class and
    the seeds of T/xxhash/TestXxHash32.java:30"""
    assert o.xxh32(b"", 0) == 0x02CC5D05
    assert o.xxh32(b"abc", 0) == 0x32D153FF
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=5000, dtype=np.uint8).tobytes()
    for seed in (0, 1, 0x9E3779B1, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000):
        for n in list(range(0, 70)) + [255, 256, 1023, 4096, 5000]:
            assert o.xxh32(data[:n], seed) == xxhash.xxh32(data[:n], seed=seed).intdigest(), (n, seed)


def test_xxh64_vs_python_xxhash(o):
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=3000, dtype=np.uint8).tobytes()
    for n in list(range(0, 300)) + [1023, 1024, 2999, 3000]:
        assert o.xxh64(data[:n], 0) == xxhash.xxh64(data[:n], seed=0).intdigest()


# ---- LZ4 error KAT: T/lz4/TestLz4.java:53-60 ----
def test_lz4_offset_kat(o):
    data = bytes([15, 0, 0, 255, 255, 0x8A, 49, 255, 255, 0])
    # Java: new byte[] {15, 0, 0, -1, -1, -118, 49, -1, -1, 0}
    with pytest.raises(OracleError) as e:
        o.decompress("lz4", data, 1024)
    assert e.value.detail == 5 and e.value.offset == 3  # "offset outside destination buffer: offset=3"


def test_lz4_empty_cases(o):
    with pytest.raises(OracleError) as e:
        o.decompress("lz4", b"", 10)
    assert e.value.detail == 1
    assert o.decompress("lz4", b"\x00", 0) == b""
    with pytest.raises(OracleError) as e:  # Java returns -1 here
        o.decompress("lz4", b"\x10a", 0)
    assert e.value.detail == 7


# T/lz4/AbstractTestLz4.java:28-67: length fields that run off the input must raise
def test_lz4_literal_length_overflow(o):
    data = bytes([0xF0]) + b"\xff" * 4000
    with pytest.raises(OracleError):
        o.decompress("lz4", data, 1 << 20)
    data = bytes([0x0F, ord("a"), 1, 0][0:1]) + b"a" * 0
    data = bytes([0x1F, ord("a"), 1, 0]) + b"\xff" * 4000
    with pytest.raises(OracleError):
        o.decompress("lz4", data, 1 << 22)


# ---- Snappy error KATs: T/snappy/TestSnappyJava.java:52-59, T/snappy/AbstractTestSnappy.java:31-57 ----
def test_snappy_offset_kat(o):
    with pytest.raises(OracleError) as e:
        o.decompress("snappy", bytes([16, 1, 0, 1, 0, 1, 0, 1, 0]), 1024)
    assert e.value.detail == 16 and e.value.offset == 2  # "Malformed input: offset=2"


def test_snappy_invalid_varint(o):
    with pytest.raises(OracleError) as e:  # negative length
        o.decompress("snappy", bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F]) + b"\x00", 10)
    assert e.value.detail == 19  # "invalid compressed length"
    with pytest.raises(OracleError) as e:
        o.decompress("snappy", bytes([0xFF, 0xFF, 0xFF, 0xFF, 0xFF]), 10)
    assert e.value.detail == 18
    with pytest.raises(OracleError) as e:
        o.decompress("snappy", bytes([0x80]), 10)
    assert e.value.detail == 17
    with pytest.raises(OracleError) as e:  # literal longer than the input
        o.decompress("snappy", bytes([10, 0xFC, 0xFF, 0xFF, 0xFF, 0x7F]) + b"abc", 100)
    assert e.value.cls == 1


def test_snappy_output_too_small(o):
    c = o.compress("snappy", b"hello world! hello world! hello world!")
    with pytest.raises(OracleError) as e:
        o.decompress("snappy", c, 10)
    assert e.value.cls == 2 and e.value.detail == 21


# ---- size formulas ----
def test_max_compressed_length(o):
    assert o.max_compressed_length("lz4", 65536) == 65_809
    assert o.max_compressed_length("snappy", 65536) == 76_490
    assert o.max_compressed_length("lz4", 0) == 16
    assert o.max_compressed_length("snappy", 0) == 32


def test_compress_buffer_too_small(o):
    with pytest.raises(OracleError) as e:
        o.compress("lz4", b"x" * 100, cap=o.max_compressed_length("lz4", 100) - 1)
    assert e.value.cls == 2
    with pytest.raises(OracleError) as e:
        o.compress("snappy", b"x" * 100, cap=o.max_compressed_length("snappy", 100) - 1)
    assert e.value.cls == 2


# ---- round trips on the harness' inputs: T/AbstractTestCompression.java:47-56,370-393,617-648 ----
@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_round_trip_hand_cases(o, codec):
    for name, data in common.HAND_CASES:
        c = o.compress(codec, data)
        assert len(c) <= o.max_compressed_length(codec, len(data))
        if codec == "lz4" and len(data) == 0:
            assert c == b"\x00"
            assert o.decompress(codec, c, 0) == b""
        else:
            assert o.decompress(codec, c, len(data)) == data, name
            assert o.decompress(codec, c, len(data) + 100) == data, name


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_round_trip_every_prefix(o, codec):
    base = common.corpus_sample()[0][1]
    for n in range(1, 256):
        d = base[:n]
        assert o.decompress(codec, o.compress(codec, d), n) == d


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_round_trip_corpus_sample_and_pinned_hashes(o, codec):
    import hashlib
    for name, data, entry in common.corpus_sample():
        c = o.compress(codec, data)
        assert hashlib.sha256(c).hexdigest() == entry[codec]["sha256"], name  # stable compressed bytes
        assert o.decompress(codec, c, len(data)) == data, name


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_round_trip_synthetic(o, codec):
    for i, data in enumerate(common.synthetic_blocks(11, 24)):
        c = o.compress(codec, data)
        if len(data) == 0 and codec == "lz4":
            continue
        assert o.decompress(codec, c, len(data)) == data, i


def test_snappy_multi_subblock(o):
    # > 64 KiB inputs are cut into independent 64 KiB sub-blocks (M/snappy/SnappyRawCompressor.java:93-99)
    data = b"".join(d for _, d, _ in common.corpus_sample()[:3]) + b"tail"
    c = o.compress("snappy", data)
    assert o.decompress("snappy", c, len(data)) == data


def test_lz4_large_single_block(o):
    data = b"".join(d for _, d, _ in common.corpus_sample()[:4])
    c = o.compress("lz4", data)
    assert o.decompress("lz4", c, len(data)) == data


# ---- the reference's synthetic generator: T/snappy/RandomGenerator.java:25-74 ----
def test_random_generator_structure(o):
    g = o.random_generator(0.5)
    assert len(g) == 1048576 + 100
    frag = g[:100]
    assert bytes(frag[:50]) == bytes(frag[50:100])
    assert not np.array_equal(g[:100], g[100:200])
    # java.util.Random(301): first nextInt(256) values (LCG restated independently here)
    seed = (301 ^ 0x5DEECE66D) & ((1 << 48) - 1)
    vals = []
    for _ in range(8):
        seed = (seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        vals.append(((256 * (seed >> 17)) >> 31) & 0xFF)
    assert list(g[:8]) == vals
    g1 = o.random_generator(0.1)
    assert bytes(g1[:10]) * 10 == bytes(g1[:100])


# ---- Zstd encoder (level 3): frame-header KATs T/zstd/TestCompressor.java:52-98 ----
def test_zstd_frame_header_kats(o):
    import ctypes
    o.lib.orc_zstd_write_frame_header.restype = ctypes.c_int32
    o.lib.orc_zstd_write_frame_header.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    o.lib.orc_zstd_read_frame_header.restype = ctypes.c_int32
    o.lib.orc_zstd_read_frame_header.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    big = 65536 + 256
    kats = [(1, 1024, (2, -1, 1)), (256, 1024, (3, -1, 256)), (big, 2048, (6, 2048, big)), (2**31 - 1, 1024, (6, 1024, 2**31 - 1))]
    kats += [(big, 1024 + 128 * k, (6, 1024 + 128 * k, big)) for k in range(1, 9)]
    for input_size, window, (hsize, wsize, csize) in kats:
        buf = np.zeros(14, dtype=np.uint8)
        n = o.lib.orc_zstd_write_frame_header(buf.ctypes.data, input_size, window)
        assert n == hsize, (input_size, window, n)
        out4 = np.zeros(4, dtype=np.int64)
        assert o.lib.orc_zstd_read_frame_header(buf.ctypes.data, 14, out4.ctypes.data) == 0
        assert tuple(out4) == (hsize, wsize, csize, 1), (input_size, window, tuple(out4))
    buf = np.zeros(14, dtype=np.uint8)
    assert o.lib.orc_zstd_write_frame_header(buf.ctypes.data, 2000, 1023) == -1  # "Minimum window size is 1024"
    assert o.lib.orc_zstd_write_frame_header(buf.ctypes.data, 2000, 1025) == -2  # "... must be multiple of 128"


def test_zstd_encoder_round_trips_through_own_decoder(o):
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()] + common.synthetic_blocks(17, 18)
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()[:5]))  # multi-block frame (5 x 64 KiB)
    blocks.append(common.golden_zstd("large-rle"))
    blocks.append(common.golden_zstd("incompressible"))
    for i, d in enumerate(blocks):
        z = o.compress("zstd", d, o.max_compressed_length("zstd", len(d)))
        assert len(z) <= o.max_compressed_length("zstd", len(d))
        assert z[:4] == bytes([0x28, 0xB5, 0x2F, 0xFD])
        assert o.decompress("zstd", z, len(d)) == d, i
    with pytest.raises(OracleError) as e:  # T/zstd/TestCompressor.java:33-41
        o.compress("zstd", b"x" * 100, cap=3)
    assert e.value.cls == 2


def test_zstd_encoder_cross_decodes_with_libzstd_when_available(o):
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("zstd")
    for _, d, _ in common.corpus_sample()[:8]:
        z = o.compress("zstd", d, o.max_compressed_length("zstd", len(d)))
        assert codec.decompress(z, decompressed_size=len(d)).to_pybytes() == d
