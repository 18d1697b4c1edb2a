"""The C-ABI shared library: loads, exports every symbol include/*.h declares, host-only helpers
agree with the oracle, and the codecs fail loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from tests import common, oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_library()
    import aircompressor_amd as A
    return A.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "aircompressor_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(achip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 40
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "aircompressor_amd", "libaircompressor_hip.so")],
                         check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (achip_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    from aircompressor_amd import native
    assert sorted(native.SIGNATURES) == names  # the Python binding types exactly the declared surface


def test_no_torch_or_oracle_in_product():
    # the product library depends on libamdhip64 only; the package never touches oracle/
    out = subprocess.run(["ldd", os.path.join(ROOT, "aircompressor_amd", "libaircompressor_hip.so")], capture_output=True, text=True).stdout
    assert "torch" not in out and "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aircompressor_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "errors.py", (dirpath, f)
                if f != "sharding.py":  # torch.distributed control plane only (barrier / max of times); never on the data path
                    assert "import torch" not in text, (dirpath, f)


def test_size_helpers_match_reference_kats(lib):
    # T/zstd/AbstractTestZstd.java:140-147 + SURVEY Appendix B
    assert lib.achip_zstd_max_compressed_length(0) == 64
    assert lib.achip_zstd_max_compressed_length(64 * 1024) == 65_824
    assert lib.achip_zstd_max_compressed_length(128 * 1024) == 131_584
    assert lib.achip_zstd_max_compressed_length(128 * 1024 + 1) == 131_585
    assert lib.achip_lz4_max_compressed_length(65536) == 65_809
    assert lib.achip_snappy_max_compressed_length(65536) == 76_490
    o = oracle_lib.load()
    for n in (0, 1, 5, 254, 255, 256, 65535, 65536, 131072, 1 << 20, 0x7E000000):
        for codec in ("lz4", "snappy", "zstd"):
            got = getattr(lib, "achip_%s_max_compressed_length" % codec)(n)
            assert got == ctypes.c_int32(o.max_compressed_length(codec, n)).value


def test_hadoop_stream_bound_and_messages(lib):
    # the capacity the one-shot Hadoop stream writers ask for (per chunk of bufferSize - overhead bytes: two big-endian ints + the codec's
    # bound; M/lz4/Lz4HadoopOutputStream.java:44-46,107-131) against the oracle's, and the stream classes' exception texts
    o = oracle_lib.load()
    for codec_id, codec in ((0, "lz4"), (1, "snappy")):
        for buf in (262144, 70000, 1024, 64):
            for n in (0, 1, 900, 1014, 1015, 218422, 259523, 259524, 700000, 1 << 22):
                assert lib.achip_hadoop_max_compressed_length(codec_id, n, buf) == o.hadoop_max_compressed_length(codec, n, buf), (codec, buf, n)
    assert lib.achip_hadoop_max_compressed_length(0, -1, 262144) < 0 and lib.achip_hadoop_max_compressed_length(2, 10, 262144) < 0
    assert lib.achip_hadoop_max_compressed_length(1, 10, 36) < 0  # (Snappy's overhead eats the whole buffer)
    assert lib.achip_detail_message(104) == b"Stream is truncated"
    assert lib.achip_detail_message(105) == b"encountered EOF while reading block data"
    assert lib.achip_detail_message(106) == b"Chunk uncompressed size is greater than block size"
    assert lib.achip_detail_message(108) == b"All input was not consumed"


def test_zstd_stream_bound(lib):
    # achip_zstdstream_max_compressed_length against the oracle's, and against what the oracle's stream writer actually needs for
    # incompressible input (every block stored: 3 bytes per 128 KiB, the longest header, the checksum)
    o = oracle_lib.load()
    rng = np.random.default_rng(2)
    for n in (0, 1, 255, 256, 65791, 65792, 131072, 131073, 1 << 20, (1 << 20) + 1, (4 << 20) - 1, 4 << 20, 9999999):
        assert lib.achip_zstdstream_max_compressed_length(n) == o.lib.orc_zstd_stream_max_compressed_length(n), n
    for n in (0, 1, 300, 131072, 131073, 400000, (1 << 20) + 5):
        noise = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert len(o.zstd_stream_compress(noise)) <= lib.achip_zstdstream_max_compressed_length(n), n
    assert lib.achip_zstdstream_max_compressed_length(-1) < 0 and lib.achip_zstdstream_max_compressed_length(0x7FFFFFFF) < 0


def test_status_helpers(lib):
    st = -(1 + 16 * 5)
    assert lib.achip_status_class(st) == 1 and lib.achip_status_detail(st) == 5
    assert lib.achip_detail_message(5) == b"offset outside destination buffer"
    assert lib.achip_status_class(12) == 0
    assert lib.achip_version().startswith(b"aircompressor-hip")


def test_snappy_uncompressed_length(lib):
    o = oracle_lib.load()
    for data in (b"", b"x", b"hello" * 1000, bytes(70000)):
        c = np.frombuffer(o.compress("snappy", data), dtype=np.uint8)
        eo = ctypes.c_int64()
        assert lib.achip_snappy_uncompressed_length(c.ctypes.data, len(c), ctypes.byref(eo)) == len(data)
    bad = np.frombuffer(bytes([0xFF] * 5), dtype=np.uint8)
    eo = ctypes.c_int64()
    r = lib.achip_snappy_uncompressed_length(bad.ctypes.data, 5, ctypes.byref(eo))
    assert lib.achip_status_detail(r) == 18
    r = lib.achip_snappy_uncompressed_length(bad.ctypes.data, 1, ctypes.byref(eo))
    assert lib.achip_status_detail(r) == 17


def test_zstd_decompressed_size(lib):
    o = oracle_lib.load()
    for name in ("with-checksum.zst", "multiple-frames.zst", "offset-before-start.zst"):
        z = np.frombuffer(common.golden_zstd(name), dtype=np.uint8)
        eo = ctypes.c_int64()
        eo2 = ctypes.c_int64()
        assert lib.achip_zstd_decompressed_size(z.ctypes.data, len(z), ctypes.byref(eo)) == \
            o.lib.orc_zstd_decompressed_size(z.ctypes.data, len(z), ctypes.byref(eo2))
    junk = np.frombuffer(b"\x00\x01\x02\x03\x04\x05", dtype=np.uint8)
    eo = ctypes.c_int64()
    r = lib.achip_zstd_decompressed_size(junk.ctypes.data, 6, ctypes.byref(eo))
    assert lib.achip_status_detail(r) == 35


def test_zstd_decompress_bound(lib):
    """achip_zstd_decompress_bound (host code): an upper bound of what the frames of a buffer decode to from frame and block headers alone --
    what the one-shot reader of streams WITHOUT a content size needs (ZstdOutputStream's from 4 MiB on).  Against the oracle's decoder: never
    below the decoded size, exact where the content size is known, less than one block above it for the writer's chunked streams; the
    reference's fixtures (two frames) and truncations."""
    o = oracle_lib.load()
    eo = ctypes.c_int64()

    def bound(z):
        a = np.frombuffer(bytes(z), dtype=np.uint8)
        return lib.achip_zstd_decompress_bound(a.ctypes.data if len(a) else None, len(a), ctypes.byref(eo))

    text = b"".join(d for _, d, _ in common.corpus_sample())
    for n in (0, 1, 1000, 131072, 131073, 300000, 700001):
        data = text[:n]
        for z in (o.compress("zstd", data), o.zstd_stream_compress(data)):
            assert bound(z) == n, n  # single-segment / content size present: exact
    tiled = (text * 16)[:(4 << 20) + 12345]
    z = o.zstd_stream_compress(tiled)
    assert lib.achip_zstd_decompressed_size(np.frombuffer(z, dtype=np.uint8).ctypes.data, len(z), ctypes.byref(eo)) == -1  # no content size
    b = bound(z)
    assert len(tiled) <= b < len(tiled) + 131072
    assert o.decompress("zstd", z, b) == tiled
    assert bound(z + o.compress("zstd", text[:5000])) == b + 5000  # frames add up
    for name, want, frames in (("with-checksum.zst", 11359, 1), ("multiple-frames.zst", 22718, 2)):  # (no content size in these: a block of 128 KiB per frame)
        assert bound(common.golden_zstd(name)) == 131072 * frames >= want
    assert bound(b"") == 0
    for cut in (3, 5, 9, len(z) // 2, len(z) - 1):
        r = bound(z[:cut])
        assert r < 0 and lib.achip_status_class(r) == 1 and lib.achip_status_detail(r) == 32, cut  # "Not enough input bytes"
    r = bound(b"\x00\x01\x02\x03\x04\x05")
    assert lib.achip_status_detail(r) == 35


def test_partition_blocks():
    import aircompressor_amd as A
    w = np.full(1000, 65536 + 30000, dtype=np.int64)
    for parts in (1, 2, 4, 8):
        s = A.partition_blocks(w, parts)
        assert s[0] == 0 and s[-1] == 1000 and all(s[i] <= s[i + 1] for i in range(parts))
        sizes = np.diff(s)
        assert sizes.max() - sizes.min() <= 1
    s = A.partition_blocks([1, 1, 1, 1, 10, 1, 1, 1], 2)
    assert list(s) == [0, 5, 8]
    assert list(A.partition_blocks([], 4)) == [0, 0, 0, 0, 0]


def test_multi_batch_host_argument_checks(lib):
    """achip_multi_batch_host without a device: the argument checks that come before any context is touched"""
    import ctypes
    z = np.zeros(4, dtype=np.int64)
    args = [0, None, None, z.ctypes.data, z.ctypes.data, None, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, z.ctypes.data, 1, None]
    assert lib.achip_status_class(lib.achip_multi_batch_host(None, 0, *args)) == 3                       # no contexts
    one_null = (ctypes.c_void_p * 2)(None, None)
    assert lib.achip_status_class(lib.achip_multi_batch_host(one_null, 2, *args)) == 3                    # a null context
    assert b"null" in lib.achip_last_error()
    assert lib.achip_status_class(lib.achip_multi_batch_host(one_null, 65, *args)) == 3


def test_codecs_fail_loudly_without_gpu(lib):
    import aircompressor_amd as A
    if lib.achip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    assert not A.Lz4HipCompressor.is_enabled()
    with pytest.raises(A.HipUnavailableError):
        A.Lz4HipCompressor()
    with pytest.raises(A.HipUnavailableError):
        A.HipBatchCodec()
    for cls in (A.Lz4HipDecompressor, A.SnappyHipCompressor, A.SnappyHipDecompressor, A.ZstdHipCompressor, A.ZstdHipDecompressor, A.Lz4FrameHipCompressor,
                A.Lz4FrameHipDecompressor, A.SnappyFramedHipCompressor, A.SnappyFramedHipDecompressor):
        with pytest.raises(A.HipUnavailableError):
            cls()


def test_java_sources_are_consistent_with_the_header_and_with_each_other():
    """No JDK in this image: the Java side cannot be compiled here.  What can be checked without javac is: every
    @NativeSignature names a symbol the header declares, with as many arguments as the declaration has parameters; every HipNative.member
    another class uses exists in HipNative.java (the advisor's round-1 finding was exactly such a drift); braces and parentheses balance."""
    import glob
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "aircompressor_hip.h")).read(), flags=re.S)
    batch_params = len([p for p in re.search(r"#define ACHIP_BATCH_ARGS\s+((?:.*\\\n)*.*)", header).group(1).replace("\\\n", " ").split(",") if p.strip()])
    native = open(os.path.join(ROOT, "java", "io", "airlift", "compress", "v3", "hip", "HipNative.java")).read()
    batch_decl = re.search(r"#define ACHIP_BATCH_ARGS\s+((?:.*\\\n)*.*)", header).group(1).replace("\\\n", " ")

    def java_type(c_decl):
        """the java.lang.foreign carrier NativeLoader can bind for a C parameter / return declaration (M/internal/NativeLoader.java:138-153)"""
        c = " ".join(c_decl.replace("const", " ").split())
        if "*" in c:
            return "MemorySegment"
        base = c.split()[0] if c else "void"
        return {"int32_t": "int", "int64_t": "long", "int8_t": "byte", "float": "float", "void": "void"}[base]

    bound = set()
    for name, ret, args in re.findall(r'@NativeSignature\(name = "([a-z0-9_]+)", returnType = ([A-Za-z.]+)\.class, argumentTypes = ([^)]*)\)', native):
        m = re.search(r"([A-Za-z0-9_ ]+?\**)\s*\b%s\(([^;]*)\);" % name, header)
        assert m, name
        bound.add(name)
        decl = m.group(2).strip().replace("ACHIP_BATCH_ARGS", batch_decl)
        want = [] if decl in ("", "void") else [java_type(part) for part in decl.split(",") if part.strip()]
        got = re.findall(r"\b([A-Za-z]+)\.class", args)
        assert got == want, (name, got, want)          # argument TYPES, in order: int / long / MemorySegment <-> int32 / int64 / pointer
        assert ret == java_type(m.group(1)), (name, ret, m.group(1))
    assert batch_params == len([p for p in batch_decl.split(",") if p.strip()])
    # every symbol the header declares is bound (round 6: the twelve helper symbols -- events, statistics, achip_partition_blocks ... -- included)
    assert bound == set(re.findall(r"\b(achip_[a-z0-9_]+)\s*\(", header)), sorted(set(re.findall(r"\b(achip_[a-z0-9_]+)\s*\(", header)) ^ bound)
    declared = set(re.findall(r"\b(?:public|static|final|private|protected)\s+(?:static\s+|final\s+)*[A-Za-z<>\[\].]+\s+([A-Za-z_][A-Za-z0-9_]*)\s*(?:\(|=|;)", native))
    declared |= set(re.findall(r"\b(?:class|record|enum|interface)\s+([A-Za-z_][A-Za-z0-9_]*)", native))
    for path in glob.glob(os.path.join(ROOT, "java", "**", "*.java"), recursive=True):
        text = open(path).read()
        code = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))  # (no comments, no string contents)
        assert code.count("{") == code.count("}") and code.count("(") == code.count(")"), path
        if path.endswith("HipNative.java"):
            continue
        for member in set(re.findall(r"\bHipNative\.([A-Za-z_][A-Za-z0-9_]*)", code)):
            assert member in declared, (os.path.basename(path), member)
