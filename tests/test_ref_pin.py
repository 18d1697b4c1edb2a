"""The pin of the oracle against the REFERENCE'S OWN CODE (round 3): oracle/_ref/libref.so is the reference's Java sources -- read from
/root/reference where they lie -- turned into C++ token by token by tools/j2c.py (rules in its docstring, the hand patches in
oracle/ref/patches.txt: a lambda, a constructor delegation, two try/finally, the java.lang.foreign overloads; none touches codec
arithmetic) and compiled with oracle/ref/jrt.h (Java's integer semantics, arrays, Unsafe).  No JVM exists here; this is the reference's
encoder, decoder and stream classes EXECUTING, statement for statement.

What is compared:
  * compress side (what round 1 and 2 could not pin): Lz4RawCompressor / SnappyRawCompressor / ZstdFrameCompressor at level 3 over all 42
    corpus files whole and every 64 / 128 KiB cut == tests/golden/oracle_manifest.tsv (668 lines), ZstdOutputStream (write + close, incl.
    the 14 MB stream with window slides) == tests/golden/oracle_stream_manifest.tsv, Lz4HadoopOutputStream / SnappyHadoopOutputStream ==
    oracle/hadoop_streams.c -- byte for byte;
  * decode side (pinned by the reference's vectors before; now also differentially): Lz4RawDecompressor / SnappyRawDecompressor /
    ZstdFrameDecompressor and the two Hadoop input streams against the oracle on valid, truncated, bit-flipped and extended streams:
    same plaintext, same exception kind, same MalformedInputException offset, same message.

The library is built here when /root/reference exists (`make -C oracle/ref`), and travels prebuilt otherwise (oracle/_ref/ is
git-ignored, not gpurun-ignored); with neither the tests skip.  Nothing under aircompressor_amd/ knows of it."""
import ctypes
import hashlib
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import common
from tests.oracle_lib import OracleError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "ref")
LIB = os.path.join(ROOT, "oracle", "_ref", "libref.so")
REFERENCE = "/root/reference/src/main/java/io/airlift/compress/v3"

i64 = ctypes.c_int64
vp = ctypes.c_void_p


class Ref:
    """ctypes view of libref.so; results as (code, bytes, offset, message): code >= 0 bytes written, -1 MalformedInputException, -2
    IllegalArgumentException, -3 another RuntimeException (message tells), -4 IOException, -5 the Java method returned a negative value"""

    def __init__(self, lib):
        self.lib = lib
        for c in ("lz4", "snappy", "zstd"):
            f = getattr(lib, "ref_%s_compress" % c)
            f.restype, f.argtypes = i64, [vp, i64, vp, i64]
            f = getattr(lib, "ref_%s_decompress" % c)
            f.restype, f.argtypes = i64, [vp, i64, vp, i64, ctypes.POINTER(i64)]
            f = getattr(lib, "ref_%s_max_compressed_length" % c)
            f.restype, f.argtypes = i64, [i64]
        lib.ref_zstd_stream_compress.restype, lib.ref_zstd_stream_compress.argtypes = i64, [vp, i64, vp, i64]
        lib.ref_zstd_stream_decompress.restype, lib.ref_zstd_stream_decompress.argtypes = i64, [vp, i64, vp, i64, ctypes.POINTER(i64)]
        lib.ref_hadoop_compress.restype, lib.ref_hadoop_compress.argtypes = i64, [ctypes.c_int32, vp, i64, vp, i64, ctypes.c_int32]
        lib.ref_hadoop_decompress.restype, lib.ref_hadoop_decompress.argtypes = i64, [ctypes.c_int32, vp, i64, vp, i64, ctypes.c_int32, ctypes.POINTER(i64)]
        lib.ref_xxh64.restype, lib.ref_xxh64.argtypes = ctypes.c_uint64, [vp, i64, ctypes.c_uint64]
        lib.ref_snappy_uncompressed_length.restype, lib.ref_snappy_uncompressed_length.argtypes = i64, [vp, i64, ctypes.POINTER(i64)]
        lib.ref_zstd_decompressed_size.restype, lib.ref_zstd_decompressed_size.argtypes = i64, [vp, i64, ctypes.POINTER(i64)]
        for n in ("ref_last_error", "ref_zstd_last_error", "ref_streams_last_error"):
            getattr(lib, n).restype = ctypes.c_char_p

    def compress(self, codec, data, cap=None):
        cap = getattr(self.lib, "ref_%s_max_compressed_length" % codec)(len(data)) if cap is None else cap
        out = ctypes.create_string_buffer(max(cap, 1))
        r = getattr(self.lib, "ref_%s_compress" % codec)(data, len(data), out, cap)
        assert r >= 0, (codec, r, self.message(codec))
        return out.raw[:r]

    def message(self, codec):
        return (self.lib.ref_zstd_last_error() if codec == "zstd" else self.lib.ref_last_error()).decode()

    def decompress(self, codec, data, cap):
        src = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
        out = ctypes.create_string_buffer(max(cap, 1))
        eo = i64(0)
        r = getattr(self.lib, "ref_%s_decompress" % codec)(src, len(data), out, cap, ctypes.byref(eo))
        off = eo.value
        if codec == "zstd" and r == -1:
            # ZstdFrameDecompressor reports ADDRESSES (`verify(cond, input, ..)` with `input` the Unsafe address: 16 + index for a byte[],
            # the segment's address for native memory -- M/zstd/ZstdFrameDecompressor.java:150-214); the oracle and the ABI report the index
            off -= ctypes.addressof(src)
        return r, out.raw[:max(r, 0)], off, self.message(codec)

    def zstd_stream_compress(self, data):
        cap = len(data) + len(data) // 64 + 4096
        out = ctypes.create_string_buffer(cap)
        r = self.lib.ref_zstd_stream_compress(data, len(data), out, cap)
        assert r >= 0, (r, self.lib.ref_zstd_last_error())
        return out.raw[:r]

    def hadoop_compress(self, codec, data, buffer_size):
        cap = len(data) + len(data) // 4 + 65536
        out = ctypes.create_string_buffer(cap)
        r = self.lib.ref_hadoop_compress(0 if codec == "lz4" else 1, data, len(data), out, cap, buffer_size)
        assert r >= 0, (r, self.lib.ref_streams_last_error())
        return out.raw[:r]

    def hadoop_decompress(self, codec, data, cap, buffer_size):
        out = ctypes.create_string_buffer(max(cap, 1))
        eo = i64(0)
        r = self.lib.ref_hadoop_decompress(0 if codec == "lz4" else 1, bytes(data), len(data), out, cap, buffer_size, ctypes.byref(eo))
        return r, out.raw[:max(r, 0)], eo.value, self.lib.ref_streams_last_error().decode()


@pytest.fixture(scope="module")
def ref():
    if os.path.isdir(REFERENCE):
        subprocess.run(["make", "-s", "-C", REF_DIR], check=True)
    if not os.path.exists(LIB):
        pytest.skip("oracle/_ref/libref.so absent and /root/reference not here to build it from")
    return Ref(ctypes.CDLL(LIB))


@pytest.fixture(scope="module")
def detail_text():
    """detail id -> the Java message it stands for (the product's table: achip_detail_message; a lookup, no GPU involved)"""
    from aircompressor_amd import native
    lib = native.load_library()
    lib.achip_detail_message.restype = ctypes.c_char_p
    lib.achip_detail_message.argtypes = [ctypes.c_int32]
    return lambda d: lib.achip_detail_message(d).decode()


def test_generated_sources_stay_out_of_the_repository():
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.split()
    assert tracked == []  # generated from reference text: never committed
    # the hand patches are plumbing and few: every patched Java line range is listed in the patch file with its reason
    ranges = [l for l in open(os.path.join(REF_DIR, "patches.txt")) if l.startswith("@@ ")]
    lines = 0
    for l in ranges:
        a, _, b = l.split()[2].partition("-")
        lines += int(b or a) - int(a) + 1
        assert "#" in l, "a patch without a reason: " + l
    assert len(ranges) <= 20 and lines <= 300, (len(ranges), lines)


def test_block_encoders_equal_the_oracle_manifest(ref):
    """a2 / a4 / a11-a14: the reference's encoders, executing, produce the streams the oracle produced (and the GPU produces: tests/test_gpu_corpus.py
    compares the GPU with the same manifest)"""
    rows = common.read_manifest_tsv("oracle_manifest.tsv")
    corpus = common.corpus_full()
    seen = {"lz4": 0, "snappy": 0, "zstd": 0}
    for file, off, length, codec, clen, sha in rows:
        c = ref.compress(codec, corpus[file][off:off + length])
        assert len(c) == clen and hashlib.sha256(c).hexdigest() == sha, (file, off, length, codec)
        seen[codec] += 1
    assert min(seen.values()) > 140 and sum(seen.values()) == len(rows) > 600


def test_block_encoders_equal_the_oracle_on_shaped_inputs(ref, oracle):
    """beyond the corpus: the hand cases, every prefix of a text up to 300 bytes (all the short-input paths), synthetic shapes (runs, noise, periodic
    data around the 64 KiB / 128 KiB / 256 KiB parameter boundaries), the multi-block shapes of the Zstd tests"""
    rng = np.random.default_rng(17)
    base = common.corpus_sample()[0][1]
    inputs = [d for _, d in common.HAND_CASES] + [base[:n] for n in range(0, 300)]
    inputs += common.synthetic_blocks(9, 12)
    for n in (65535, 65536, 65537, 131071, 131072, 131073, 262144, 262145, 300000):
        inputs.append((base * (n // len(base) + 1))[:n])
        inputs.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        inputs.append(bytes(n))
    inputs += [p for p in common.multi_block_plains() if len(p) <= 1 << 20]
    for k, b in enumerate(inputs):
        for codec in ("lz4", "snappy", "zstd"):
            assert ref.compress(codec, b) == oracle.compress(codec, b), (k, len(b), codec)


def test_stream_writer_equals_the_oracle_stream_manifest(ref):
    """f3 writer: new ZstdOutputStream(sink); write(buffer, 0, n); close() -- every corpus file, and the whole corpus as one 14 MB stream (chunks
    flushed before close(), the window slid six times, the blocks that find no match behind every slide)"""
    rows = common.read_manifest_tsv("oracle_stream_manifest.tsv")
    corpus = common.corpus_full()
    order = [e["file"] for e in json.load(open(os.path.join(common.GOLDEN, "corpus_full.json")))]
    for file, off, length, codec, clen, sha in rows:
        data = corpus[file] if file != "*" else b"".join(corpus[f] for f in order)
        c = ref.zstd_stream_compress(data)
        assert len(c) == clen and hashlib.sha256(c).hexdigest() == sha, file
    assert len(rows) == len(corpus) + 1


def test_stream_writer_equals_the_oracle_around_the_chunking_sizes(ref, oracle):
    rng = np.random.default_rng(23)
    text = b"".join(d for _, d, _ in common.corpus_sample())
    for n in (0, 1, 1000, 131072, 131073, 524288, 524289, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, 6 << 20):
        data = (text * (n // len(text) + 1))[:n]
        assert ref.zstd_stream_compress(data) == oracle.zstd_stream_compress(data), n
    noise = rng.integers(0, 256, 5 << 20, dtype=np.uint8).tobytes()
    assert ref.zstd_stream_compress(noise) == oracle.zstd_stream_compress(noise)


def mutations(rng, good, rounds):
    yield bytes(good)
    for k in range(rounds):
        c = bytearray(good)
        kind = k % 4
        if kind == 0 and len(c) > 1:
            c = c[:int(rng.integers(0, len(c)))]
        elif kind == 1 and len(c):
            for _ in range(int(rng.integers(1, 4))):
                c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2 and len(c):
            c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
        else:
            c += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
        yield bytes(c)


# Where the reference throws something that is NOT a MalformedInputException on corrupt input -- an ArrayIndexOutOfBoundsException from a
# table lookup with a corrupt index (e.g. a negative RLE symbol: M/zstd/ZstdFrameDecompressor.java:612-617 stores it, :410-413 index with it)
# -- the oracle (and the ABI, which has no "ArrayIndexOutOfBounds" status) reports malformed input with detail "Input is corrupted" at the
# place the corrupt value is read.  Those cases are counted, not compared field by field.
ZSTD_CORRUPTED = 34


@pytest.mark.parametrize("codec", ["lz4", "snappy", "zstd"])
def test_decoders_agree_on_valid_and_damaged_streams(ref, oracle, detail_text, codec):
    rng = np.random.default_rng({"lz4": 3, "snappy": 5, "zstd": 7}[codec])
    plains = [d for _, d in common.HAND_CASES] + [d[:int(rng.integers(50, 6000))] for _, d, _ in common.corpus_sample()]
    plains += [b[:int(rng.integers(100, 3000))] for b in common.synthetic_blocks(4, 5)]
    compared = errors = lenient = 0
    for b in plains:
        good = oracle.compress(codec, b)
        for c in mutations(rng, good, 24):
            cap = max(len(b) + (int(rng.integers(-2, 20)) if rng.integers(0, 3) else 0), 0)
            try:
                expect = oracle.decompress(codec, c, cap)
                e = None
            except OracleError as err:
                expect, e = None, err
            r, out, off, msg = ref.decompress(codec, c, cap)
            compared += 1
            if e is None:
                assert r >= 0 and out == expect, (codec, len(b), r, msg)
                continue
            errors += 1
            if r == -5:  # Lz4RawDecompressor.java:52-57: an empty output buffer makes the method return -1
                assert codec == "lz4" and cap == 0 and detail_text(e.detail) == "Output buffer too small"
            elif r == -3 or (codec == "zstd" and e.detail == ZSTD_CORRUPTED and not msg.startswith("Input is corrupted")):
                assert codec == "zstd" and e.cls == 1 and e.detail == ZSTD_CORRUPTED, (r, msg, e)
                lenient += 1
            elif r == -2:
                assert e.cls == 2, (msg, e)
            else:
                assert r == -1 and e.cls == 1, (r, msg, e)
                assert off == e.offset, (codec, msg, off, e)
                # "<reason>: offset=<n>" -- the reason's fixed part is the detail's text
                assert msg.endswith(": offset=%d" % (off if codec != "zstd" else off + 0)) or codec == "zstd"
                assert msg[:12] == detail_text(e.detail)[:12], (msg, detail_text(e.detail))
    assert compared > 800 and errors > 300
    assert lenient * 50 < errors  # the non-Malformed exceptions of the reference are rare corners


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_hadoop_block_streams_equal_the_oracle(ref, oracle, detail_text, codec):
    """f2 (Hadoop half): the reference's own stream classes -- Lz4HadoopOutputStream / InputStream, SnappyHadoopOutputStream / InputStream over
    Lz4JavaCompressor etc., driven as T/HadoopCodecCompressor.java:57-72 and T/HadoopCodecDecompressor.java:37-60 drive them -- against
    oracle/hadoop_streams.c: the writer byte for byte at three buffer sizes, the reader on the streams of every branch the oracle's own tests
    build and on random damage"""
    rng = np.random.default_rng(41 if codec == "lz4" else 43)
    data = b"".join(d for _, d, _ in common.corpus_sample())[:700000]
    for buf in (262144, 4096, 70000):
        for d in (data, data[:1], b"", data[:buf - buf // 100 - 1], data[:300000]):
            s = ref.hadoop_compress(codec, d, buf)
            assert s == oracle.hadoop_compress(codec, d, buf), (buf, len(d))
            r, out, _, msg = ref.hadoop_decompress(codec, s, len(d), buf)
            assert r == len(d) and out == d, (r, msg)

    def be(v):
        return struct.pack(">i", v)

    def stream(pieces):
        out = b""
        for declared, plain in pieces:
            c = oracle.compress(codec, plain)
            out += (be(declared) if declared is not None else b"") + be(len(c)) + c
        return out

    a, b, c = b"hello hello hello hello " * 40, b"abcdefgh" * 300, bytes(range(256)) * 3
    good = stream([(len(a), a)])
    cases = []
    s = stream([(len(a) + len(b), a), (None, b)]) + be(0) + be(0) + stream([(len(c), c)]) + be(0)
    cases += [(s, cap) for cap in (len(a + b + c), len(a + b + c) + 1000, len(a + b + c) - 1, len(a) + 5, len(a), 10, 0)]
    cases += [(good + be(-1), len(a) + 10), (good + be(50) + be(-1), len(a) + 10), (good + be(-1) + b"garbage", len(a) + 10), (good + be(77), len(a))]
    cases += [(good[:cut], len(a)) for cut in (2, 6, len(good) - 1, 9)]
    cases += [(be(10) + be(-5) + b"xxxxx", 100)]
    cases += [(stream([(len(a) - 1, a)]), len(a)), (stream([(len(a), a)]) + be(5) + be(1) + b"\x00", 3 * len(a)), (stream([(5, a), (None, b)]), len(a + b)),
              (stream([(len(a) + 100, a)]), len(a) + 50)]
    real = oracle.hadoop_compress(codec, data[:20000], 4096)
    for m in mutations(rng, real, 120):
        cases.append((m, 20000 + int(rng.integers(-3, 30))))
    agree = deviations = 0
    for s, cap in cases:
        for buf in (262144, 4096) if len(s) < 3000 else (4096,):
            try:
                expect = oracle.hadoop_decompress(codec, s, cap, buf)
                e = None
            except OracleError as err:
                expect, e = None, err
            r, out, off, msg = ref.hadoop_decompress(codec, s, cap, buf)
            if e is None:
                assert r >= 0 and out == expect, (len(s), cap, buf, r, msg)
            else:
                assert r < 0, (len(s), cap, buf, e)
                text = detail_text(e.detail)
                if e.detail == 109:
                    # documented deviation of oracle/hadoop_streams.c and the kernels, SNAPPY reader only, for a field no writer produces: a
                    # negative chunk length other than -1 is reported as such.  SnappyHadoopInputStream.java:110-133 goes on to parse
                    # whatever its buffer still holds from the chunk before, then fails in the block codec's range check: an exception
                    # follows either way, of a kind that depends on stale state.  (LZ4: any negative length ends the stream -- exact.)
                    assert codec == "snappy" and r < 0, (r, msg)
                    deviations += 1
                elif r == -1:
                    assert e.cls == 1 and off == e.offset and msg[:12] == text[:12], (msg, e, text)
                else:
                    assert msg[:12] == text[:12] or (r == -2 and e.cls == 2), (r, msg, e, text)
            agree += 1
    assert agree > 150 and deviations * 20 < agree


def test_xxh64_and_size_probes(ref, oracle):
    rng = np.random.default_rng(9)
    for n in list(range(0, 100)) + [255, 256, 257, 4095, 4096, 65536, 100003]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 0x9E3779B185EBCA87):
            assert ref.lib.ref_xxh64(d, n, seed) == oracle.xxh64(d, seed)
    text = common.corpus_sample()[0][1]
    eo = i64(0)
    for n in (0, 1, 127, 128, len(text)):
        c = oracle.compress("snappy", text[:n])
        assert ref.lib.ref_snappy_uncompressed_length(c, len(c), ctypes.byref(eo)) == n
        z = oracle.compress("zstd", text[:n])
        assert ref.lib.ref_zstd_decompressed_size(z, len(z), ctypes.byref(eo)) == n


def test_input_stream_reads_what_the_one_shot_decoder_reads(ref, oracle):
    """f3 reader: the reference's ZstdInputStream (over ZstdIncrementalFrameDecompressor: input in pieces, a window buffer, frames without a
    content size) read to the end delivers what the oracle's one-shot decoder -- the GPU decoders' checker -- delivers for the same bytes,
    given the capacity achip_zstd_decompress_bound computes: the writer's streams below and beyond 4 MiB, libzstd-shaped multi-block
    frames, concatenated frames"""
    from aircompressor_amd import native
    lib = native.load_library()
    text = b"".join(d for _, d, _ in common.corpus_sample())
    tiled = text * 16
    plains = [b"", b"q", text[:1000], text[:300000], tiled[:(4 << 20) + 5], tiled[7:6400007]] + [p for p in common.multi_block_plains() if len(p) <= 1 << 20][:6]
    streams = [oracle.zstd_stream_compress(p) for p in plains] + [oracle.compress("zstd", p) for p in plains[1:4]]
    streams.append(streams[3] + streams[4] + streams[2])
    for z in streams:
        eo = i64(0)
        a = np.frombuffer(z, dtype=np.uint8)
        bound = lib.achip_zstd_decompress_bound(a.ctypes.data, len(a), ctypes.byref(eo))
        assert bound >= 0
        want = oracle.decompress("zstd", z, bound)
        out = ctypes.create_string_buffer(max(bound, 1))
        r = ref.lib.ref_zstd_stream_decompress(z, len(z), out, bound, ctypes.byref(eo))
        assert r == len(want) and out.raw[:r] == want, (len(z), r, len(want), ref.lib.ref_zstd_last_error())


def test_input_stream_corners_the_gpu_reader_is_held_to(ref, oracle):
    """What the reference's ZstdInputStream does where tools/fuzz_zstd_stream.py found the GPU reader doing something else (round 5), pinned here against the
    reference's own code so that tests/test_gpu_zstd_stream.py::test_reader_corners_the_stream_fuzzer_found rests on a fact, not on a reading of the source:
    up to three bytes behind the last frame end the stream quietly, four are an invalid magic, fewer than four in front of the FIRST frame are an IOException;
    RAW and RLE blocks that say more than 128 KiB are decoded for what they say."""
    ref.lib.ref_zstd_stream_decompress_partial.restype = i64
    ref.lib.ref_zstd_stream_decompress_partial.argtypes = [vp, i64, vp, i64, ctypes.c_int32, ctypes.POINTER(i64), ctypes.POINTER(i64)]

    def read(stream, cap):
        out = ctypes.create_string_buffer(cap + 1)
        delivered, eo = i64(0), i64(0)
        src = ctypes.create_string_buffer(bytes(stream), len(stream))
        r = ref.lib.ref_zstd_stream_decompress_partial(src, len(stream), out, cap + 1, 65536, ctypes.byref(delivered), ctypes.byref(eo))
        return r, (out.raw[:r] if r >= 0 else out.raw[:delivered.value])

    text = b"".join(d for _, d, _ in common.corpus_sample())
    z = oracle.zstd_stream_compress(text[:300000])
    for tail in (b"\x01", b"\x28\xb5", b"\x28\xb5\x2f"):
        r, got = read(z + tail, 400000)
        assert r == 300000 and got == text[:300000], (tail, r)
    r, got = read(z + b"\x01\x02\x03\x04", 400000)
    assert r == -1 and got == text[:len(got)]  # MalformedInputException ("Invalid magic prefix"); what it delivered before is plaintext
    assert read(b"\x28\xb5", 100)[0] == -4     # IOException ("Not enough input bytes"): INITIAL is not a stopping point
    assert read(b"", 100)[0] == -4

    def block(kind, size, payload, last=0):
        return int((size << 3) | (kind << 1) | last).to_bytes(3, "little") + payload
    raw1, raw2 = text[1000:151000], b"tail!"
    frame = b"\x28\xb5\x2f\xfd" + b"\x00" + b"\x50" + block(1, 200000, b"q") + block(0, len(raw1), raw1) + block(1, 131073, b"\x00") + block(0, len(raw2), raw2, 1)
    plain = b"q" * 200000 + raw1 + bytes(131073) + raw2
    r, got = read(z + frame + oracle.zstd_stream_compress(text[:5000]), 2000000)
    assert r == 300000 + len(plain) + 5000 and got == text[:300000] + plain + text[:5000]
