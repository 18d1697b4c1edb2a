"""GPU parity (through the C ABI) of the Zstd STREAM writer: achip_zstdstream_compress* must produce what
ZstdOutputStream (M/zstd/ZstdOutputStream.java:30-221) puts on its sink for write(buffer, 0, n) + close() -- here: what the oracle's
restatement produces (tests/test_oracle_zstd_stream.py pins that one against the Java source's implications): one chunk below 4 MiB, chunks
flushed before close() with window slides from there on (the last test).

Added at the very end of round 2, after the round's GPU minutes were spent: the kernel is the frame compressor's device code with the
stream's parameters (GPU-verified paths) plus the chunk loop; both forms are byte-identical with the oracle on the CPU emulator
(tools/hostemu/check_enc.py, access-granular lockstep), but these tests themselves first run on the driver's box."""
import io

import numpy as np
import pytest

from tests import common, oracle_lib

# (first GPU run of these kernels is the driver's: a test that does not come back is ended with its process -- which frees the device --
# instead of holding the box until the run's own limit)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]
OP_ZSTD_DECOMPRESS = 4
OP_ZSTDSTREAM_COMPRESS = 14


@pytest.fixture(scope="module")
def gb():
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0)


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def stream_inputs():
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(41)
    noise = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    tiled = (whole * 4)[:(4 << 20) - 1]
    sizes = (0, 1, 7, 100, 4096, 16384, 16385, 100000, 131072, 131073, 262144, 262145, 300000, 524288, 524289, 700001, 1 << 20, (1 << 20) + 1)
    return [whole[:n] for n in sizes] + [whole, tiled, tiled[:3000000], noise, noise[:100] + whole[:200000] + noise, b"\0" * 1000000, b"ab" * 400000]


def test_stream_writer_is_bit_exact_with_the_oracle(gb, o):
    inputs = stream_inputs()
    caps = [o.lib.orc_zstd_stream_max_compressed_length(len(b)) for b in inputs]
    outs, status, err = gb.run(OP_ZSTDSTREAM_COMPRESS, inputs, caps)
    for i, b in enumerate(inputs):
        assert status[i] == 0, (i, len(b), status[i])
        assert outs[i] == o.zstd_stream_compress(b), "input %d (len %d)" % (i, len(b))
    # ... which the frame compressor's output is beyond 512 KiB and is not below (other parameters)
    assert outs[15] == o.compress("zstd", inputs[15]) and outs[9] != o.compress("zstd", inputs[9])
    # and the GPU decoder reads them back (frames of up to 32 blocks: the pipeline's multi-block stages)
    back, status, err = gb.run(OP_ZSTD_DECOMPRESS, outs, [max(len(b), 1) for b in inputs])
    assert all(s == 0 for s in status), status
    assert [bytes(p) for p in back] == inputs


def test_the_sink_only_has_to_hold_the_stream(gb, o):
    """ZstdOutputStream compresses every block in its OWN buffer (ZstdOutputStream.java:55-58) and hands the sink the result: a caller's buffer of
    exactly the stream's size -- far below the advertised bound -- is enough, one byte less is "output too small" (round 2 passed the caller's
    remaining room to the block compressor and failed here: ADVICE)."""
    inputs = stream_inputs()[:14] + stream_inputs()[-3:]
    want = [o.zstd_stream_compress(b) for b in inputs]
    for extra in (0, 1, 20):
        outs, status, _ = gb.run(OP_ZSTDSTREAM_COMPRESS, inputs, [len(w) + extra for w in want])
        assert all(s == 0 for s in status), (extra, status)
        assert outs == want, extra
    outs, status, _ = gb.run(OP_ZSTDSTREAM_COMPRESS, inputs, [len(w) - 1 for w in want])
    for b, s, w in zip(inputs, status, want):
        with pytest.raises(oracle_lib.OracleError) as e:
            o.zstd_stream_compress(b, len(w) - 1)
        assert s == e.value.status, (len(b), s, e.value.status)


def test_a_stream_that_would_flush_before_close_can_be_refused(gb, o):
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    big = (whole * 5)[:4 << 20]
    gb.set_option("zstd.stream.chunked", 0)
    try:
        outs, status, err = gb.run(OP_ZSTDSTREAM_COMPRESS, [big[:-1], big, whole], [o.lib.orc_zstd_stream_max_compressed_length(len(big))] * 3)
    finally:
        gb.set_option("zstd.stream.chunked", 1)
    assert status[0] == 0 and status[2] == 0 and outs[2] == o.zstd_stream_compress(whole)
    assert oracle_lib.status_class(status[1]) == 3 and oracle_lib.status_detail(status[1]) == 103  # INVALID_ARGUMENT / ACHIP_D_UNSUPPORTED


def test_output_stream_twin(o):
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    sink = io.BytesIO()
    sink.close = lambda: None  # (keep the buffer readable after the stream closes its sink)
    s = A.ZstdHipOutputStream(sink)
    s.write(whole[:1000])
    s.write(whole, 1000, 299000)
    s.close()
    s.close()
    assert sink.getvalue() == o.zstd_stream_compress(whole[:300000])
    with pytest.raises(IOError):
        s.write(b"x")


def test_corpus_files_match_the_stream_manifest(gb, o):
    """every corpus file below 4 MiB as one stream: the GPU's bytes against tests/golden/oracle_stream_manifest.tsv (the lines a JDK box
    pins against the real ZstdOutputStream: tools/java/GoldenStreamDump.java)"""
    import hashlib
    rows = [r for r in common.read_manifest_tsv("oracle_stream_manifest.tsv") if r[0] != "*" and r[2] < (4 << 20)]
    corpus = common.corpus_full()
    inputs = [bytes(corpus[r[0]]) for r in rows]
    outs, status, err = gb.run(OP_ZSTDSTREAM_COMPRESS, inputs, [o.lib.orc_zstd_stream_max_compressed_length(len(b)) for b in inputs])
    for r, c, s in zip(rows, outs, status):
        assert s == 0 and len(c) == r[4] and hashlib.sha256(c).hexdigest() == r[5], r[0]



def test_streams_from_4_mib_on_are_written_in_chunks(gb, o):
    """(last in the file: the chunked writer -- 23 blocks flushed at 4 MiB, the window slid by 1920 KiB, 15 blocks per further 1920 KiB, and as
    in the reference 7 blocks after every slide without a match -- was verified on the CPU emulator only before its first GPU run:
    tools/hostemu/check_enc.py --chunked)  The oracle's bytes for streams with no, one, two and six slides; the GPU decoder reads them back;
    the whole corpus as one 14 MB stream against the manifest line a JDK box pins (tools/java/GoldenStreamDump.java)."""
    import hashlib
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(43)
    tiled = whole * 16
    inputs = [tiled[:4 << 20], tiled[:(4 << 20) + 1], tiled[:4 * (1 << 20) + 1920 * 1024], tiled[100:6400100],
              tiled[:3000000] + rng.integers(0, 256, 700000, dtype=np.uint8).tobytes() + tiled[:2500000], tiled[:15 << 20]]
    caps = [o.lib.orc_zstd_stream_max_compressed_length(len(b)) for b in inputs]
    outs, status, err = gb.run(OP_ZSTDSTREAM_COMPRESS, inputs, caps)
    for i, b in enumerate(inputs):
        assert status[i] == 0, (i, len(b), status[i])
        assert outs[i] == o.zstd_stream_compress(b), "input %d (len %d)" % (i, len(b))
    back, status, err = gb.run(OP_ZSTD_DECOMPRESS, outs, [len(b) for b in inputs])
    assert all(s == 0 for s in status), status
    assert [bytes(p) for p in back] == inputs
    star = [r for r in common.read_manifest_tsv("oracle_stream_manifest.tsv") if r[0] == "*"]
    assert len(star) == 1
    import json
    import os
    corpus = common.corpus_full()
    order = [e["file"] for e in json.load(open(os.path.join(common.GOLDEN, "corpus_full.json")))]
    everything = b"".join(bytes(corpus[f]) for f in order)
    assert len(everything) == star[0][2]
    outs, status, err = gb.run(OP_ZSTDSTREAM_COMPRESS, [everything], [o.lib.orc_zstd_stream_max_compressed_length(len(everything))])
    assert status[0] == 0 and len(outs[0]) == star[0][4] and hashlib.sha256(outs[0]).hexdigest() == star[0][5]


def test_input_stream_twin_reads_streams_without_a_content_size(o):
    """ZstdHipInputStream (aircompressor_amd/codecs.py; Java: ZstdHipInputStream.java) = ZstdInputStream read to the end: the repo's own
    chunked streams (>= 4 MiB: no content size in the frame header), several frames back to back, small frames -- with NO size hint: the
    capacity comes from achip_zstd_decompress_bound.  The transliterated reference stream (oracle/_ref, where present) reads the same."""
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    tiled = whole * 16
    plains = [tiled[:(4 << 20) + 1], tiled[100:6400100], whole[:70000], b"x", b""]
    streams = [o.zstd_stream_compress(p) for p in plains]
    streams.append(streams[2] + streams[0] + o.compress("zstd", whole[:300]))  # three frames, the middle one without a content size
    plains.append(plains[2] + plains[0] + whole[:300])
    for z, p in zip(streams, plains):
        with A.ZstdHipInputStream(io.BytesIO(z)) as s:
            got = b""
            buf = bytearray(1 << 20)
            while True:
                n = s.read_into(buf, 0, len(buf))
                if n < 0:
                    break
                got += bytes(buf[:n])
            assert got == p, (len(z), len(p), len(got))
        assert A.ZstdHipInputStream(io.BytesIO(z)).read() == p
    with pytest.raises(IOError):
        A.ZstdHipInputStream(io.BytesIO(b"")).read()
    with pytest.raises(Exception):
        A.ZstdHipInputStream(io.BytesIO(streams[0][:len(streams[0]) // 2])).read()
    s = A.ZstdHipInputStream(io.BytesIO(streams[3]))
    s.close()
    with pytest.raises(IOError):
        s.read()


def test_input_stream_twin_caps_the_one_shot_allocation(o):
    """a few KB of RLE block headers announce gigabytes (the bound adds up to 128 KiB per 4-byte block).  The twin refuses a
    stream whose bound exceeds max_decoded_bytes BEFORE allocating; legal streams below the cap decode as before, and None lifts the cap."""
    import struct
    import aircompressor_amd as A
    # one frame, no content size, window descriptor 128 KiB; 1000 RLE blocks of 131072 bytes each (4 bytes of input apiece), the last one marked
    blocks = b"".join(struct.pack("<I", (131072 << 3) | (1 << 1) | (1 if i == 999 else 0))[:3] + b"z" for i in range(1000))
    amplified = struct.pack("<I", 0xFD2FB528) + bytes([0x00, 0x38]) + blocks
    assert len(amplified) < 4200
    with pytest.raises(IOError, match="max_decoded_bytes"):
        A.ZstdHipInputStream(io.BytesIO(amplified), max_decoded_bytes=1 << 20).read()
    plain = A.ZstdHipInputStream(io.BytesIO(amplified), max_decoded_bytes=None).read()  # 125 MiB: legal, decodes when the caller allows it
    assert len(plain) == 1000 * 131072 and plain.count(b"z") == len(plain)
    small = o.zstd_stream_compress(b"abc" * 1000)
    assert A.ZstdHipInputStream(io.BytesIO(small), max_decoded_bytes=4096).read() == b"abc" * 1000


# ---- the reader, a step at a time (achip_zstdstream_decompress_begin / _feed / _end): SURVEY 8f row 3, ZstdIncrementalFrameDecompressor.java:44-72,216-234 ----

def _device_bytes_in_use():
    import torch
    free, total = torch.cuda.mem_get_info(0)
    return total - free


def test_a_long_stream_decodes_in_bounded_memory(o):
    """ONE frame of 320 MiB (libzstd level 3: a 2 MiB window, thousands of blocks, tables and repeat offsets inherited from block to block, a
    checksum over everything) read through ZstdHipInputStream a megabyte at a time: the plaintext comes back byte for byte, and the device
    memory the open stream holds stays a small constant (the whole-buffer twin of rounds 2-4 needed input + bound: > 400 MB here)."""
    import hashlib
    import pyarrow as pa
    import torch
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(3)
    pieces = []
    total = 0
    while total < 320 << 20:   # text with random splices: matches reach back up to the window
        at = int(rng.integers(0, len(whole) - 70000))
        n = int(rng.integers(1000, 70000))
        pieces.append(whole[at:at + n])
        total += n
    plain = b"".join(pieces)
    z = pa.Codec("zstd", compression_level=3).compress(plain, asbytes=True)
    want = hashlib.sha256(plain).hexdigest()
    del pieces
    torch.cuda.synchronize()
    before = _device_bytes_in_use()
    h = hashlib.sha256()
    got = 0
    peak = 0
    with A.ZstdHipInputStream(io.BytesIO(z)) as s:
        buf = bytearray(3 << 20)
        while True:
            n = s.read_into(buf, 0, len(buf))
            if n < 0:
                break
            h.update(memoryview(buf)[:n])
            got += n
            peak = max(peak, _device_bytes_in_use() - before)
    assert got == len(plain) and h.hexdigest() == want
    assert peak < (48 << 20), "the open stream held %d MiB of device memory" % (peak >> 20)


def test_streams_of_every_shape_read_in_small_pieces(o):
    """The writer's streams (one frame, no content size from 4 MiB on), frames back to back, libzstd frames with and without checksum, raw and RLE
    blocks, a window larger than a step -- read with small and odd buffer sizes from a source that hands out a few bytes at a time."""
    import pyarrow as pa
    import aircompressor_amd as A

    class Dribble(io.RawIOBase):  # a source whose read() returns at most 70 001 bytes whatever is asked
        def __init__(self, data):
            self.data, self.at = data, 0

        def read(self, n=-1):
            n = 70001 if n is None or n < 0 else min(n, 70001)
            piece = self.data[self.at:self.at + n]
            self.at += len(piece)
            return piece

    whole = b"".join(d for _, d, _ in common.corpus_sample())
    tiled = whole * 16
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, 400000, dtype=np.uint8).tobytes()
    plains = [tiled[:(4 << 20) + 1], tiled[100:9400100], whole[:70000], b"x", b"", noise + b"\0" * 900000 + noise[:100000], tiled[:20 << 20]]
    zc = pa.Codec("zstd", compression_level=3)
    streams = [o.zstd_stream_compress(p) for p in plains[:6]] + [zc.compress(plains[6], asbytes=True)]
    streams.append(streams[2] + streams[0] + o.compress("zstd", whole[:300]) + zc.compress(noise, asbytes=True))
    plains.append(plains[2] + plains[0] + whole[:300] + noise)
    for k, (z, p) in enumerate(zip(streams, plains)):
        for size in (1 << 20, 77777, 5):
            if size == 5 and len(p) > 100000:
                continue
            with A.ZstdHipInputStream(Dribble(z)) as s:
                got = bytearray()
                buf = bytearray(size + 3)
                while True:
                    n = s.read_into(buf, 3, size)
                    if n < 0:
                        break
                    assert 0 < n <= size
                    got += buf[3:3 + n]
                assert bytes(got) == p, (k, size, len(got), len(p))
    with pytest.raises(IOError, match="Not enough input"):
        A.ZstdHipInputStream(io.BytesIO(b"")).read()
    with pytest.raises(IOError, match="Not enough input"):
        A.ZstdHipInputStream(io.BytesIO(streams[0][:len(streams[0]) // 2])).read()


def test_a_damaged_stream_fails_at_the_read_that_reaches_the_damage(o):
    """Damage in the middle of a long frame: every byte of the blocks in front of the damaged block is delivered, then the read fails -- where
    ZstdInputStream throws.  Held against the transliterated reference reader where oracle/_ref/libref.so is present: it fails too, having
    delivered no more than this reader (it keeps a window's worth of decoded bytes back)."""
    import ctypes
    import os
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    plain = (whole * 8)[:6000000]
    z = bytearray(o.zstd_stream_compress(plain))
    # block boundaries of the one frame (header: magic 4 + descriptor 1 + window 1 [+ content size])
    fhd = z[4]
    pos = 5 + (0 if fhd & 0x20 else 1) + ((1 if fhd & 0x20 else 0) if fhd >> 6 == 0 else 1 << (fhd >> 6))
    starts = []
    while True:
        hd = int.from_bytes(z[pos:pos + 3], "little")
        starts.append(pos)
        pos += 3 + (1 if (hd >> 1) & 3 == 1 else hd >> 3)
        if hd & 1:
            break
    assert len(starts) > 30
    ref_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref.so")
    ref = ctypes.CDLL(ref_path) if os.path.exists(ref_path) else None
    if ref is not None:
        ref.ref_zstd_stream_decompress_partial.restype = ctypes.c_int64
        ref.ref_zstd_stream_decompress_partial.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    for victim in (2, 17, len(starts) - 2):
        g = bytearray(z)
        g[starts[victim] + 3] ^= 0xFF  # the damaged block's literals section header: the block cannot be parsed
        g[starts[victim] + 4] ^= 0xFF
        s = A.ZstdHipInputStream(io.BytesIO(bytes(g)))
        got = bytearray()
        buf = bytearray(200000)
        failed = False
        try:
            while True:
                n = s.read_into(buf, 0, len(buf))
                if n < 0:
                    break
                got += buf[:n]
        except (A.MalformedInputException, IOError):
            failed = True
        assert failed, victim
        assert bytes(got) == plain[:len(got)] and len(got) > 0
        # every block in front of the victim was delivered whole: the stream writer's blocks are 128 KiB of input each
        assert len(got) == victim * 131072, (victim, len(got))
        if ref is not None:
            out = ctypes.create_string_buffer(len(plain) + 1)
            delivered, eo = ctypes.c_int64(0), ctypes.c_int64(0)
            src = ctypes.create_string_buffer(bytes(g), len(g))
            r = ref.ref_zstd_stream_decompress_partial(src, len(g), out, len(plain) + 1, 200000, ctypes.byref(delivered), ctypes.byref(eo))
            assert r < 0, "the reference reader decodes the damaged stream"
            assert delivered.value <= len(got) and out.raw[:delivered.value] == plain[:delivered.value]


def test_output_stream_twin_writes_in_chunks_whatever_the_write_sizes(o):
    """ZstdHipOutputStream over achip_zstdstream_compress_begin / _feed / _finish: the bytes of ZstdOutputStream (the oracle's) for streams below and
    beyond 4 MiB -- no, one, two and six flushes -- written in pieces of many sizes; the sink receives the flushed blocks BEFORE close() (the
    whole-buffer twin of rounds 2-4 held everything until then); an empty stream; and the incremental reader reads the longest one back."""
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(47)
    tiled = whole * 16
    inputs = [b"", b"z", whole[:300000], tiled[:4 << 20], tiled[:(4 << 20) + 1], tiled[100:6400100],
              tiled[:3000000] + rng.integers(0, 256, 700000, dtype=np.uint8).tobytes() + tiled[:2500000], tiled[:15 << 20]]
    for k, data in enumerate(inputs):
        want = o.zstd_stream_compress(data)
        for piece in ((1 << 30, 1 << 20, 70001, 999) if len(data) <= (7 << 20) else (3000001,)):
            if piece == 999 and len(data) > 400000:
                continue
            sink = io.BytesIO()
            sink.close = lambda: None
            s = A.ZstdHipOutputStream(sink)
            before_close = 0
            for at in range(0, len(data), piece):
                s.write(data, at, min(piece, len(data) - at))
                before_close = len(sink.getvalue())
            s.close()
            assert sink.getvalue() == want, (k, len(data), piece, len(sink.getvalue()), len(want))
            if len(data) >= (4 << 20):
                assert before_close > 0, "nothing reached the sink before close()"
    assert A.ZstdHipInputStream(io.BytesIO(o.zstd_stream_compress(inputs[-1]))).read() == inputs[-1]


def test_reader_corners_the_stream_fuzzer_found(o):
    """tools/fuzz_zstd_stream.py against the transliterated ZstdInputStream (round 5): (1) up to three bytes behind the last frame end the stream quietly -- the Java
    reader asks for a magic's four bytes, does not get them and is at a stopping point (ZstdInputStream.java:81-86) --, four bytes of garbage are an invalid magic,
    and fewer than four bytes in front of the FIRST frame are "Not enough input bytes" (the Java state is INITIAL, not a stopping point); (2) RAW and RLE blocks that
    say more than 128 KiB are decoded for what they say (ZstdIncrementalFrameDecompressor.java:204-226 has no bound on them; here they become several blocks);
    (3) an empty stream written right after other streams is a whole frame (its state used to be cleared on another stream than the one its step runs on)."""
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    z = o.zstd_stream_compress(whole[:300000])
    for tail in (b"\x01", b"\x28\xb5", b"\x28\xb5\x2f"):
        assert A.ZstdHipInputStream(io.BytesIO(z + tail)).read() == whole[:300000]
    with pytest.raises((A.MalformedInputException, IOError)):
        A.ZstdHipInputStream(io.BytesIO(z + b"\x01\x02\x03\x04")).read()
    with pytest.raises(IOError, match="Not enough input"):
        A.ZstdHipInputStream(io.BytesIO(b"\x28\xb5")).read()
    # a frame by hand: window 1 MiB, no checksum, no content size; a compressed-free body of RLE 200 000 x 'q', RAW 150 000 bytes, RLE 131 073 x 0, RAW 5 (last)
    raw1, raw2 = whole[1000:151000], b"tail!"
    def block(kind, size, payload, last=0):
        return int((size << 3) | (kind << 1) | last).to_bytes(3, "little") + payload
    frame = b"\x28\xb5\x2f\xfd" + b"\x00" + b"\x50" + block(1, 200000, b"q") + block(0, len(raw1), raw1) + block(1, 131073, b"\x00") + block(0, len(raw2), raw2, 1)
    plain = b"q" * 200000 + raw1 + bytes(131073) + raw2
    stream = z + frame + o.zstd_stream_compress(whole[:5000])
    for size in (1 << 20, 4097):
        s = A.ZstdHipInputStream(io.BytesIO(stream))
        got = bytearray()
        buf = bytearray(size)
        while True:
            n = s.read_into(buf, 0, size)
            if n < 0:
                break
            got += buf[:n]
        assert bytes(got) == whole[:300000] + plain + whole[:5000], (size, len(got))
    # (3)
    big = io.BytesIO()
    big.close = lambda: None
    with A.ZstdHipOutputStream(big) as s:
        s.write(whole[:2000000])
    for _ in range(12):
        sink = io.BytesIO()
        sink.close = lambda: None
        A.ZstdHipOutputStream(sink).close()
        assert sink.getvalue() == o.zstd_stream_compress(b"")



def test_reader_window_rules_are_the_java_readers(o):
    """(1) a window descriptor above 8 MiB -- the Java reader copies the frame's RAW / RLE blocks and fails its first COMPRESSED block with
    "Window size too large (not yet supported)" (ZstdFrameDecompressor.java:303): so does this one; (2) the same blocks behind a 1 MiB descriptor decode."""
    import aircompressor_amd as A
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    def block(kind, size, payload, last=0):
        return int((size << 3) | (kind << 1) | last).to_bytes(3, "little") + payload
    raw = whole[2000:9000]
    magic = b"\x28\xb5\x2f\xfd"
    big_window = b"\x00" + bytes([14 << 3])  # no single segment, no checksum, no content size; 2^(10 + 14) = 16 MiB
    assert A.ZstdHipInputStream(io.BytesIO(magic + big_window + block(1, 70000, b"w") + block(0, len(raw), raw, 1))).read() == b"w" * 70000 + raw
    # a compressed block of the Java encoder's (a single-segment frame with a checksum: header 4 + 1 + content-size bytes, then the blocks, then 4 bytes)
    plain = whole[:100000]
    f = o.compress("zstd", plain)
    fhd = f[4]
    assert fhd & 0x20 and fhd & 4
    cs = fhd >> 6
    body = f[5 + (1 if cs == 0 else 1 << cs):-4]
    assert A.ZstdHipInputStream(io.BytesIO(magic + b"\x00" + bytes([10 << 3]) + body)).read() == plain  # 1 MiB: fine
    s = A.ZstdHipInputStream(io.BytesIO(magic + big_window + block(0, len(raw), raw) + body))
    with pytest.raises((A.MalformedInputException, IOError), match="Window size too large"):
        got = s.read()
        raise AssertionError("decoded %d bytes of a frame the Java reader refuses" % len(got))
