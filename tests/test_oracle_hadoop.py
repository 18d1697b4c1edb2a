"""The oracle's Hadoop LZ4 / Snappy block streams (oracle/hadoop_streams.c) against a byte-level description of the format
(M/lz4/Lz4HadoopOutputStream.java:107-118: [BE int plaintext length][BE int compressed length][block], chunks of bufferSize - overhead
plaintext bytes) and hand-built streams for every branch of the two readers (M/lz4/Lz4HadoopInputStream.java:47-156,
M/snappy/SnappyHadoopInputStream.java:44-141), driven like T/HadoopCodecDecompressor.java:40-60.  The reference's own tests round-trip these
streams through org.apache.hadoop's codecs, which are not available here: there are no golden vectors to pin against."""
import struct

import numpy as np
import pytest

from tests import common
from tests.oracle_lib import OracleError

D = {"TRUNCATED_INT": 104, "EOF_BLOCK_DATA": 105, "CHUNK_EXCEEDS_BLOCK": 106, "LENGTH_MISMATCH": 107, "NOT_CONSUMED": 108, "NEGATIVE_LENGTH": 109, "MAX_OUTPUT": 110}
BUF = 262144


def input_max(codec, buf):
    return buf - (max(int(buf * 0.01), 10) if codec == "lz4" else buf // 6 + 32)


def be(v):
    return struct.pack(">i", v)


def stream(codec, oracle, pieces):
    """pieces: (declared block length or None, plaintext) -> [U][clen][block] per piece; None = a further chunk of the open block"""
    out = b""
    for declared, plain in pieces:
        c = oracle.compress(codec, plain)
        out += (be(declared) if declared is not None else b"") + be(len(c)) + c
    return out


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_writer_emits_the_documented_format(oracle, codec):
    assert input_max("lz4", BUF) == 259523 and input_max("snappy", BUF) == 218422
    data = b"".join(d for _, d, _ in common.corpus_sample())[:700000]
    for buf in (BUF, 4096, 70000):
        chunk = input_max(codec, buf)
        s = oracle.hadoop_compress(codec, data, buf)
        pos = 0
        plain = b""
        n_chunks = 0
        while pos < len(s):
            u, c = struct.unpack(">ii", s[pos:pos + 8])
            block = s[pos + 8:pos + 8 + c]
            assert block == oracle.compress(codec, data[len(plain):len(plain) + u])
            assert u == min(chunk, len(data) - len(plain))
            plain += oracle.decompress(codec, block, u)
            pos += 8 + c
            n_chunks += 1
        assert plain == data and n_chunks == -(-len(data) // chunk)
        assert len(s) <= oracle.hadoop_max_compressed_length(codec, len(data), buf)
        assert oracle.hadoop_decompress(codec, s, len(data), buf) == data
    assert oracle.hadoop_compress(codec, b"") == b""
    assert oracle.hadoop_decompress(codec, b"", 10) == b""
    with pytest.raises(OracleError) as e:
        oracle.hadoop_compress(codec, data, BUF, cap=oracle.hadoop_max_compressed_length(codec, len(data)) - 1)
    assert e.value.cls == 2 and e.value.detail == D["MAX_OUTPUT"]


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_reader_branches(oracle, codec):
    a, b, c = b"hello hello hello hello " * 40, b"abcdefgh" * 300, bytes(range(256)) * 3
    # blocks of several chunks, empty blocks between them (skipped: `while (uncompressedBlockLength == 0)`), a trailing empty block
    s = stream(codec, oracle, [(len(a) + len(b), a), (None, b)]) + be(0) + be(0) + stream(codec, oracle, [(len(c), c)]) + be(0)
    assert oracle.hadoop_decompress(codec, s, len(a + b + c)) == a + b + c
    assert oracle.hadoop_decompress(codec, s, len(a + b + c) + 1000) == a + b + c
    # the destination is too small: after the output is full, read() finds a byte
    for cap in (len(a + b + c) - 1, len(a) + 5, len(a), 10, 0):
        with pytest.raises(OracleError) as e:
            oracle.hadoop_decompress(codec, s, cap)
        assert e.value.cls == 2 and e.value.detail == D["NOT_CONSUMED"], cap
    # -1 where a length is expected ends the stream quietly (the Java code cannot tell it from EOF)
    assert oracle.hadoop_decompress(codec, stream(codec, oracle, [(len(a), a)]) + be(-1), len(a) + 10) == a
    assert oracle.hadoop_decompress(codec, stream(codec, oracle, [(len(a), a)]) + be(50) + be(-1), len(a) + 10) == a
    # ... for the read that sees it: the harness's closing read() carries on behind it
    with pytest.raises(OracleError) as e:
        oracle.hadoop_decompress(codec, stream(codec, oracle, [(len(a), a)]) + be(-1) + b"garbage", len(a) + 10)
    assert e.value.cls == 1 and e.value.detail == D["TRUNCATED_INT"]
    # a block length with no chunk behind it: end of stream
    assert oracle.hadoop_decompress(codec, stream(codec, oracle, [(len(a), a)]) + be(77), len(a)) == a
    # truncated ints and chunk data
    good = stream(codec, oracle, [(len(a), a)])
    for cut, detail in ((2, "TRUNCATED_INT"), (6, "TRUNCATED_INT"), (len(good) - 1, "EOF_BLOCK_DATA"), (9, "EOF_BLOCK_DATA")):
        with pytest.raises(OracleError) as e:
            oracle.hadoop_decompress(codec, good[:cut], len(a))
        assert e.value.cls == 1 and e.value.detail == D[detail], cut
    # a negative chunk length: the LZ4 reader takes ANY negative value for the end of the stream for the read that sees it
    # (Lz4HadoopInputStream.java:51-54,65-68 `compressedChunkLength < 0`; the harness's closing read() then meets "xxxxx" as a chunk header
    # whose data is missing), the Snappy reader only -1 (documented deviation for the other values: oracle/hadoop_streams.c)
    with pytest.raises(OracleError) as e:
        oracle.hadoop_decompress(codec, be(10) + be(-5) + b"xxxxx", 100)
    assert e.value.detail == (D["EOF_BLOCK_DATA"] if codec == "lz4" else D["NEGATIVE_LENGTH"])
    if codec == "lz4":
        assert oracle.hadoop_decompress(codec, good + be(10) + be(-5), len(a) + 10) == a
    # a corrupt chunk: the block codec's own exception and offset
    bad = bytearray(good)
    bad[8 + 3] ^= 0xFF
    try:
        expected = None
        oracle.decompress(codec, bytes(bad[8:]), len(a))
    except OracleError as e0:
        expected = (e0.status, e0.offset)
    if expected is not None:
        with pytest.raises(OracleError) as e:
            oracle.hadoop_decompress(codec, bytes(bad), len(a))
        assert (e.value.status, e.value.offset) == expected


def test_snappy_reader_specifics(oracle):
    a = b"0123456789" * 100
    # a chunk that announces more than its block has left
    with pytest.raises(OracleError) as e:
        oracle.hadoop_decompress("snappy", stream("snappy", oracle, [(len(a) - 1, a)]), len(a))
    assert e.value.cls == 1 and e.value.detail == D["CHUNK_EXCEEDS_BLOCK"]
    # a chunk of no bytes ends the stream (`if (uncompressedChunkLength == 0) return -1`)
    s = stream("snappy", oracle, [(len(a), a)]) + be(5) + be(1) + b"\x00"
    assert oracle.hadoop_decompress("snappy", s, 3 * len(a)) == a
    # ... for the read that sees it; the harness's closing read() takes what follows for that block's next chunk
    with pytest.raises(OracleError) as e:
        oracle.hadoop_decompress("snappy", s + stream("snappy", oracle, [(len(a), a)]), 3 * len(a))
    assert e.value.cls == 1 and e.value.detail == D["EOF_BLOCK_DATA"]


def test_lz4_reader_specifics(oracle):
    a, b = b"0123456789" * 100, b"xyz" * 500
    # LZ4 has no announced chunk length: a block whose chunks produce more than it declared just goes negative and every following
    # [length][block] pair is read as a chunk of it
    s = stream("lz4", oracle, [(5, a), (None, b)])
    assert oracle.hadoop_decompress("lz4", s, len(a + b)) == a + b
    # remaining < declared block length: the chunk goes through the stream's own buffer (bufferSize + 8 bytes) and is handed out in parts
    s = stream("lz4", oracle, [(len(a) + 100, a)])
    assert oracle.hadoop_decompress("lz4", s, len(a) + 50) == a
    # ... whose capacity decides what the block decoder says: here the chunk does not fit a 256 + 8 byte buffer
    with pytest.raises(OracleError) as e:
        oracle.hadoop_decompress("lz4", s, len(a) + 50, buffer_size=256)
    assert e.value.cls in (1, 2)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_round_trip_over_the_corpus_sample(oracle, codec):
    rng = np.random.default_rng(5)
    for _, d, _ in common.corpus_sample()[:8]:
        for buf in (BUF, int(rng.integers(300, 70000))):
            assert oracle.hadoop_decompress(codec, oracle.hadoop_compress(codec, d, buf), len(d), buf) == d
