"""The oracle's x-snappy-framed restatement (oracle/snappy_framed.c) against the reference's own tests
(T/snappy/TestSnappyStream.java:50-176) and the RFC 3720 B.4 CRC-32C vectors.  CPU only."""
import struct

import pytest

from tests import oracle_lib
from tests.oracle_lib import OracleError

HEADER = bytes([0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59])  # M/snappy/SnappyFramed.java:31
D = dict(eof_stream_header=88, bad_stream_header=89, eof_block_header=90, eof_frame=91, stream_id_length=92, unskippable=93,
         invalid_length=94, checksum=95, output_too_small=96, max_output=97)


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def stream(block):
    return HEADER + bytes(block)  # blockToStream :271-277


def test_crc32c_known_answers(o):
    assert o.crc32c(b"123456789") == 0xE3069283
    assert o.crc32c(bytes(32)) == 0x8A9136AA            # RFC 3720 B.4
    assert o.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert o.crc32c(bytes(range(32))) == 0x46DD794E
    assert o.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert o.crc32c(b"") == 0
    crc = o.crc32c(b"abc")
    assert o.crc32c(b"abc", masked=True) == (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF  # Crc32C.java:47-50


def test_simple(o):  # testSimple :50-79
    original = b"aaaaaaaaaaaabbbbbbbaaaaaa"
    c = o.compress("snappyframed", original)
    assert o.decompress("snappyframed", c, len(original)) == original
    assert len(c) == 37            # 10 byte stream header, 4 byte block header, 4 byte crc, 19 bytes
    assert c[:10] == HEADER
    assert c[10] == 0x00           # compressed data
    assert c[11:14] == bytes([0x17, 0, 0])
    assert c[14:18] == bytes([0xA8, 0xCD, 0x74, 0x92])  # crc32c: 0x9274cda8


def test_uncompressible(o):  # testUncompressible :81-100
    random = o.random_generator(1.0)[:5000].tobytes()
    c = o.compress("snappyframed", random)
    assert o.decompress("snappyframed", c, 5000) == random
    assert len(c) == 5000 + 10 + 4 + 4
    assert c[10] == 0x01           # uncompressed data
    assert c[11:14] == bytes([0x8c, 0x13, 0x00])  # length 5004


def test_empty(o):  # testEmptyCompression :102-109
    assert o.compress("snappyframed", b"") == HEADER
    assert o.decompress("snappyframed", HEADER, 0) == b""


@pytest.mark.parametrize("block, detail", [
    ([0], "eof_block_header"),                                      # testShortBlockHeader :111-117
    ([1, 8, 0, 0, 0, 0, 0, 0, ord("x"), ord("x")], "eof_frame"),    # testShortBlockData :119-126
    ([1, 4, 0, 0, 0, 0, 0, 0], "invalid_length"),                   # testInvalidBlockSizeZero :151-158
    ([1, 5, 0, 0, 0, 0, 0, 0, ord("a")], "checksum"),               # testInvalidChecksum :160-167
    ([0xff, 5, 0, 0, 1, 2, 3, 4, 5], "stream_id_length"),
])
def test_malformed_streams(o, block, detail):
    with pytest.raises(OracleError) as e:
        o.decompress("snappyframed", stream(block), 1024)
    assert (e.value.cls, e.value.detail) == (1, D[detail])


def test_chunk_flags(o):  # testUnskippableChunkFlags :128-136, testSkippableChunkFlags :138-149
    for flag in range(2, 0x80):
        with pytest.raises(OracleError) as e:
            o.decompress("snappyframed", stream([flag, 5, 0, 0, 0, 0, 0, 0, 0]), 16)
        assert e.value.detail == D["unskippable"]
    for flag in range(0x80, 0xff):
        assert o.decompress("snappyframed", stream([flag, 5, 0, 0, 0, 0, 0, 0, 0]), 16) == b""


def test_stream_header_errors(o):
    with pytest.raises(OracleError) as e:
        o.decompress("snappyframed", HEADER[:9], 16)
    assert e.value.detail == D["eof_stream_header"]
    with pytest.raises(OracleError) as e:
        o.decompress("snappyframed", b"\xff\x06\x00\x00sNaPpX", 16)
    assert e.value.detail == D["bad_stream_header"]


@pytest.mark.parametrize("size, compressed", [(100000, False), (500000, True), (100000, True)])
def test_larger_frames(o, size, compressed):  # testLargerFrames_* :178-268: chunks beyond 64 KiB are read
    random = o.random_generator(0.5)[:size].tobytes()
    data = o.compress("snappy", random) if compressed else random
    s = HEADER + bytes([0 if compressed else 1]) + struct.pack("<I", len(data) + 4)[:3] + struct.pack("<I", o.crc32c(random, masked=True)) + data
    assert o.decompress("snappyframed", s, size) == random


def test_large_writes_round_trip(o):  # testLargeWrites :279-312
    random = o.random_generator(0.5)[:500000].tobytes()
    c = o.compress("snappyframed", random)
    assert len(c) < len(random)
    assert o.decompress("snappyframed", c, len(random)) == random
    # chunk walk: full 64 KiB blocks then the rest, each with the masked CRC of its plaintext
    pos, off = 10, 0
    while pos < len(c):
        flag, length = c[pos], int.from_bytes(c[pos + 1:pos + 4], "little")
        n = min(65536, len(random) - off)
        assert struct.unpack("<I", c[pos + 4:pos + 8])[0] == o.crc32c(random[off:off + n], masked=True)
        body = c[pos + 8:pos + 4 + length]
        assert (o.decompress("snappy", body, n) if flag == 0 else body) == random[off:off + n]
        pos += 4 + length
        off += n
    assert off == len(random)


def test_capacity_rules_of_the_one_shot_form(o):
    random = o.random_generator(0.5)[:70000].tobytes()
    c = o.compress("snappyframed", random)
    with pytest.raises(OracleError) as e:
        o.decompress("snappyframed", c, 69999)
    assert (e.value.cls, e.value.detail) == (2, D["output_too_small"])
    with pytest.raises(OracleError) as e:
        o.compress("snappyframed", random, cap=o.max_compressed_length("snappyframed", 70000) - 1)
    assert (e.value.cls, e.value.detail) == (2, D["max_output"])
    assert o.max_compressed_length("snappyframed", 70000) == 10 + 2 * 8 + 70000
