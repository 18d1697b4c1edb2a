"""The oracle's LZ4 frame codec (oracle/lz4_frame.c) against the reference's own test vectors (rebuilt in
tests/lz4_frame_vectors.py from T/lz4/TestLz4FrameDecompressor.java:61-230) and against liblz4's frame codec (pyarrow's
"lz4" codec = the third-party verifier, like lz4-java's frame streams in T/lz4/TestLz4FrameJava.java)."""
import numpy as np
import pytest

from tests import common, oracle_lib, lz4_frame_vectors
from tests.oracle_lib import OracleError


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def message(o, status):
    import ctypes
    import aircompressor_amd as A
    lib = A.load_library()
    return lib.achip_detail_message(oracle_lib.status_detail(status)).decode()


def test_reference_vectors(o):
    for name, frame, cap, want, msg in lz4_frame_vectors.cases(o):
        if want is not None:
            assert o.decompress("lz4frame", frame, cap) == want, name
        else:
            with pytest.raises(OracleError) as e:
                o.decompress("lz4frame", frame, cap)
            if msg is not None:
                assert msg in message(o, e.value.status), (name, message(o, e.value.status))


def test_max_compressed_length(o):
    # header 7 + end mark 4 + 4 per 4 MiB block + the data (Lz4FrameCompression.java:70-83)
    assert o.max_compressed_length("lz4frame", 0) == 11
    assert o.max_compressed_length("lz4frame", 1) == 16
    assert o.max_compressed_length("lz4frame", 4 << 20) == 11 + 4 + (4 << 20)
    assert o.max_compressed_length("lz4frame", (4 << 20) + 1) == 11 + 8 + (4 << 20) + 1


def test_round_trip_and_structure(o):
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()[:6]] + common.synthetic_blocks(3, 12)
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()) * 4)  # 4.75 MiB: two blocks
    blocks.append(common.synthetic_blocks(9, 1)[0][:1000] + bytes(np.random.default_rng(1).integers(0, 256, 70000, dtype=np.uint8)))  # incompressible => stored block
    for b in blocks:
        z = o.compress("lz4frame", b)
        assert len(z) <= o.max_compressed_length("lz4frame", len(b))
        assert z[:7] == bytes([0x04, 0x22, 0x4D, 0x18, 0x60, 0x70, 0x73])  # magic, FLG, BD 4 MB and its checksum byte
        assert o.decompress("lz4frame", z, len(b)) == b
        assert o.decompress("lz4frame", z + z, 2 * len(b)) == b + b
        if len(b) > 0:
            with pytest.raises(OracleError):
                o.decompress("lz4frame", z, len(b) - 1)
        with pytest.raises(OracleError):
            o.compress("lz4frame", b, len(z) - 1)


def test_cross_decode_with_liblz4(o):
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("lz4")
    blocks = [d for _, d, _ in common.corpus_sample()[:8]] + common.synthetic_blocks(4, 8) + [b"", b"a", b"abc" * 1000]
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()) * 4)
    for b in blocks:
        theirs = codec.compress(b, asbytes=True)
        if theirs[4] & 0x20:
            assert o.decompress("lz4frame", theirs, len(b)) == b        # liblz4's frames (block size id, content size / checksums as it writes them)
        else:
            # liblz4 links the blocks of multi-block frames; the reference rejects those (Lz4FrameCompression.java:212-214)
            with pytest.raises(OracleError) as e:
                o.decompress("lz4frame", theirs, len(b))
            assert oracle_lib.status_detail(e.value.status) == 72
        ours = o.compress("lz4frame", b)
        assert codec.decompress(ours, decompressed_size=len(b)).to_pybytes() == b   # liblz4 accepts the Java-format frames
