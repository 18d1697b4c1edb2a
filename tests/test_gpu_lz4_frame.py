"""GPU parity tests (through the C ABI) for the LZ4 frame container: the reference's rebuilt test vectors
(T/lz4/TestLz4FrameDecompressor.java:61-230), byte-identical frames on encode, liblz4 frames on decode, systematic
corruption with the oracle's status / offset, and a full-size property."""
import os

import numpy as np
import pytest

from tests import common, oracle_lib, lz4_frame_vectors
from tests.oracle_lib import OracleError

pytestmark = pytest.mark.gpu
OP_DECOMPRESS, OP_COMPRESS = 6, 7


# every reader variant runs: 2 the default (the block list when a probe finds short sequences, else a wavefront per item), 0 always a wavefront
# per item, 1 always the block list through the two-pass block decoder
_VARIANTS = [2, 0, 1]


@pytest.fixture(scope="module", params=_VARIANTS, ids=["auto", "wave-per-item", "block-list"])
def gb(request):
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0, options={"lz4frame.decompress.variant": request.param})


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def expect(o, data, cap):
    try:
        return 0, 0, o.decompress("lz4frame", data, cap)
    except OracleError as e:
        return e.status, e.offset, None


def check_decode(gb, o, cases, unaligned=False):
    outs, status, err = gb.run(OP_DECOMPRESS, [c for c, _ in cases], [cap for _, cap in cases], unaligned=unaligned)
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = expect(o, c, cap)
        assert status[i] == est, "case %d: gpu status %d (offset %d) oracle %d (offset %d)" % (i, status[i], err[i], est, eoff)
        if est != 0:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)
        else:
            assert outs[i] == eout, "case %d" % i


def test_reference_vectors(gb, o):
    cases = [(frame, cap) for _, frame, cap, _, _ in lz4_frame_vectors.cases(o)]
    check_decode(gb, o, cases)
    check_decode(gb, o, cases, unaligned=True)
    # the expectations themselves (plaintext / message) are asserted on the oracle in tests/test_oracle_lz4_frame.py


def inputs():
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()[:8]] + common.synthetic_blocks(3, 18)
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()) * 4)   # 4.75 MiB: two blocks
    blocks.append(bytes(np.random.default_rng(1).integers(0, 256, 70000, dtype=np.uint8)))  # incompressible: stored block
    base = common.corpus_sample()[0][1]
    blocks += [base[:n] for n in (1, 2, 11, 12, 13, 63, 64, 65, 4095, 4096, 4097)]
    return blocks


def test_compress_is_byte_identical_and_round_trips(gb, o):
    blocks = inputs()
    caps = [o.max_compressed_length("lz4frame", len(b)) for b in blocks]
    outs, status, _ = gb.run(OP_COMPRESS, blocks, caps)
    assert all(s == 0 for s in status), status
    for i, (b, z) in enumerate(zip(blocks, outs)):
        assert z == o.compress("lz4frame", b), "input %d (len %d)" % (i, len(b))
    plain, status, err = gb.run(OP_DECOMPRESS, outs, [len(b) for b in blocks])
    assert all(s == 0 for s in status), (status, err)
    assert plain == blocks
    # concatenated frames, with exact-fit and roomy outputs
    cat = [outs[i] + outs[i + 1] for i in range(0, 10, 2)]
    want = [blocks[i] + blocks[i + 1] for i in range(0, 10, 2)]
    plain, status, err = gb.run(OP_DECOMPRESS, cat + cat, [len(w) for w in want] + [len(w) + 100 for w in want])
    assert all(s == 0 for s in status) and plain == want + want


def test_compress_output_too_small(gb, o):
    b = common.corpus_sample()[0][1]
    full = o.compress("lz4frame", b)
    caps = [0, 3, 6, 7, 10, 11, len(full) - 1, len(full), len(full) + 1]
    outs, status, _ = gb.run(OP_COMPRESS, [b] * len(caps), caps)
    for c, z, s in zip(caps, outs, status):
        try:
            want = o.compress("lz4frame", b, c)
            assert s == 0 and z == want, c
        except OracleError as e:
            assert s == e.status, (c, s, e.status)


def test_liblz4_frames_and_corruptions(gb, o):
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("lz4")
    rng = np.random.default_rng(17)
    blocks = [d for _, d, _ in common.corpus_sample()[:6]] + common.synthetic_blocks(4, 6) + [b"", b"a", b"abc" * 1000]
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()) * 2)   # liblz4 links the blocks of this one: rejected like the reference
    cases = []
    for b in blocks:
        z = codec.compress(b, asbytes=True)
        cases += [(z, len(b)), (z, len(b) + 33)]
        if len(b) > 0:
            cases.append((z, len(b) - 1))
    for b in blocks[:5]:
        z = bytearray(o.compress("lz4frame", b))
        cases += [(bytes(z[:len(z) // 2]), len(b)), (bytes(z[:-1]), len(b)), (bytes(z[:-4]), len(b)), (bytes(z), len(b) // 2)]
        for _ in range(12):
            m = bytearray(z)
            for _ in range(int(rng.integers(1, 3))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cases += [(bytes(m), len(b)), (bytes(m), len(b) + 64)]
    check_decode(gb, o, cases)


def test_host_api_mirrors_reference(o):
    import aircompressor_amd as A
    comp, decomp = A.Lz4FrameHipCompressor(), A.Lz4FrameHipDecompressor()
    data = common.corpus_sample()[1][1]
    cap = comp.max_compressed_length(len(data))
    assert cap == o.max_compressed_length("lz4frame", len(data)) == len(data) + 7 + 4 + 4
    out = bytearray(cap + 5)
    n = comp.compress(data, 0, len(data), out, 5, cap)
    assert bytes(out[5:5 + n]) == o.compress("lz4frame", data)
    back = bytearray(len(data))
    assert decomp.decompress(out, 5, n, back, 0, len(data)) == len(data) and bytes(back) == data
    with pytest.raises(A.MalformedInputException) as e:
        decomp.decompress(bytes(out[5:5 + n - 4]), 0, n - 4, back, 0, len(data))
    assert str(e.value).startswith("Truncated LZ4 frame: missing block size")
    with pytest.raises(A.IllegalArgumentException):
        comp.compress(data, 0, len(data), bytearray(20), 0, 20)
    with pytest.raises(A.IllegalArgumentException):
        comp.max_compressed_length(-1)


def test_full_size_property(gb, o):
    """512 frames of ~1.2 MiB (the corpus sample rotated): encode on the GPU, decode on the GPU, every plaintext restored"""
    import hashlib
    sample = b"".join(d for _, d, _ in common.corpus_sample())
    inputs_ = [sample[(i * 4099) % 65536:] + sample[:(i * 4099) % 65536] for i in range(64)]
    caps = [o.max_compressed_length("lz4frame", len(b)) for b in inputs_]
    frames, status, _ = gb.run(OP_COMPRESS, inputs_, caps)
    assert all(s == 0 for s in status)
    assert frames[0] == o.compress("lz4frame", inputs_[0]) and frames[37] == o.compress("lz4frame", inputs_[37])
    plain, status, err = gb.run(OP_DECOMPRESS, frames * 8, [len(b) for b in inputs_] * 8)
    assert all(s == 0 for s in status)
    want = [hashlib.sha256(b).digest() for b in inputs_] * 8
    assert [hashlib.sha256(p).digest() for p in plain] == want


def test_writer_with_more_frames_than_resident_wavefronts(gb, o):
    """Round 4: the frame writer runs one wavefront (and one 4 MiB slab of scratch) per resident slot, up to 2 304 of them, and its items are drawn
    from a counter.  3 000 frames -- more than one draw per wavefront, more than round 3's 512 slots -- of sizes around the piece / block borders
    must be byte-identical to the Java writer's and decode to the plaintext."""
    import hashlib
    base = b"".join(d for _, d, _ in common.corpus_sample())
    sizes = [1, 11, 12, 13, 64, 4096, 65535, 65536, 65537, 100000]
    blocks = [base[(i * 7919) % 50000:(i * 7919) % 50000 + sizes[i % len(sizes)]] for i in range(3000)]
    caps = [o.max_compressed_length("lz4frame", len(b)) for b in blocks]
    outs, status, _ = gb.run(OP_COMPRESS, blocks, caps)
    assert all(s == 0 for s in status)
    want = {}
    for i in range(0, 3000, 37):  # (every distinct size several times; the oracle is the slow side)
        key = (len(blocks[i]), hashlib.sha256(blocks[i]).digest())
        if key not in want:
            want[key] = o.compress("lz4frame", blocks[i])
        assert outs[i] == want[key], i
    plain, status, _ = gb.run(OP_DECOMPRESS, outs, [len(b) for b in blocks])
    assert all(s == 0 for s in status) and plain == blocks
