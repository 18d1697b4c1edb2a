"""GPU parity tests (through the C ABI) for x-snappy-framed streams: the reference's own cases
(T/snappy/TestSnappyStream.java:50-312) rebuilt, byte-identical streams on encode, systematic corruption with the oracle's
status / offset, chunks beyond 64 KiB, and the wavefront CRC-32C against the oracle at every length class."""
import struct

import os

import numpy as np
import pytest

from tests import common, oracle_lib
from tests.oracle_lib import OracleError

pytestmark = pytest.mark.gpu
OP_DECOMPRESS, OP_COMPRESS = 8, 9
HEADER = bytes([0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59])


# every reader variant runs: 3 the default (chunk list, rings or two-pass decoder by a probe of the element lengths), 1 / 2 always the rings /
# the two-pass decoder, 0 a wavefront per stream
_VARIANTS = [3, 1, 2, 0]


@pytest.fixture(scope="module", params=_VARIANTS, ids=["block-lists-auto", "block-lists-rings", "block-lists-two-pass", "wave-per-stream"])
def gb(request):
    """reader / writer under test: chunk list + batched block decoders, block list + two-tier block encoder + compaction (defaults);
    one wavefront per stream (their fallback)"""
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0, options={"snappyframed.decompress.variant": request.param, "snappyframed.compress.variant": min(request.param, 1)})


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def expect(o, data, cap):
    try:
        return 0, 0, o.decompress("snappyframed", data, cap)
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


def chunk(flag, data, crc=None, o=None, plain=None):
    crc = o.crc32c(plain if plain is not None else data, masked=True) if crc is None else crc
    return bytes([flag]) + struct.pack("<I", len(data) + 4)[:3] + struct.pack("<I", crc) + data


def inputs(o):
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()[:8]] + common.synthetic_blocks(3, 18)
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()[:5]))                    # 320 KiB: five full blocks
    blocks.append(o.random_generator(0.5)[:500000].tobytes())                             # testLargeWrites :279-312
    blocks.append(o.random_generator(1.0)[:5000].tobytes())                               # testUncompressible :81-100
    blocks.append(bytes(np.random.default_rng(1).integers(0, 256, 70000, dtype=np.uint8)))  # raw chunk + raw remainder
    blocks.append(b"aaaaaaaaaaaabbbbbbbaaaaaa")                                           # testSimple :50-79
    base = common.corpus_sample()[0][1]
    blocks += [base[:n] for n in (1, 2, 3, 4, 5, 255, 256, 257, 511, 512, 513, 4095, 4096, 4097, 65535)]
    blocks.append(base + base[:1])                                                        # 65537: a one-byte last block
    return blocks


def test_compress_is_byte_identical_and_round_trips(gb, o):
    blocks = inputs(o)
    caps = [o.max_compressed_length("snappyframed", len(b)) for b in blocks]
    outs, status, _ = gb.run(OP_COMPRESS, blocks, caps, unaligned=True)
    assert all(s == 0 for s in status), status
    for i, (b, z) in enumerate(zip(blocks, outs)):
        assert z == o.compress("snappyframed", b), "stream %d (%d bytes)" % (i, len(b))
    simple = outs[blocks.index(b"aaaaaaaaaaaabbbbbbbaaaaaa")]
    assert len(simple) == 37 and simple[10:18] == bytes([0x00, 0x17, 0, 0, 0xA8, 0xCD, 0x74, 0x92])  # T/snappy/TestSnappyStream.java:60-78
    plain, status, err = gb.run(OP_DECOMPRESS, outs, [len(b) for b in blocks], unaligned=True)
    for i, (b, p, s) in enumerate(zip(blocks, plain, status)):
        assert s == 0 and p == b, (i, s, err[i])


def test_reference_error_cases_and_chunk_kinds(gb, o):
    cases = [(HEADER, 0), (HEADER, 16), (HEADER[:9], 16), (b"", 16), (b"\xff\x06\x00\x00sNaPpX", 16)]
    cases += [(HEADER + bytes(blk), 1024) for blk in (
        [0],                                                # testShortBlockHeader :111-117
        [1, 8, 0, 0, 0, 0, 0, 0, ord("x"), ord("x")],       # testShortBlockData :119-126
        [1, 4, 0, 0, 0, 0, 0, 0],                           # testInvalidBlockSizeZero :151-158
        [1, 5, 0, 0, 0, 0, 0, 0, ord("a")],                 # testInvalidChecksum :160-167
        [0xff, 5, 0, 0, 1, 2, 3, 4, 5], [0, 4, 0, 0, 1, 2, 3, 4], [0, 5, 0, 0, 0, 0, 0, 0, 0x80])]
    cases += [(HEADER + bytes([flag, 5, 0, 0, 0, 0, 0, 0, 0]), 16) for flag in range(2, 0xff)]  # :128-149: unskippable / skippable
    a = b"some plain bytes, twice: some plain bytes"
    good = chunk(1, a, o=o)
    cases += [(HEADER + good, len(a)), (HEADER + good, len(a) - 1), (HEADER + HEADER + good + HEADER + good, 2 * len(a)),
              (HEADER + bytes([0x80, 200, 0, 0]) + b"x" * 10, 16),                       # a skippable chunk running off the end: skipped quietly
              (HEADER + bytes([0x80, 3, 0, 0]) + b"xyz" + good, len(a)),
              (HEADER + chunk(0, o.compress("snappy", a), o=o, plain=a), len(a)),
              (HEADER + chunk(0, o.compress("snappy", a), crc=1), len(a)),
              (HEADER + chunk(0, o.compress("snappy", a)[:-3], o=o, plain=a), len(a))]
    for size, compressed in ((100000, False), (500000, True), (100000, True)):             # testLargerFrames_* :178-268
        random = o.random_generator(0.5)[:size].tobytes()
        data = o.compress("snappy", random) if compressed else random
        cases.append((HEADER + chunk(0 if compressed else 1, data, o=o, plain=random), size))
        cases.append((HEADER + chunk(0 if compressed else 1, data, o=o, plain=random), size - 1))
    check_decode(gb, o, cases)
    check_decode(gb, o, cases, unaligned=True)


def test_corrupted_streams(gb, o):
    rng = np.random.default_rng(5)
    cases = []
    for b in [d for _, d, _ in common.corpus_sample()[:3]] + [o.random_generator(0.5)[:150000].tobytes()]:
        c = bytearray(o.compress("snappyframed", b))
        cases += [(bytes(c[:n]), len(b)) for n in (9, 10, 11, 13, 14, 17, 18, 19, len(c) // 2, len(c) - 1)]
        for _ in range(24):
            m = bytearray(c)
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cases.append((bytes(m), len(b)))
        for _ in range(6):  # hits in the headers
            m = bytearray(c)
            m[int(rng.integers(0, 24))] ^= 1 << int(rng.integers(0, 8))
            cases.append((bytes(m), len(b)))
    check_decode(gb, o, cases, unaligned=True)


def test_wave_crc32c_at_every_length_class(gb, o):
    """a raw chunk of every length around the 256-byte rows and 4-byte words the wavefront CRC works in: decodes only with the right CRC"""
    base = b"".join(d for _, d, _ in common.corpus_sample()[:2])
    sizes = list(range(1, 20)) + [252, 253, 254, 255, 256, 257, 258, 259, 260, 511, 512, 513, 767, 768, 1023, 1024, 1025, 4096, 65535, 65536, 65537, 100001]
    good = [(HEADER + chunk(1, base[7:7 + n], o=o), n) for n in sizes]
    bad = [(HEADER + chunk(1, base[7:7 + n], crc=(o.crc32c(base[7:7 + n], masked=True) ^ 0x80) & 0xFFFFFFFF), n) for n in sizes]
    check_decode(gb, o, good + bad, unaligned=True)


def test_host_api_twins(o):
    """SnappyFramedHipCompressor / SnappyFramedHipDecompressor: the Compressor / Decompressor call shapes over host buffers"""
    import aircompressor_amd as A
    comp, decomp = A.SnappyFramedHipCompressor(), A.SnappyFramedHipDecompressor()
    data = common.corpus_sample()[1][1] + common.corpus_sample()[2][1][:1234]
    cap = comp.max_compressed_length(len(data))
    assert cap == o.max_compressed_length("snappyframed", len(data)) == 10 + 2 * 8 + len(data)
    out = bytearray(cap + 5)
    n = comp.compress(data, 0, len(data), out, 5, cap)
    assert bytes(out[5:5 + n]) == o.compress("snappyframed", data)
    back = bytearray(len(data))
    assert decomp.decompress(out, 5, n, back, 0, len(data)) == len(data) and bytes(back) == data
    with pytest.raises(A.MalformedInputException) as e:
        decomp.decompress(bytes(out[5:5 + n - 3]), 0, n - 3, back, 0, len(data))
    assert str(e.value).startswith("unexpectd EOF when reading frame")
    corrupt = bytearray(out[5:5 + n])
    corrupt[14] ^= 1  # the first chunk's stored CRC
    with pytest.raises(A.MalformedInputException) as e:
        decomp.decompress(bytes(corrupt), 0, n, back, 0, len(data))
    assert str(e.value).startswith("Corrupt input: invalid checksum")
    with pytest.raises(A.IllegalArgumentException):
        comp.compress(data, 0, len(data), bytearray(cap - 1), 0, cap - 1)
    with pytest.raises(A.IllegalArgumentException):
        comp.max_compressed_length(-1)


def test_full_size_property(gb, o):
    """1024 streams of 1 MiB: encode -> decode restores the plaintext; every stream equals the oracle's for a sample"""
    import torch
    import aircompressor_amd as A
    dev = torch.device("cuda", 0)
    codec = gb.codec
    n, size = 1024, 1 << 20
    sample = torch.from_numpy(np.frombuffer(b"".join(d for _, d, _ in common.corpus_sample()), dtype=np.uint8).copy()).to(dev)
    plain = sample.repeat((n * size + sample.numel() - 1) // sample.numel())[:n * size].contiguous()
    cap = codec.lib.achip_snappyframed_max_compressed_length(size)
    cs = (cap + 15) // 16 * 16
    i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
    so = torch.arange(n, **i64) * size; sl = torch.full((n,), size, **i32)
    comp = torch.empty(n * cs + 64, dtype=torch.uint8, device=dev); co = torch.arange(n, **i64) * cs; cc = torch.full((n,), cap, **i32)
    cl = torch.zeros(n, **i32); st = torch.zeros(n, **i32); eo = torch.zeros(n, **i64)
    torch.cuda.synchronize()
    codec.launch(A.OP_SNAPPYFRAMED_COMPRESS, plain, so, sl, comp, co, cc, cl, st, eo, n); codec.synchronize()
    assert int((st != 0).sum()) == 0
    back = torch.empty(n * size + 64, dtype=torch.uint8, device=dev); bl = torch.zeros(n, **i32)
    codec.launch(A.OP_SNAPPYFRAMED_DECOMPRESS, comp, co, cl, back, so, sl, bl, st, eo, n); codec.synchronize()
    assert int((st != 0).sum()) == 0 and int((bl != size).sum()) == 0
    assert bool((back[:n * size] == plain).all())
    for i in (0, 1, 517, n - 1):
        z = comp[i * cs:i * cs + int(cl[i])].cpu().numpy().tobytes()
        assert z == o.compress("snappyframed", plain[i * size:(i + 1) * size].cpu().numpy().tobytes())
