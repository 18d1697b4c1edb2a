"""GPU parity tests (through the C ABI) for Hadoop LZ4 / Snappy block streams (second half): byte-identical streams
on encode, every branch of the two readers (blocks of several chunks, empty blocks, -1 lengths, truncations, the streams' own buffer
when the destination has less room than a block declares), corruption with the oracle's status / offset, both reader paths (chunk
list through the batched block decoders; one wavefront per stream), non-default buffer sizes, a full-size property."""
import os
import struct

import numpy as np
import pytest

from tests import common, oracle_lib
from tests.oracle_lib import OracleError

pytestmark = pytest.mark.gpu
OPS = {"lz4": (10, 11), "snappy": (12, 13)}  # (decompress, compress)
BUF = 262144


# every reader variant runs: 3 the default (chunk list, the block decoder chosen by a probe of the sequence lengths: rings for long copies, the
# two-pass decoders for text), 1 / 2 the chunk list always through the rings / the two-pass decoders, 0 a wavefront per stream (the fallback of all)
_VARIANTS = [3, 1, 2, 0]


@pytest.fixture(scope="module", params=_VARIANTS, ids=["chunk-list-auto", "chunk-list-rings", "chunk-list-two-pass", "wave-per-stream"])
def gb(request):
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0, options={"hadoop.decompress.variant": request.param})


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def be(v):
    return struct.pack(">i", v)


def stream(codec, o, pieces):
    out = b""
    for declared, plain in pieces:
        c = o.compress(codec, plain)
        out += (be(declared) if declared is not None else b"") + be(len(c)) + c
    return out


def expect(o, codec, data, cap, buf):
    try:
        return 0, 0, o.hadoop_decompress(codec, data, cap, buf)
    except OracleError as e:
        return e.status, e.offset, None


def check_decode(gb, o, codec, cases, buf=BUF, unaligned=False):
    gb.set_option("hadoop.buffer_size", buf)
    outs, status, err = gb.run(OPS[codec][0], [c for c, _ in cases], [cap for _, cap in cases], unaligned=unaligned)
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = expect(o, codec, c, cap, buf)
        assert status[i] == est, "case %d: gpu status %d (offset %d) oracle %d (offset %d)" % (i, status[i], err[i], est, eoff)
        if est != 0:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)
        else:
            assert outs[i] == eout, "case %d" % i


def inputs(o):
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()[:6]] + common.synthetic_blocks(3, 10)
    blocks.append(b"".join(d for _, d, _ in common.corpus_sample()[:12]))  # 768 KiB: several chunks at the default buffer size
    blocks.append(o.random_generator(0.5)[:600000].tobytes())
    blocks.append(o.random_generator(1.0)[:300000].tobytes())            # incompressible: chunks at their bound
    base = common.corpus_sample()[0][1]
    blocks += [base[:n] for n in (1, 2, 15, 16, 17, 255, 4095)]
    return blocks


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
@pytest.mark.parametrize("buf", [BUF, 70000, 1024])
def test_encode_is_byte_identical_and_round_trips(gb, o, codec, buf):
    blocks = inputs(o)
    gb.set_option("hadoop.buffer_size", buf)
    caps = [o.hadoop_max_compressed_length(codec, len(b), buf) for b in blocks]
    outs, status, _ = gb.run(OPS[codec][1], blocks, caps, unaligned=True)
    assert (np.asarray(status) == 0).all(), status
    for b, c in zip(blocks, outs):
        assert c == o.hadoop_compress(codec, b, buf), len(b)
    check_decode(gb, o, codec, [(c, len(b)) for b, c in zip(blocks, outs)] + [(c, len(b) + 100) for b, c in zip(blocks, outs)], buf)
    # a destination below the bound this API asks for
    outs, status, _ = gb.run(OPS[codec][1], blocks[-3:], [c - 1 for c in caps[-3:]])
    assert all(oracle_lib.status_class(s) == 2 and oracle_lib.status_detail(s) == 110 for s in status)
    assert gb.codec.lib.achip_hadoop_max_compressed_length(0 if codec == "lz4" else 1, 700000, buf) == o.hadoop_max_compressed_length(codec, 700000, buf)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_reader_branches(gb, o, codec):
    a, b, c = b"hello hello hello hello " * 40, b"abcdefgh" * 300, bytes(range(256)) * 3
    multi = stream(codec, o, [(len(a) + len(b), a), (None, b)]) + be(0) + be(0) + stream(codec, o, [(len(c), c)]) + be(0)
    good = stream(codec, o, [(len(a), a)])
    cases = [(multi, len(a + b + c)), (multi, len(a + b + c) + 1000)]
    cases += [(multi, cap) for cap in (len(a + b + c) - 1, len(a) + 5, len(a), 10, 0)]                  # All input was not consumed
    cases += [(good + be(-1), len(a) + 10), (good + be(50) + be(-1), len(a) + 10), (good + be(-1) + b"garbage", len(a) + 10), (good + be(77), len(a))]
    cases += [(good[:cut], len(a)) for cut in (0, 2, 4, 6, 8, 9, len(good) - 1)]                        # truncated ints / chunk data
    cases += [(be(10) + be(-5) + b"xxxxx", 100), (be(-7) + good[4:], len(a)), (be(0) * 5, 0), (b"", 0), (b"", 5)]
    for flip in (8, 11, 20, len(good) - 2):                                                             # corrupt chunks: the block codec's own verdict
        bad = bytearray(good + good)
        bad[flip] ^= 0xA5
        cases += [(bytes(bad), 2 * len(a)), (bytes(bad), 2 * len(a) + 17)]
    if codec == "snappy":
        cases += [(stream("snappy", o, [(len(a) - 1, a)]), len(a)),                                       # chunk announces more than its block has left
                  (good + be(5) + be(1) + b"\x00", 3 * len(a)), (good + be(5) + be(1) + b"\x00" + good, 3 * len(a)),  # a chunk of no bytes
                  (be(5) + be(0), 10)]                                                                    # a chunk that ends inside its preamble
    else:
        cases += [(stream("lz4", o, [(5, a), (None, b)]), len(a + b)),                                    # chunks beyond the declared block length
                  (stream("lz4", o, [(len(a) + 100, a)]), len(a) + 50), (stream("lz4", o, [(len(a) + 100, a)]), len(a) - 1),  # through the stream's own buffer
                  (be(5) + be(0), 10)]
    check_decode(gb, o, codec, cases)
    check_decode(gb, o, codec, cases, unaligned=True)
    if codec == "lz4":  # ... whose capacity (bufferSize + 8) decides what the block decoder says
        check_decode(gb, o, codec, [(stream("lz4", o, [(len(a) + 100, a)]), len(a) + 50), (good, len(a))], buf=256)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_random_corruption_reports_the_oracles_status(gb, o, codec):
    rng = np.random.default_rng(11)
    data = b"".join(d for _, d, _ in common.corpus_sample()[:3])
    buf = 50000
    s = o.hadoop_compress(codec, data, buf)
    cases = []
    for _ in range(60):
        m = bytearray(s)
        for _ in range(int(rng.integers(1, 3))):
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        cases += [(bytes(m), len(data)), (bytes(m[:int(rng.integers(0, len(m)))]), len(data))]
    check_decode(gb, o, codec, cases, buf)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_twins_and_host_api(o, codec):
    import aircompressor_amd as A
    data = b"".join(d for _, d, _ in common.corpus_sample()[:6])
    C, D = (A.Lz4HadoopHipCompressor, A.Lz4HadoopHipDecompressor) if codec == "lz4" else (A.SnappyHadoopHipCompressor, A.SnappyHadoopHipDecompressor)
    for buf in (BUF, 100000):
        c, d = C(buffer_size=buf), D(buffer_size=buf)
        out = bytearray(c.max_compressed_length(len(data)))
        n = c.compress(data, 0, len(data), out, 0, len(out))
        assert bytes(out[:n]) == o.hadoop_compress(codec, data, buf)
        back = bytearray(len(data))
        assert d.decompress(bytes(out[:n]), 0, n, back, 0, len(back)) == len(data) and bytes(back) == data
        with pytest.raises(A.IllegalArgumentException):   # All input was not consumed
            d.decompress(bytes(out[:n]), 0, n, bytearray(len(data) - 1), 0, len(data) - 1)
        with pytest.raises(A.MalformedInputException) as e:
            d.decompress(bytes(out[:n - 1]), 0, n - 1, back, 0, len(back))
        assert "encountered EOF while reading block data" in str(e.value)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_full_size_round_trip_property(codec):
    """1024 streams x 1 MiB on the device: encode, decode, compare (no oracle at this size)."""
    import torch
    import aircompressor_amd as A
    codec_h = A.HipBatchCodec(0)
    dev = torch.device("cuda", 0)
    n, size = 1024, 1 << 20
    g = torch.Generator(device=dev); g.manual_seed(7)
    frag = torch.randint(0, 256, (n * size // 100 + 1, 37), dtype=torch.uint8, device=dev, generator=g)
    plain = frag.repeat(1, 3)[:, :100].reshape(-1)[:n * size].contiguous()
    cap = codec_h.lib.achip_hadoop_max_compressed_length(0 if codec == "lz4" else 1, size, BUF)
    i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
    so = torch.arange(n, **i64) * size; sl = torch.full((n,), size, **i32)
    comp = torch.zeros(n * cap, dtype=torch.uint8, device=dev); co = torch.arange(n, **i64) * cap; cc = torch.full((n,), cap, **i32)
    cl = torch.zeros(n, **i32); st = torch.zeros(n, **i32); eo = torch.zeros(n, **i64)
    dop, cop = (A.OP_LZ4HADOOP_DECOMPRESS, A.OP_LZ4HADOOP_COMPRESS) if codec == "lz4" else (A.OP_SNAPPYHADOOP_DECOMPRESS, A.OP_SNAPPYHADOOP_COMPRESS)
    torch.cuda.synchronize()  # (the buffers above are filled on torch's stream)
    codec_h.launch(cop, plain, so, sl, comp, co, cc, cl, st, eo, n); codec_h.synchronize()
    assert int((st != 0).sum()) == 0 and int(cl.min()) > 0
    back = torch.zeros(n * size, dtype=torch.uint8, device=dev); bl = torch.zeros(n, **i32)
    torch.cuda.synchronize()
    codec_h.launch(dop, comp, co, cl, back, so, sl, bl, st, eo, n); codec_h.synchronize()
    assert int((st != 0).sum()) == 0 and bool((bl == size).all()) and bool((back == plain).all())
