"""GPU parity tests (through the C ABI) for the Zstd frame decoder against the CPU oracle and the reference's
golden decoder fixtures (T/zstd/AbstractTestZstd.java:41-78,175-184).  Frames with diverse features come from
libzstd via pyarrow (a third-party encoder, like zstd-jni in T/zstd/TestZstd.java:21-47)."""
import numpy as np
import pytest

from tests import common, oracle_lib
from tests.oracle_lib import OracleError

pytestmark = pytest.mark.gpu
OP_ZSTD_DECOMPRESS = 4


@pytest.fixture(scope="module")
def gb():
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0)


@pytest.fixture(scope="module", params=[1, 0], ids=["pipeline", "one-kernel"])
def gbd(request):
    """decoder under test: the five-stage pipeline (default) and the one-kernel decoder it falls back to"""
    from tests.gpu_harness import GpuBatch
    g = GpuBatch(0, options={"zstd.decompress.variant": request.param})
    g.variant = request.param
    return g


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


DEFAULT_SEQ_WAVES = 1  # (the library's default of zstd.decompress.seq_waves, restored by the test that changes it)
DEFAULT_LIT_ITEMS = 13  # (... of zstd.decompress.lit_items)


def zstd_frames(blocks, level):
    import pyarrow as pa
    codec = pa.Codec("zstd", compression_level=level)
    return [codec.compress(b, asbytes=True) if len(b) else codec.compress(b"", asbytes=True) for b in blocks]


def plain_blocks():
    sample = [d for _, d, _ in common.corpus_sample()]
    blocks = list(sample)
    blocks += [sample[i] + sample[i + 1] for i in range(0, len(sample) - 1, 2)]  # 128 KiB: one full zstd block
    blocks += common.synthetic_blocks(21, 18)
    blocks += [d for _, d in common.HAND_CASES]
    return blocks


def test_golden_fixtures(gbd, o):
    z1, p1 = common.golden_zstd("with-checksum.zst"), common.golden_zstd("with-checksum")
    z2, p2 = common.golden_zstd("multiple-frames.zst"), common.golden_zstd("multiple-frames")
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, [z1, z2, z1, z2], [len(p1), len(p2), len(p1) + 500, len(p2) + 1], unaligned=True)
    assert status == [0, 0, 0, 0], (status, err)
    assert outs[0] == p1 and outs[2] == p1 and outs[1] == p2 and outs[3] == p2


def _expect(o, data, cap):
    try:
        return 0, 0, o.decompress("zstd", data, cap)
    except OracleError as e:
        return e.status, e.offset, None


def test_error_fixtures_and_corruptions(gbd, o):
    rng = np.random.default_rng(5)
    z1, p1 = common.golden_zstd("with-checksum.zst"), common.golden_zstd("with-checksum")
    cases = [
        (common.golden_zstd("offset-before-start.zst"), 1 << 20),  # "Input is corrupted"
        (common.golden_zstd("bad-second-frame.zst"), 1 << 20),      # "Invalid magic prefix"
        (z1, len(p1) - 1),                                          # "Output buffer too small"
        (z1[:-1], len(p1)), (z1[:-4], len(p1)), (z1[:100], len(p1)), (z1[:3], 10), (b"", 10), (z1, 0),
        (b"\x27\xb5\x2f\xfd" + z1[4:], len(p1)),                    # v0.7 magic
        (z1[:4] + bytes([z1[4] | 1]) + z1[5:], len(p1)),            # dictionary id flag
    ]
    bad_sum = bytearray(z1)
    bad_sum[-2] ^= 0x40
    cases.append((bytes(bad_sum), len(p1)))
    frames = zstd_frames([d for _, d, _ in common.corpus_sample()[:5]], 3)
    plains = [d for _, d, _ in common.corpus_sample()[:5]]
    for z, p in zip(frames, plains):
        cases.append((z, len(p) - 7))
        for cut in (len(z) // 3, len(z) - 2):
            cases.append((z[:cut], len(p)))
        for _ in range(10):
            m = bytearray(z)
            for _ in range(int(rng.integers(1, 3))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            cases.append((bytes(m), len(p)))
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, [c for c, _ in cases], [cap for _, cap in cases])
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = _expect(o, c, cap)
        assert status[i] == est, "case %d: gpu status %d oracle %d (gpu offset %d, oracle %d)" % (i, status[i], est, err[i], eoff)
        if est == 0:
            assert outs[i] == eout, "case %d" % i
        else:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)


def test_offsets_beyond_28_bits_are_rejected(gbd, o):
    """Offset code 28 expresses offsets up to 0x1FFFFFFC.  A hand-made 20-byte frame (one raw literal, one sequence, RLE tables: literal
    length code 0, offset code 28, match length code 3) whose offset is 0x10000000 + the extra bits: "Input is corrupted" at offset 16
    in the Java decoder (the match starts before the output).  Round 1's one-kernel decoder kept 28 bits of the offset -- such an
    offset wrapped to 0 and the pointer-jumping loop never ended.  The extra bits are varied over the wrap point."""
    base = bytes.fromhex("28b52ffd20045d000008410154001c0003000010")
    cases = [base]
    for b0 in (0, 1, 2, 3, 4, 8, 0x80, 0xFF):
        for b1 in (0, 0x40):
            for b2 in (0x10, 0x11, 0x14, 0x18, 0x1F):
                cases.append(base[:-3] + bytes([b0, b1, b2]))
    for cap in (4, 1 << 16):
        outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, cases, [cap] * len(cases))
        for i, c in enumerate(cases):
            est, eoff, eout = _expect(o, c, cap)
            assert (status[i], err[i] if est else 0) == (est, eoff), "case %d: gpu %d@%d oracle %d@%d" % (i, status[i], err[i], est, eoff)
            if est == 0:
                assert outs[i] == eout


@pytest.mark.parametrize("level", [1, 3, 9])
def test_libzstd_frames_decode_to_plaintext(gbd, o, level):
    blocks = plain_blocks()
    frames = zstd_frames(blocks, level)
    for pad in (0, 64):
        outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, frames, [len(b) + pad for b in blocks], unaligned=(pad == 0))
        for i, (b, p, s) in enumerate(zip(blocks, outs, status)):
            assert s == 0, (i, len(b), s, err[i])
            assert p == b, "block %d (len %d)" % (i, len(b))
    for b, z in list(zip(blocks, frames))[:8]:
        assert o.decompress("zstd", z, len(b)) == b


def test_multi_block_frames_and_concatenated_frames(gbd, o):
    whole = b"".join(d for _, d, _ in common.corpus_sample())  # ~1.2 MB: ten 128 KiB blocks with cross-block history
    frames = zstd_frames([whole, whole[:300000], whole[100000:100000 + 131073]], 3)
    cat = frames[1] + frames[2] + zstd_frames([b""], 3)[0] + frames[1]
    plain_cat = whole[:300000] + whole[100000:100000 + 131073] + whole[:300000]
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, frames + [cat], [len(whole), 300000, 131073, len(plain_cat)])
    assert status == [0, 0, 0, 0], (status, err)
    assert outs[0] == whole and outs[1] == whole[:300000] and outs[2] == whole[100000:100000 + 131073] and outs[3] == plain_cat
    assert o.decompress("zstd", cat, len(plain_cat)) == plain_cat


def frame_blocks(f):
    """number of blocks of the (single) frame f"""
    fhd = f[4]
    pos = 5 + (0 if fhd & 0x20 else 1) + ((1 if fhd & 0x20 else 0) if fhd >> 6 == 0 else 1 << (fhd >> 6))
    n = 0
    while True:
        h = int.from_bytes(f[pos:pos + 3], "little")
        pos += 3 + (1 if (h >> 1) & 3 == 1 else h >> 3)
        n += 1
        if h & 1:
            return n


@pytest.mark.timeout(900, method="thread")  # (the default pass size was last changed after the round's GPU minutes: CPU-emulator-verified)
@pytest.mark.parametrize("stream_blocks", [65536, 16])
def test_multi_block_frames_take_the_multi_block_stages(o, stream_blocks):
    """frames of several blocks (ZstdOutputStream / ZstdFrameCompressor / libzstd beyond 128 KiB) go through the
    pipeline's multi-block stages -- window, repeat offsets, Huffman and FSE tables carried from block to block -- and not to the
    one-kernel decoder; with the smallest passes (room for 16 blocks of 128 KiB, 128 block slots) the batch takes several passes and a
    frame of more blocks than a pass has slots falls back."""
    from tests.gpu_harness import GpuBatch
    g = GpuBatch(0, options={"zstd.decompress.stream_blocks": stream_blocks})
    plains = common.multi_block_plains()
    encoders = [("oracle", lambda b: o.compress("zstd", b))] + [("libzstd-%d" % l, (lambda l: lambda b: zstd_frames([b], l)[0])(l)) for l in (1, 3, 9, 19)]
    unfit = 0
    for name, enc in encoders:
        frames = [enc(b) for b in plains]
        for pad in (0, 41):
            outs, status, err = g.run(OP_ZSTD_DECOMPRESS, frames, [len(b) + pad for b in plains], unaligned=(pad == 0))
            assert all(s == 0 for s in status), (name, status, err)
            for i, (b, got) in enumerate(zip(plains, outs)):
                assert got == b, "%s frame %d (len %d)" % (name, i, len(b))
        fits = sum(1 for f in frames if frame_blocks(f) <= 8 * stream_blocks)
        assert g.codec.native.get_stat("zstd.decompress.multiblock_items") == len(frames), name
        assert g.codec.native.get_stat("zstd.decompress.multiblock_fast_items") == fits, name
        assert g.codec.native.get_stat("zstd.decompress.fallback_items") == len(frames) - fits, name
        assert g.codec.native.get_stat("zstd.decompress.multiblock_blocks") == sum(frame_blocks(f) for f in frames if frame_blocks(f) <= 8 * stream_blocks), name
        unfit += len(frames) - fits
    assert (unfit > 0) == (stream_blocks == 16)  # (libzstd level 3 on the mixed data: ~260 blocks)


def test_damaged_multi_block_frames_report_what_the_oracle_reports(gbd, o):
    """whatever the multi-block stages make of a damaged frame -- decode it (the damage may be harmless, or only change bytes) or hand
    it to the one-kernel decoder -- the caller sees the oracle's output, status and offset"""
    rng = np.random.default_rng(23)
    plains = common.multi_block_plains()[1:6]
    cases = []
    for name, enc in (("oracle", lambda b: o.compress("zstd", b)), ("libzstd-3", lambda b: zstd_frames([b], 3)[0])):
        for b in plains:
            f = enc(b)
            for k in range(18):
                m = bytearray(f)
                kind = k % 6
                if kind == 0:
                    m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:
                    m[int(rng.integers(4, min(40, len(m))))] = int(rng.integers(0, 256))
                elif kind == 2:
                    m = m[:int(rng.integers(8, max(9, len(m))))]
                elif kind == 3:
                    m += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tolist())
                elif kind == 4:
                    at = int(rng.integers(0, max(1, len(m) - 16)))
                    m[at:at + 16] = bytes(min(16, len(m) - at))
                else:
                    m[len(m) - 1 - int(rng.integers(0, 6))] ^= 0x10
                cases.append((bytes(m), len(b) if k % 3 else len(b) - int(rng.integers(1, 5000))))
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, [c for c, _ in cases], [cap for _, cap in cases])
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = _expect(o, c, cap)
        assert status[i] == est, "case %d: gpu status %d oracle %d (gpu offset %d, oracle %d)" % (i, status[i], est, err[i], eoff)
        if est == 0:
            assert outs[i] == eout, "case %d" % i
        else:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)


def test_pipeline_small_tiles(o):
    """a batch larger than the pipeline's tile goes through it in several passes (here: tiles of 64 items)"""
    from tests.gpu_harness import GpuBatch
    g = GpuBatch(0, options={"zstd.decompress.tile": 64})
    blocks = [b for b in plain_blocks() if 0 < len(b) <= 131072]
    blocks = (blocks * 6)[:300]
    frames = zstd_frames(blocks, 3)
    outs, status, err = g.run(OP_ZSTD_DECOMPRESS, frames, [len(b) for b in blocks])
    assert all(s == 0 for s in status), status
    assert outs == blocks
    assert g.codec.native.get_stat("zstd.decompress.fallback_items") <= 6 * 12  # only the raw-block / tiny frames


def test_single_block_host_api(o):
    import aircompressor_amd as A
    z1, p1 = common.golden_zstd("with-checksum.zst"), common.golden_zstd("with-checksum")
    d = A.ZstdHipDecompressor()
    out = bytearray(len(p1) + 9)
    n = d.decompress(z1, 0, len(z1), out, 9, len(p1))
    assert n == len(p1) and bytes(out[9:]) == p1
    assert d.get_decompressed_size(z1, 0, len(z1)) == -1
    with pytest.raises(A.MalformedInputException) as e:
        d.decompress(common.golden_zstd("offset-before-start.zst"), 0, 1559, bytearray(1 << 16), 0, 1 << 16)
    assert str(e.value).startswith("Input is corrupted")
    with pytest.raises(A.MalformedInputException) as e:
        z = common.golden_zstd("bad-second-frame.zst")
        d.decompress(z, 0, len(z), bytearray(1 << 16), 0, 1 << 16)
    assert str(e.value).startswith("Invalid magic prefix")


def test_many_frames_full_size_property(gbd, o):
    """1024 x 128 KiB frames (libzstd level 3) -> GPU decode must restore every block (checked by hash)."""
    import hashlib
    rng = np.random.default_rng(77)
    sample = b"".join(d for _, d, _ in common.corpus_sample())
    blocks = []
    for i in range(256):
        off = int(rng.integers(0, len(sample) - 131072))
        blocks.append(sample[off:off + 131072])
    frames = zstd_frames(blocks, 3)
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, frames * 4, [131072] * 1024)
    assert all(s == 0 for s in status)
    want = [hashlib.sha256(b).digest() for b in blocks] * 4
    assert [hashlib.sha256(p).digest() for p in outs] == want
    if gbd.variant == 1:
        # single-block frames must stay on the pipeline: none handed to the one-kernel decoder
        stages = [gbd.codec.native.get_stat("zstd.decompress.fallback_stage%d" % k) for k in range(1, 6)]
        assert gbd.codec.native.get_stat("zstd.decompress.fallback_items") == 0, stages


@pytest.mark.parametrize("waves", [1, 2, 4])
def test_sequence_stage_wavefronts_per_workgroup(o, waves):
    """zstd.decompress.seq_waves: the pipeline's sequence stage with its 64 items a workgroup on 1, 2 or 4 wavefronts (lanes without an item beside lanes
    with one; single-block frames and the multi-block stages) -- every frame restored, nothing handed to the one-kernel decoder."""
    import hashlib
    from tests.gpu_harness import GpuBatch
    rng = np.random.default_rng(78)
    sample = b"".join(d for _, d, _ in common.corpus_sample())
    blocks = []
    for i in range(250):
        off = int(rng.integers(0, len(sample) - 131072))
        blocks.append(sample[off:off + int(rng.integers(2000, 131073))])
    blocks += [sample[:300000], sample[100000:100000 + 500000]]  # (frames of several blocks: the multi-block stages' instantiation)
    frames = zstd_frames(blocks, 3)
    g = GpuBatch(0, options={"zstd.decompress.variant": 1})
    try:
        g.set_option("zstd.decompress.seq_waves", waves)
        outs, status, err = g.run(OP_ZSTD_DECOMPRESS, frames * 16, [len(b) for b in blocks] * 16)
        assert all(s == 0 for s in status)
        want = [hashlib.sha256(b).digest() for b in blocks] * 16
        assert [hashlib.sha256(p).digest() for p in outs] == want
        assert g.codec.native.get_stat("zstd.decompress.fallback_items") == 0
    finally:
        g.set_option("zstd.decompress.seq_waves", DEFAULT_SEQ_WAVES)  # (process-wide)


@pytest.mark.parametrize("items", [8, 10, 13, 16, 20])
def test_literal_stage_items_per_wavefront(o, items):
    """zstd.decompress.lit_items: the pipeline's literal stage with 8, 10 or 16 items a wavefront (4 KiB of LDS an item: 5 / 4 / 2 wavefronts a CU), with 13
    (the default: symbol bytes and length nibbles apart, 3 KiB an item) and with 20 (16 items of symbol bytes + lengths by symbol) -- every
    frame restored (libzstd's frames and the Java encoder's, single-block and multi-block), nothing handed to the one-kernel decoder."""
    import hashlib
    from tests.gpu_harness import GpuBatch
    rng = np.random.default_rng(79)
    sample = b"".join(d for _, d, _ in common.corpus_sample())
    blocks = []
    for i in range(203):  # (not a multiple of any of the three item counts)
        off = int(rng.integers(0, len(sample) - 131072))
        blocks.append(sample[off:off + int(rng.integers(2000, 131073))])
    blocks += [sample[:300000], sample[100000:100000 + 500000]]
    frames = zstd_frames(blocks, 3)
    frames[:60] = [o.compress("zstd", b) for b in blocks[:60]]
    g = GpuBatch(0, options={"zstd.decompress.variant": 1})
    try:
        g.set_option("zstd.decompress.lit_items", items)
        outs, status, err = g.run(OP_ZSTD_DECOMPRESS, frames * 8, [len(b) for b in blocks] * 8)
        assert all(s == 0 for s in status)
        want = [hashlib.sha256(b).digest() for b in blocks] * 8
        assert [hashlib.sha256(p).digest() for p in outs] == want
        assert g.codec.native.get_stat("zstd.decompress.fallback_items") == 0
    finally:
        g.set_option("zstd.decompress.lit_items", DEFAULT_LIT_ITEMS)  # (process-wide)


def test_pipeline_takes_java_encoded_frames(gbd, o):
    """Frames as ZstdFrameCompressor writes them (single segment, one block, checksum) stay on the fast path;
    corrupting one sends exactly that item to the one-kernel decoder, which reports the Java-exact error."""
    blocks = plain_blocks()
    blocks = [b for b in blocks if 0 < len(b) <= 131072]
    frames = [o.compress("zstd", b) for b in blocks]
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, frames, [len(b) for b in blocks])
    assert all(s == 0 for s in status), status
    assert outs == blocks
    if gbd.variant == 1:
        raw = sum(1 for f in frames if ((f[4 + 2 + (0 if f[4] >> 6 == 0 else (1 << (f[4] >> 6)) - 1)] >> 1) & 3) != 2)
        assert gbd.codec.native.get_stat("zstd.decompress.fallback_items") <= raw + sum(1 for b in blocks if len(b) < 16)
    big = max(range(len(blocks)), key=lambda i: len(blocks[i]))
    bad = bytearray(frames[big])
    bad[len(bad) // 2] ^= 0x10
    outs, status, err = gbd.run(OP_ZSTD_DECOMPRESS, [frames[big], bytes(bad), frames[big]], [len(blocks[big])] * 3)
    est, eoff, eout = _expect(o, bytes(bad), len(blocks[big]))
    assert status[0] == 0 and status[2] == 0 and outs[0] == blocks[big] and outs[2] == blocks[big]
    assert status[1] == est and (est == 0 or err[1] == eoff)
    if gbd.variant == 1:
        assert gbd.codec.native.get_stat("zstd.decompress.fallback_items") <= 1


# ---- encoder (level 3), rows a11-a14 -------------------------------------------------------------------------
OP_ZSTD_COMPRESS = 5


def encoder_inputs():
    sample = [d for _, d, _ in common.corpus_sample()]
    blocks = [d for _, d in common.HAND_CASES]
    blocks += sample
    blocks += [sample[i] + sample[i + 1] for i in range(0, len(sample) - 1, 2)]   # 128 KiB: one full block
    blocks += common.synthetic_blocks(31, 24)
    blocks.append(b"".join(sample[:5]) + b"tail")                                  # 3 blocks sharing tables / repcodes / Huffman reuse
    blocks.append(b"".join(sample[3:9]))                                           # > 256 KiB: default parameter row, 2^17 long table
    blocks.append(common.golden_zstd("large-rle"))
    blocks.append(common.golden_zstd("incompressible"))
    base = sample[0]
    blocks += [base[:n] for n in (1, 2, 6, 7, 8, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 16383, 16384, 16385)]
    return blocks


@pytest.mark.parametrize("variant", [3, 0, 1, 2], ids=["two-kernel-window-match-finder", "two-kernel-batch-probe", "two-kernel-serial-probe", "one-kernel"])
def test_zstd_compress_is_bit_exact_with_oracle(gb, o, variant):
    blocks = encoder_inputs()
    blocks += [b for b in common.synthetic_blocks(77, 12)]   # RandomGenerator-style data: long literal runs, many same-hash positions per batch
    caps = [o.max_compressed_length("zstd", len(b)) for b in blocks]
    gb.set_option("zstd.compress.variant", variant)
    try:
        outs, status, _ = gb.run(OP_ZSTD_COMPRESS, blocks, caps)
    finally:
        gb.set_option("zstd.compress.variant", 3)
    assert all(s == 0 for s in status), status
    for i, (b, z) in enumerate(zip(blocks, outs)):
        assert z == o.compress("zstd", b, caps[i]), "block %d (len %d): gpu %d bytes" % (i, len(b), len(z))
    # and the GPU decoder restores the plaintext from the GPU encoder's frames
    plain, status, err = gb.run(OP_ZSTD_DECOMPRESS, outs, [max(len(b), 1) for b in blocks])
    for i, (b, p, s) in enumerate(zip(blocks, plain, status)):
        if len(b) == 0:
            continue
        assert s == 0 and p == b, (i, s, err[i])


def test_zstd_compress_pinned_hashes_and_third_party_decoder(gb, o):
    import hashlib
    import pyarrow as pa
    sample = common.corpus_sample()
    blocks = [d for _, d, _ in sample]
    outs, status, _ = gb.run(OP_ZSTD_COMPRESS, blocks, [o.max_compressed_length("zstd", len(b)) for b in blocks])
    assert all(s == 0 for s in status)
    codec = pa.Codec("zstd")
    for (name, d, e), z in zip(sample, outs):
        assert hashlib.sha256(z).hexdigest() == e["zstd"]["sha256"], name       # committed hash of the oracle's stream
        assert codec.decompress(z, decompressed_size=len(d)).to_pybytes() == d  # libzstd accepts the frame


def test_zstd_compress_output_too_small(gb, o):
    b = common.corpus_sample()[0][1]
    cap = o.max_compressed_length("zstd", len(b))
    outs, status, _ = gb.run(OP_ZSTD_COMPRESS, [b, b, b], [10, cap, 20000])
    assert status[1] == 0 and outs[1] == o.compress("zstd", b, cap)
    for i, c in ((0, 10), (2, 20000)):
        try:
            expect = o.compress("zstd", b, c)
            assert status[i] == 0 and outs[i] == expect
        except OracleError as e:
            assert status[i] == e.status


def test_zstd_host_api_round_trip(o):
    import aircompressor_amd as A
    data = common.corpus_sample()[2][1]
    comp, decomp = A.ZstdHipCompressor(), A.ZstdHipDecompressor()
    cap = comp.max_compressed_length(len(data))
    assert cap == o.max_compressed_length("zstd", len(data))
    out = bytearray(cap)
    n = comp.compress(data, 0, len(data), out, 0, cap)
    assert bytes(out[:n]) == o.compress("zstd", data, cap)
    assert decomp.get_decompressed_size(out, 0, n) == len(data)
    back = bytearray(len(data))
    assert decomp.decompress(out, 0, n, back, 0, len(data)) == len(data) and bytes(back) == data
    with pytest.raises(A.IllegalArgumentException):
        comp.compress(data, 0, len(data), bytearray(5), 0, 5)
