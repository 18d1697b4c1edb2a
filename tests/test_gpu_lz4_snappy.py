"""GPU parity tests (through the C ABI) for the LZ4 and Snappy block codecs against the CPU oracle:
bit-exact plaintext on decode, bit-exact compressed streams on encode, identical status/offset on the
reference's error vectors.  Mirrors T/AbstractTestCompression.java's conformance cases."""
import hashlib
import os

import numpy as np
import pytest

from tests import common, oracle_lib
from tests.oracle_lib import OracleError

pytestmark = pytest.mark.gpu

CODECS = {
    "lz4": dict(c=1, d=0, group_opt="lz4.decompress.group", variant_opt="lz4.decompress.variant"),
    "snappy": dict(c=3, d=2, group_opt="snappy.decompress.group", variant_opt="snappy.decompress.variant"),
}
# ring decoder configurations: (variant, lanes per block, ring class); variant 1 = LDS rings
DECODERS = [(1, 16, 0), (1, 16, 1), (1, 4, 0), (1, 4, 1), (1, 2, 0), (1, 2, 1), (1, 1, 0), (1, 1, 1), (1, 8, 0), (1, 8, 1), (1, 32, 0), (1, 32, 1), (1, 64, 0), (1, 64, 1), (1, 4, 3)]


def configure(gb, codec, cfg):
    variant, group, ring = cfg
    gb.set_option(CODECS[codec]["variant_opt"], variant)
    gb.set_option(CODECS[codec]["group_opt"], group)
    # ring class 3 is the LATENCY class: not an option value but what the context picks for batches of at most decompress.latency_max_blocks blocks
    # (256 by default: a wavefront and 128 KiB of LDS history per block); the other configurations are tested with it switched off
    gb.set_option("decompress.ring_class", 0 if ring == 3 else ring)
    gb.set_option("decompress.latency_max_blocks", 65536 if ring == 3 else 0)


@pytest.fixture(scope="module")
def gb():
    from tests.gpu_harness import GpuBatch
    return GpuBatch(0)


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def all_blocks():
    blocks = [d for _, d in common.HAND_CASES]
    blocks += [d for _, d, _ in common.corpus_sample()]
    blocks += common.synthetic_blocks(5, 36)
    base = common.corpus_sample()[0][1]
    blocks += [base[:n] for n in range(1, 256, 7)]
    return blocks


@pytest.mark.parametrize("codec, variant", [("lz4", 4), ("lz4", 1), ("lz4", 0), ("snappy", 4), ("snappy", 2), ("snappy", 1), ("snappy", 0)])
def test_compress_is_bit_exact_with_oracle(gb, o, codec, variant):
    # THE DEFAULT OF BOTH CODECS IS 4 = many matches per window of 64 positions (lz4_compress_mw.h / snappy_compress_mw.h, Snappy in two tiers).
    # Tested non-default variants: 0 = serial probes, 1 = 64 probes per step, 2 (Snappy) = batch probes in two tiers: hash tables in LDS and in
    # global memory
    gb.set_option("%s.compress.variant" % codec, variant)
    blocks = all_blocks()
    caps = [o.max_compressed_length(codec, len(b)) for b in blocks]
    outs, status, _ = gb.run(CODECS[codec]["c"], blocks, caps)
    assert all(s == 0 for s in status), status
    for i, (b, c) in enumerate(zip(blocks, outs)):
        assert c == o.compress(codec, b), "block %d (len %d)" % (i, len(b))
    # committed hashes of the oracle's streams for the corpus sample (tests/golden/corpus_sample.json)
    n_hand = len(common.HAND_CASES)
    for k, (_, _, e) in enumerate(common.corpus_sample()):
        assert hashlib.sha256(outs[n_hand + k]).hexdigest() == e[codec]["sha256"]
    gb.set_option("%s.compress.variant" % codec, 4)


@pytest.mark.parametrize("mem_waves", [1, 2])
def test_lz4_two_tier_encoder_is_bit_exact_with_oracle(o, mem_waves):
    """lz4.compress.mem_waves: the window encoder in workgroups of five wavefronts with their tables in LDS and one or two with theirs in global memory (batches
    of 5 120 blocks and more; here the threshold is lowered so that a small batch takes it, and a batch beyond the default threshold follows) -- the oracle's
    bytes for every block, whichever wavefront drew it."""
    from tests.gpu_harness import GpuBatch
    g = GpuBatch(0)
    blocks = [b for b in all_blocks() if len(b) <= 65536]
    caps = [o.max_compressed_length("lz4", len(b)) for b in blocks]
    want = [o.compress("lz4", b) for b in blocks]
    try:
        g.set_option("lz4.compress.mem_waves", mem_waves)
        g.set_option("lz4.compress.tier_min_blocks", 1)
        outs, status, _ = g.run(CODECS["lz4"]["c"], blocks, caps)
        assert all(s == 0 for s in status), status
        assert outs == want
        g.set_option("lz4.compress.tier_min_blocks", 5120)
        reps = 5120 // len(blocks) + 1
        outs, status, _ = g.run(CODECS["lz4"]["c"], blocks * reps, caps * reps)
        assert all(s == 0 for s in status)
        assert [hashlib.sha256(c).digest() for c in outs] == [hashlib.sha256(c).digest() for c in want] * reps
    finally:
        g.set_option("lz4.compress.mem_waves", 1)
        g.set_option("lz4.compress.tier_min_blocks", 5120)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
@pytest.mark.parametrize("cfg", DECODERS)
def test_decompress_matches_plaintext_all_decoder_configs(gb, o, codec, cfg):
    configure(gb, codec, cfg)
    blocks = [b for b in all_blocks() if not (codec == "lz4" and len(b) == 0)]
    comp = [o.compress(codec, b) for b in blocks]
    for pad in (0, 100):  # exact capacity and padded capacity (T/AbstractTestCompression.java:110-129)
        caps = [len(b) + pad for b in blocks]
        outs, status, _ = gb.run(CODECS[codec]["d"], comp, caps, unaligned=(pad == 0))
        for i, (b, p, s) in enumerate(zip(blocks, outs, status)):
            assert s == 0, (i, len(b), s)
            assert p == b, "block %d (len %d)" % (i, len(b))
    configure(gb, codec, DECODERS[0])


@pytest.mark.parametrize("parse", [1, 2], ids=["lane-per-block-parse", "wavefront-per-block-parse"])
@pytest.mark.parametrize("cfg", [(7, 4, 0)], ids=lambda c: "variant%d-gs%d-rc%d" % c)
def test_lz4_two_pass_decoder(gb, o, cfg, parse):
    """variant 7 (lz4_decompress_v7.hip: parse to records, a wavefront per block executes them), forced for any batch size, with either parser
    (lz4.decompress.parse: a lane per block -- what large batches take -- or a wavefront per block -- what batches below 32 768 blocks take):
    plaintext, status and error offsets equal the oracle's, corrupt streams included; blocks of several hundred KiB with literal runs and
    matches of that order (thousands of records from one sequence, chunk after chunk of the arena)"""
    rng = np.random.default_rng(7)
    text = b"".join(d for _, d, _ in common.corpus_sample()[:4])
    # (the last three: sequences longer than the wavefront parser's window of 64 stream positions -- 50 / 20 / 130 random bytes, each run twice -- which it reads
    #  one at a time, next to text, which it reads 64 positions a trip: both modes and the changes between them in one block)
    fragments = [np.tile(rng.integers(0, 256, size=(2000, w), dtype=np.uint8), (1, 2)).reshape(-1)[:n].tobytes() for w, n in ((50, 65536), (20, 65536), (130, 300000))]
    blocks = all_blocks() + [bytes(rng.integers(0, 256, 300000, dtype=np.uint8)), bytes(1 << 20), text[:200000] + bytes(70000) + text[:50000],
                             (bytes(rng.integers(0, 256, 700, dtype=np.uint8)) * 300)[:200001]] + fragments + [fragments[0][:30000] + text[:40000] + fragments[2][:50000] + text[:3000]]
    cases = [(o.compress("lz4", b), len(b)) for b in blocks] + [(o.compress("lz4", b), len(b) + 37) for b in blocks[:20]]
    cases += [(bytes([15, 0, 0, 255, 255, 0x8A, 49, 255, 255, 0]), 1024), (b"", 10), (b"\x00", 0), (b"\x10a", 0), (bytes([0xF0]) + b"\xff" * 4000, 1 << 16)]
    for b in [d for _, d, _ in common.corpus_sample()[:3]]:
        c = bytearray(o.compress("lz4", b))
        cases += [(bytes(c), len(b) - 1), (bytes(c[:len(c) // 2]), len(b)), (bytes(c[:-1]), len(b))]
        for _ in range(8):
            m = bytearray(c)
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cases.append((bytes(m), len(b)))
    configure(gb, "lz4", cfg)
    gb.set_option("lz4.decompress.parse", parse)
    try:
        outs, status, err = gb.run(CODECS["lz4"]["d"], [c for c, _ in cases], [cap for _, cap in cases], unaligned=True)
    finally:
        gb.set_option("lz4.decompress.parse", 0)
        configure(gb, "lz4", DECODERS[0])
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = _oracle_status(o, "lz4", c, cap)
        assert status[i] == est, "case %d: gpu status %d oracle %d" % (i, status[i], est)
        if est != 0:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)
        else:
            assert outs[i] == eout, "case %d" % i


@pytest.mark.parametrize("parse", [1, 2], ids=["lane-per-block-parse", "wavefront-per-block-parse"])
@pytest.mark.parametrize("variant", [7], ids=["two-pass"])
def test_snappy_two_pass_decoder(gb, o, variant, parse):
    """variant 7 (snappy_decompress_v5.hip), forced for any batch size, with either parser (snappy.decompress.parse: a lane per block, or a wavefront
    per block -- what batches of up to 4 096 blocks take): plaintext, status and error offsets equal the oracle's, corrupt streams included; and
    streams of random elements of every kind (copies with 4-byte offsets, runs with length bytes, runs behind runs: what the Java encoder never
    writes), some beyond 64 KiB"""
    rng = np.random.default_rng(11)
    text = b"".join(d for _, d, _ in common.corpus_sample()[:2])
    # (elements longer than the wavefront parser's window -- 50 / 20 / 58 random bytes, each run twice -- which it reads one pair at a time, alone and next to text)
    fragments = [np.tile(rng.integers(0, 256, size=(2000, w), dtype=np.uint8), (1, 2)).reshape(-1)[:n].tobytes() for w, n in ((50, 65536), (20, 65536), (58, 200000))]
    blocks = all_blocks() + fragments + [fragments[0][:30000] + text[:40000] + fragments[2][:50000] + text[:3000]]
    cases = [(o.compress("snappy", b), len(b)) for b in blocks] + [(o.compress("snappy", b), len(b) + 37) for b in blocks[:20]]
    for target in (40, 500, 3000, 20000, 70000, 150000, 400000):
        for _ in range(4):
            c, n = common.snappy_random_stream(rng, target)
            cases += [(c, n), (c, n + 9), (c, n - 1), (c[:len(c) - 1 - int(rng.integers(0, 40))], n)]
    cases += [(b"", 10), (b"\x00", 0), (b"\x05", 16), (bytes([0x80] * 5 + [1]), 16), (bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F]), 16)]
    for b in [d for _, d, _ in common.corpus_sample()[:4]]:
        c = bytearray(o.compress("snappy", b))
        cases += [(bytes(c), len(b) - 1), (bytes(c[:len(c) // 2]), len(b)), (bytes(c[:-1]), len(b))]
        for _ in range(12):
            m = bytearray(c)
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cases.append((bytes(m), len(b)))
    configure(gb, "snappy", (variant, 4, 0))
    gb.set_option("snappy.decompress.parse", parse)
    try:
        outs, status, err = gb.run(CODECS["snappy"]["d"], [c for c, _ in cases], [cap for _, cap in cases], unaligned=True)
    finally:
        gb.set_option("snappy.decompress.parse", 0)
        configure(gb, "snappy", DECODERS[0])
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = _oracle_status(o, "snappy", c, cap)
        assert status[i] == est, "case %d: gpu status %d oracle %d" % (i, status[i], est)
        if est != 0:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)
        else:
            assert outs[i] == eout, "case %d" % i


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_auto_mode_picks_a_decoder_on_the_device(gb, o, codec):
    """variant 5 (the default): batches of at least auto_min_blocks blocks are probed on the device -- groups of 16 consecutive blocks
    whose compressed sizes differ by more than 2x count as mixed, and the heads of sampled blocks say, block by block, whether its
    sequences are short -- and a mostly mixed batch or one with a third of its blocks short goes to the two-pass decoder (3), a batch of
    long copies to the rings (0); all give the oracle's plaintext"""
    text = [d for _, d, _ in common.corpus_sample()][:2]
    flat = [bytes(65536), bytes(range(256)) * 256]
    rng = np.random.default_rng(3)
    frag = [np.tile(rng.integers(0, 256, size=(656, 50), dtype=np.uint8), (1, 2)).reshape(-1)[:65536].tobytes() for _ in range(4)]
    uniform = (text * 40)[:64]
    mixed = [text[i % 2] if i % 2 == 0 else flat[(i // 2) % 2] for i in range(64)]
    longcopies = (frag * 16)[:64]
    # text and runs in separate halves: no mixed groups, and the runs' bytes swamp the pooled bytes per sequence -- but half the blocks are short
    halves = [text[1]] * 32 + (flat * 16)[:32]  # (text[1]: 18 bytes per sequence at its head; text[0] sits at the threshold)
    sprinkled = (text * 4)[:8] + (frag * 16)[:56]  # an eighth of the blocks short: stays with the rings
    gb.set_option("%s.decompress.variant" % codec, 5)
    gb.set_option("lz4.decompress.auto_min_blocks", 32)  # (one threshold for both codecs)
    gb.set_option("decompress.auto_remember", 0)  # (both decoders in every call, the probes of THIS call pick: the batches below have one shape and different data)
    try:
        for blocks, expect_mixed, expect_choice in ((uniform, False, 3), (mixed, True, 3), (longcopies, False, 0), (halves, False, 3), (sprinkled, False, 0), (mixed[:16], None, -1)):
            comp = [o.compress(codec, b) for b in blocks]
            outs, status, _ = gb.run(CODECS[codec]["d"], comp, [len(b) for b in blocks], unaligned=True)
            assert all(s == 0 for s in status) and outs == blocks
            groups = gb.codec.native.get_stat("lz4.decompress.mixed_groups")
            choice = gb.codec.native.get_stat("decompress.choice")
            if expect_mixed is None:
                assert groups == -1 and choice == -1  # below auto_min_blocks: no probe, the rings
            else:
                assert (groups * 4 > len(blocks) // 16) == expect_mixed, groups
                assert choice == expect_choice, choice
                assert gb.codec.native.get_stat("decompress.twopass_fallback_blocks") == 0
    finally:
        gb.set_option("lz4.decompress.auto_min_blocks", 4096)
        gb.set_option("decompress.auto_remember", 1)
        configure(gb, codec, DECODERS[0])


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_auto_mode_remembers_its_choice(o, codec):
    """decompress.auto_remember (round 6): a call's probe statistics come home behind its kernels; while the batches keep their shape (codec, count, buffers) only the decoder
    the last arrived statistics chose is launched -- the probes still run in every call, so a context whose data changes character follows one call later (for a caller
    that waits for its results).  Whatever runs, the bytes are the reference's."""
    import torch
    from tests.gpu_harness import GpuBatch
    import aircompressor_amd as A
    g = GpuBatch(0, options={"lz4.decompress.auto_min_blocks": 32})
    text = [d for _, d, _ in common.corpus_sample()][:2]
    rng = np.random.default_rng(5)
    frag = [np.tile(rng.integers(0, 256, size=(656, 50), dtype=np.uint8), (1, 2)).reshape(-1)[:65536].tobytes() for _ in range(4)]
    n, bs = 64, 65536
    kinds = {"text": (text * 32)[:n], "long": (frag * 16)[:n]}
    comp = {k: [o.compress(codec, b) for b in v] for k, v in kinds.items()}
    cap = (max(len(c) for v in comp.values() for c in v) + 15) & ~15
    dev = g.dev
    d_src = torch.zeros(n * cap, dtype=torch.uint8, device=dev)   # ONE pair of buffers for every call: the shape the memory keys on
    d_dst = torch.zeros(n * bs + 64, dtype=torch.uint8, device=dev)
    a_so = torch.arange(n, dtype=torch.int64, device=dev) * cap
    a_do = torch.arange(n, dtype=torch.int64, device=dev) * bs
    a_dc = torch.full((n,), bs, dtype=torch.int32, device=dev)
    o_len, st, eo = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)
    op = A.OP_LZ4_DECOMPRESS if codec == "lz4" else A.OP_SNAPPY_DECOMPRESS

    def call(kind):
        host = np.zeros(n * cap, dtype=np.uint8)
        for i, c in enumerate(comp[kind]):
            host[i * cap:i * cap + len(c)] = np.frombuffer(c, dtype=np.uint8)
        d_src.copy_(torch.from_numpy(host))
        a_sl = torch.tensor([len(c) for c in comp[kind]], dtype=torch.int32, device=dev)
        d_dst.zero_()
        torch.cuda.synchronize()
        g.codec.launch(op, d_src, a_so, a_sl, d_dst, a_do, a_dc, o_len, st, eo, n)
        g.codec.synchronize()
        assert int(st.abs().sum().item()) == 0
        assert d_dst[:n * bs].cpu().numpy().tobytes() == b"".join(kinds[kind])
        return g.codec.native.get_stat("decompress.choice")

    assert [call("text") for _ in range(3)] == [3, 3, 3]      # probed, then remembered: the two passes
    # the data changes character under the same shape: ONE call on the remembered decoder (its probes go home), then the rings
    assert [call("long") for _ in range(4)] == [3, 0, 0, 0]
    assert [call("text") for _ in range(3)] == [0, 3, 3]
    g.set_option("decompress.auto_remember", 0)                # both decoders launched, this call's probes pick
    assert [call("long"), call("text"), call("long")] == [0, 3, 0]
    g.codec.native.close()


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_round_trip_gpu_only(gb, o, codec):
    blocks = [b for b in all_blocks() if len(b) > 0]
    caps = [o.max_compressed_length(codec, len(b)) for b in blocks]
    comp, status, _ = gb.run(CODECS[codec]["c"], blocks, caps)
    assert all(s == 0 for s in status)
    plain, status, _ = gb.run(CODECS[codec]["d"], comp, [len(b) for b in blocks])
    assert all(s == 0 for s in status)
    assert plain == blocks


def _oracle_status(o, codec, data, cap):
    try:
        out = o.decompress(codec, data, cap)
        return 0, 0, out
    except OracleError as e:
        return e.status, e.offset, None


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
@pytest.mark.parametrize("cfg", [(1, 16, 0), (1, 4, 0), (1, 2, 1), (1, 1, 0), (1, 8, 1), (1, 64, 0), (7, 4, 0)])
def test_malformed_inputs_report_the_reference_errors(gb, o, codec, cfg):
    """Error KATs of the reference plus systematic corruption: status class/detail and offset must equal the oracle's
    (= what the Java decoder throws), and nothing is written outside the block's output."""
    configure(gb, codec, cfg)
    rng = np.random.default_rng(99)
    cases = []
    if codec == "lz4":
        cases.append((bytes([15, 0, 0, 255, 255, 0x8A, 49, 255, 255, 0]), 1024))  # T/lz4/TestLz4.java:53-60
        cases.append((b"", 10))
        cases.append((b"\x00", 0))
        cases.append((b"\x10a", 0))
        cases.append((bytes([0xF0]) + b"\xff" * 4000, 1 << 16))
        cases.append((bytes([0x1F, ord("a"), 1, 0]) + b"\xff" * 4000, 1 << 16))
    else:
        cases.append((bytes([16, 1, 0, 1, 0, 1, 0, 1, 0]), 1024))  # T/snappy/TestSnappyJava.java:52-59
        cases.append((bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F, 0]), 10))
        cases.append((bytes([0xFF] * 5), 10))
        cases.append((bytes([0x80]), 10))
        cases.append((b"", 10))
        cases.append((bytes([10, 0xFC, 0xFF, 0xFF, 0xFF, 0x7F]) + b"abc", 100))
    sample = [d for _, d, _ in common.corpus_sample()[:4]] + common.synthetic_blocks(8, 6)[:6]
    for b in sample:
        c = bytearray(o.compress(codec, b))
        cases.append((bytes(c), len(b) - 1))            # output one byte short
        cases.append((bytes(c[:len(c) // 2]), len(b)))  # truncated input
        cases.append((bytes(c[:-1]), len(b)))
        for _ in range(6):                               # random byte flips
            m = bytearray(c)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cases.append((bytes(m), len(b)))
            cases.append((bytes(m), len(b) + 64))
    data = [c for c, _ in cases]
    caps = [cap for _, cap in cases]
    outs, status, err = gb.run(CODECS[codec]["d"], data, [max(c, 0) for c in caps])
    for i, (c, cap) in enumerate(cases):
        est, eoff, eout = _oracle_status(o, codec, c, cap)
        assert status[i] == est, "case %d: gpu status %d oracle %d" % (i, status[i], est)
        if est != 0:
            assert err[i] == eoff, "case %d: gpu offset %d oracle %d" % (i, err[i], eoff)
        else:
            assert outs[i] == eout, "case %d" % i
    configure(gb, codec, DECODERS[0])


@pytest.mark.parametrize("variant", [7, 1], ids=["two-pass", "rings"])
def test_snappy_prefix_only_item_at_the_end_of_an_allocation(o, variant):
    """Round 2's ASan finding (achip_seqexec.h LaneFeed::init): a Snappy item that is its length prefix only and ends on a 32-byte
    boundary had the 32 bytes BEHIND it read by the two-pass parser.  Here such items end exactly where a hipMalloc'ed source buffer of a
    whole number of 2 MiB pages ends (nothing of the caller's lies behind them), beside ordinary blocks; status, offset and length must be
    the oracle's.  (tools/hostemu/asan_fuzz.py runs the same streams under AddressSanitizer, which is what can SEE the read.)"""
    import ctypes
    import aircompressor_amd as A
    from aircompressor_amd import native
    nat = A.HipNative(0)
    lib = nat.lib
    nat.set_option("snappy.decompress.variant", variant)
    text = common.corpus_sample()[0][1][:3000]
    items = [(o.compress("snappy", text), len(text))]
    for n in (0, 1, 300, 65536):  # prefix-only streams: "truncated" unless the announced length is 0
        c = bytearray()
        v = n
        while v >= 0x80:
            c.append((v & 0x7F) | 0x80)
            v >>= 7
        c.append(v)
        items.append((bytes(c), n))
    total = 2 << 20
    src = np.zeros(total, dtype=np.uint8)
    offs, lens, caps = [], [], []
    # the ordinary block first, the prefix-only items packed against the END of the buffer: the last one ends at `total`, the others
    # on 32-byte boundaries before it
    src[:len(items[0][0])] = np.frombuffer(items[0][0], dtype=np.uint8)
    offs.append(0); lens.append(len(items[0][0])); caps.append(items[0][1])
    end = total
    for c, n in reversed(items[1:]):
        src[end - len(c):end] = np.frombuffer(c, dtype=np.uint8)
        offs.append(end - len(c)); lens.append(len(c)); caps.append(max(n, 1))
        end -= 32
    n = len(offs)
    dst_off = np.cumsum([0] + [(c + 15) // 16 * 16 for c in caps[:-1]]).astype(np.int64)
    dbytes = int(dst_off[-1]) + caps[-1] + 64
    meta_h = [np.asarray(offs, dtype=np.int64), np.asarray(lens, dtype=np.int32), dst_off, np.asarray(caps, dtype=np.int32)]
    d_src = lib.achip_device_alloc(nat.ctx, total)
    d_dst = lib.achip_device_alloc(nat.ctx, dbytes)
    d_meta = [lib.achip_device_alloc(nat.ctx, m.nbytes) for m in meta_h]
    d_res = [lib.achip_device_alloc(nat.ctx, n * 4), lib.achip_device_alloc(nat.ctx, n * 4), lib.achip_device_alloc(nat.ctx, n * 8)]
    try:
        assert d_src and d_dst and all(d_meta) and all(d_res)
        assert lib.achip_memcpy_h2d(nat.ctx, d_src, src.ctypes.data, total) == 0
        for d, m in zip(d_meta, meta_h):
            assert lib.achip_memcpy_h2d(nat.ctx, d, m.ctypes.data, m.nbytes) == 0
        nat.synchronize()
        r = lib.achip_snappy_decompress_batch(nat.ctx, d_src, d_meta[0], d_meta[1], d_dst, d_meta[2], d_meta[3], d_res[0], d_res[1], d_res[2], n)
        assert r == 0, r
        out_len = np.zeros(n, dtype=np.int32); status = np.zeros(n, dtype=np.int32); err = np.zeros(n, dtype=np.int64)
        out = np.zeros(dbytes, dtype=np.uint8)
        for d, h in zip(d_res + [d_dst], (out_len, status, err, out)):
            assert lib.achip_memcpy_d2h(nat.ctx, h.ctypes.data, d, h.nbytes) == 0
        nat.synchronize()
        for i in range(n):
            comp = bytes(src[offs[i]:offs[i] + lens[i]])
            es, eo, eout = _oracle_status(o, "snappy", comp, caps[i])
            assert (int(status[i]), int(err[i]) if es else 0) == (es, eo if es else 0), "item %d" % i
            if es == 0:
                assert bytes(out[dst_off[i]:dst_off[i] + out_len[i]]) == eout
    finally:
        for d in [d_src, d_dst] + d_meta + d_res:
            if d:
                lib.achip_device_free(nat.ctx, d)
        nat.close()


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_compress_rejects_small_output(gb, o, codec):
    b = common.corpus_sample()[0][1]
    cap = o.max_compressed_length(codec, len(b))
    outs, status, _ = gb.run(CODECS[codec]["c"], [b, b], [cap - 1, cap])
    assert status[0] < 0 and ((-status[0]) & 15) == 2
    assert status[1] == 0 and outs[1] == o.compress(codec, b)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_large_inputs_single_block(gb, o, codec):
    # > 64 KiB: LZ4 needs 32-bit table entries with a 64 KiB window; Snappy walks independent sub-blocks
    big = b"".join(d for _, d, _ in common.corpus_sample()[:5]) + b"xyz"
    cap = o.max_compressed_length(codec, len(big))
    outs, status, _ = gb.run(CODECS[codec]["c"], [big, big[:70000]], [cap, cap])
    assert status == [0, 0]
    assert outs[0] == o.compress(codec, big)
    assert outs[1] == o.compress(codec, big[:70000])
    plain, status, _ = gb.run(CODECS[codec]["d"], outs, [len(big), 70000])
    assert status == [0, 0] and plain[0] == big and plain[1] == big[:70000]


def test_options_that_would_return_wrong_data_do_not_exist():
    """Round 2 shipped development aids behind public options (executor / parser variants that skip work, an encoder stage that stops
    early): their results were not valid.  They were deleted in round 4 together with their build switch; the library refuses their
    numbers, the numbers of the decoder / encoder variants removed since, and values outside every variant option's documented set"""
    import aircompressor_amd as A
    from aircompressor_amd.errors import IllegalArgumentException
    nat = A.HipNative(0)
    try:
        bad = [("decompress.exec_variant", v) for v in (121, 122, 123, 124, 125, 201, 302, 304, 308, 0, 1, 3)]
        bad += [("zstd.compress.variant", v) for v in (100, 4, -1)]
        bad += [("lz4.compress.variant", v) for v in (2, 3, 100)] + [("snappy.compress.variant", v) for v in (3, 5, -1, 100)]
        bad += [("lz4.decompress.variant", v) for v in (0, 2, 3, 4, 6, 8)] + [("snappy.decompress.variant", v) for v in (0, 2, 3, 4, 6, 8)]
        bad += [("hadoop.decompress.variant", 4), ("lz4frame.decompress.variant", 3), ("snappyframed.decompress.variant", 4), ("snappyframed.compress.variant", 2),
                ("zstd.decompress.variant", 2), ("zstd.decompress.exec", 3), ("decompress.ring_class", 3), ("zstd.decompress.lit_items", 12),
                ("zstd.decompress.seq_items", 8), ("zstd.decompress.exec_window", 8192), ("no.such.option", 1)]
        for name, value in bad:
            with pytest.raises(IllegalArgumentException):
                nat.set_option(name, value)
        nat.set_option("decompress.exec_variant", 2)  # (the product's value stays settable)
    finally:
        nat.close()


def test_single_block_host_api_mirrors_reference_interface(o):
    """The drop-in classes: same calls and exceptions as Lz4Java*/SnappyJava* (M/Compressor.java, M/Decompressor.java)."""
    import aircompressor_amd as A
    data = common.corpus_sample()[1][1]
    for comp, decomp, codec in ((A.Lz4HipCompressor(), A.Lz4HipDecompressor(), "lz4"),
                                (A.SnappyHipCompressor(), A.SnappyHipDecompressor(), "snappy")):
        assert comp.is_enabled()
        cap = comp.max_compressed_length(len(data))
        assert cap == o.max_compressed_length(codec, len(data))
        out = bytearray(cap + 10)
        n = comp.compress(data, 0, len(data), out, 5, cap)
        assert bytes(out[5:5 + n]) == o.compress(codec, data)
        back = bytearray(len(data) + 3)
        m = decomp.decompress(out, 5, n, back, 3, len(data))
        assert m == len(data) and bytes(back[3:]) == data
        m = decomp.decompress_segment(memoryview(out)[5:5 + n], memoryview(back)[:len(data)])
        assert m == len(data) and bytes(back[:len(data)]) == data
        with pytest.raises(A.IllegalArgumentException):
            comp.compress(data, 0, len(data) + 1, out, 0, cap)       # verifyRange
        with pytest.raises(A.IllegalArgumentException):
            comp.compress(data, 0, len(data), out, 0, cap - 1 if codec == "snappy" else 10)  # undersized output
        with pytest.raises(A.MalformedInputException):
            decomp.decompress(bytes(out[5:5 + n // 2]), 0, n // 2, back, 0, len(data))
    with pytest.raises(A.MalformedInputException) as e:  # T/lz4/TestLz4.java:53-60
        A.Lz4HipDecompressor().decompress(bytes([15, 0, 0, 255, 255, 0x8A, 49, 255, 255, 0]), 0, 10, bytearray(1024), 0, 1024)
    assert "offset outside destination buffer: offset=3" in str(e.value)
    with pytest.raises(A.MalformedInputException) as e:  # T/snappy/TestSnappyJava.java:52-59
        A.SnappyHipDecompressor().decompress(bytes([16, 1, 0, 1, 0, 1, 0, 1, 0]), 0, 9, bytearray(1024), 0, 1024)
    assert "Malformed input: offset=2" in str(e.value)
    assert A.Lz4HipDecompressor().decompress(b"\x10a", 0, 2, bytearray(0), 0, 0) == -1  # M/lz4/Lz4RawDecompressor.java:52-57
    sd = A.SnappyHipDecompressor()
    assert sd.get_uncompressed_length(o.compress("snappy", data), 0) == len(data)


@pytest.mark.parametrize("codec", ["lz4", "snappy"])
def test_full_size_properties(gb, o, codec):
    """Size-independent properties on 1024 x 64 KiB random-fragment blocks: GPU compress -> GPU decompress must be the identity and a
    sampled subset of the streams must be the oracle's bytes.  (The BASELINE-size batches -- 262 144 blocks, 65 536 frames, oracle-written
    streams, every output byte compared -- are tests/test_gpu_baseline_size.py.)"""
    rng = np.random.default_rng(2024)
    n, size = 1024, 65536
    blocks = []
    for i in range(n):
        raw = [10, 25, 50, 75, 100][i % 5]
        frags = rng.integers(0, 256, size=(size // 100 + 1, raw), dtype=np.uint8)
        blocks.append(np.tile(frags, (1, 100 // raw + 1))[:, :100].reshape(-1)[:size].tobytes())
    cap = o.max_compressed_length(codec, size)
    comp, status, _ = gb.run(CODECS[codec]["c"], blocks, [cap] * n)
    assert all(s == 0 for s in status)
    for i in range(0, n, 37):
        assert comp[i] == o.compress(codec, blocks[i]), i
    plain, status, _ = gb.run(CODECS[codec]["d"], comp, [size] * n)
    assert all(s == 0 for s in status)
    assert plain == blocks
