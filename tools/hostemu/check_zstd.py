"""Runs the Zstd decode pipeline (zstd_decompress_pipe.hip) on the CPU (tools/hostemu/libemu_zstd.so) over frames of the oracle's encoder
(the Java compressor restated) and of libzstd, and compares with the plaintext.  Items the pipeline hands to its fallback list (the
one-kernel decoder, which the emulator does not run) are counted; `--expect-fast` makes any of them a failure."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib, native_libs

try:
    import pyarrow as pa  # (bundles its own libzstd, whose level 3 splits blocks where the statistics change: frames of many short blocks)

    def libzstd(p, level):
        return pa.Codec("zstd", compression_level=level).compress(p, asbytes=True)
    HAVE_LIBZSTD = True
except ImportError:
    libzstd = native_libs.zstd_compress
    HAVE_LIBZSTD = native_libs.available()

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_zstd.so"))
o = oracle_lib.load()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def run(frames, caps, tile=0, exec_mode=1, pass_blocks=0, mb_max_bytes=0):
    n = len(frames)
    src_off = np.zeros(n, dtype=np.int64); src_len = np.zeros(n, dtype=np.int32)
    dst_off = np.zeros(n, dtype=np.int64); dst_cap = np.array(caps, dtype=np.int32)
    pos = 64
    for i, f in enumerate(frames):
        src_off[i] = pos; src_len[i] = len(f); pos += len(f) + 3  # (unaligned on purpose)
    src = np.full(pos + 64, 0x5A, dtype=np.uint8)
    for i, f in enumerate(frames):
        src[src_off[i]:src_off[i] + len(f)] = np.frombuffer(f, dtype=np.uint8)
    pos = 64
    for i, c in enumerate(caps):
        dst_off[i] = pos; pos += c + 64 + (i % 5)
    dst = np.full(pos + 64, 0xA5, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.int32); status = np.zeros(n, dtype=np.int32); err = np.zeros(n, dtype=np.int64)
    fb = np.zeros(n + 1, dtype=np.int32)
    counters = np.zeros(64, dtype=np.int32)
    k = emu.emu_zstd_pipe(P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(dst_cap), P(out_len), P(status), P(err), n, tile, exec_mode, P(fb), pass_blocks, P(counters), ctypes.c_int64(mb_max_bytes))
    run.counters = counters
    outs = []
    for i in range(n):
        outs.append(dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() if status[i] == 0 else None)
        lo = dst_off[i] + caps[i]
        hi = dst_off[i + 1] if i + 1 < n else len(dst)
        # (a few bytes beyond the capacity may be scribbled by whole-vector stores only where the product allows it: nowhere)
        assert (dst[lo:hi] == 0xA5).all(), "item %d: wrote beyond its capacity" % i
    return outs, status, sorted(int(x) for x in fb[:k])


def frame_blocks(f):
    fhd = f[4]
    pos = 5 + (0 if fhd & 0x20 else 1) + ((1 if fhd & 0x20 else 0) if fhd >> 6 == 0 else 1 << (fhd >> 6))
    n = 0
    while True:
        h = int.from_bytes(f[pos:pos + 3], "little")
        pos += 3 + (1 if (h >> 1) & 3 == 1 else h >> 3)
        n += 1
        if h & 1:
            return n


def multi_block(expect_fast):
    """frames of several blocks (and of raw / RLE blocks) through the multi-block stages, at several pass sizes"""
    plains = common.multi_block_plains()  # (tests/common.py: also the GPU test's inputs)
    long_frame = len(plains) - 1
    bad = 0
    encs = [("oracle", lambda p: o.compress("zstd", p))]
    if HAVE_LIBZSTD:
        encs += [("libzstd-%d" % l, (lambda l: lambda p: libzstd(p, l))(l)) for l in ((1, 3, 9, 19) if "--quick" not in sys.argv else (3, 19))]
    for name, enc in encs:
        frames = [bytes(enc(p)) for p in plains]
        for pass_blocks, pad in (((2048, 0), (16, 11), (64, 0), (8192, 3)) if "--quick" not in sys.argv else ((2048, 0), (16, 11)) if name == "oracle" else (((8192, 3), (16, 11)) if name == "libzstd-3" else ((2048, 0),))):
            # (the provider refuses more than 400 MB: more than this batch needs -- the stages ask for what the batch needs, not for a full pass)
            outs, status, fb = run(frames, [len(p) + pad for p in plains], pass_blocks=pass_blocks, mb_max_bytes=400 << 20 if pass_blocks == 8192 else 0)
            c = run.counters
            for i, p in enumerate(plains):
                if i in fb:
                    continue
                if status[i] != 0 or outs[i] != p:
                    bad += 1
                    first = -1
                    if outs[i] is not None:
                        m = min(len(outs[i]), len(p))
                        d = np.nonzero(np.frombuffer(outs[i][:m], dtype=np.uint8) != np.frombuffer(p[:m], dtype=np.uint8))[0]
                        first = int(d[0]) if len(d) else m
                    print("MISMATCH %s pass %d item %d (len %d): status %d, produced %s, first difference at %d" % (name, pass_blocks, i, len(p), status[i], None if outs[i] is None else len(outs[i]), first))
            too_long = [i for i, f in enumerate(frames) if frame_blocks(f) > 8 * pass_blocks]
            print("%s, passes of %d: %d frames, listed %d (%d blocks), fast %d, fallback list %s (more blocks than a pass has slots: %s), by stage %s" % (
                name, pass_blocks, len(frames), c[40], c[41], c[42], fb, too_long, [int(x) for x in c[33:39]]))
            if fb != too_long:
                bad += 1
                print("MISMATCH %s pass %d: fallback list %s, expected %s" % (name, pass_blocks, fb, too_long))
    # the multi-block execute stage with the 8 KiB window (option zstd.decompress.exec_window): the oracle's frames again
    frames = [bytes(encs[0][1](p)) for p in plains]
    # (exec_mode bits 4 .. 6: the sequence stage in full workgroups of 4 wavefronts x 16 block slots -- option zstd.decompress.seq_waves)
    outs, status, fb = run(frames, [len(p) for p in plains], exec_mode=1 | (4 << 4) | (3 << 2), pass_blocks=2048)  # (... and the literal stage's split tables)
    wide = sum(1 for i, p in enumerate(plains) if i in fb or status[i] != 0 or outs[i] != p)
    print("oracle, 8 KiB executor window: %d frames, %d mismatches" % (len(frames), wide))
    bad += wide
    # a provider that gives nothing: every listed frame must come back on the fallback list, none touched
    frames = [bytes(encs[0][1](p)) for p in plains[:6]]
    outs, status, fb = run(frames, [len(p) for p in plains[:6]], pass_blocks=2048, mb_max_bytes=1)
    if fb != list(range(6)):
        bad += 1
        print("MISMATCH no scratch: fallback list %s" % fb)
    print("zstd multi-block stages: %d mismatches" % bad)
    return bad


def mutations(expect_fast):
    """damaged multi-block frames: whatever the stages do NOT hand to the fallback list must be exactly what the oracle's decoder returns"""
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(23)
    plains = [whole[:300000], whole[200000:200000 + 140000], b"ab" * 90000 + whole[:50000]]
    bad = 0; kept = 0; total = 0
    encs = [("oracle", lambda p: o.compress("zstd", p))]
    if HAVE_LIBZSTD:
        encs.append(("libzstd-3", lambda p: libzstd(p, 3)))
    for name, enc in encs:
        cases = []; caps = []
        for p in plains:
            f = bytearray(enc(p))
            for k in range(24 if "--quick" not in sys.argv else 8):
                g = bytearray(f)
                kind = k % 6
                if kind == 0:    # a byte somewhere
                    g[int(rng.integers(0, len(g)))] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:  # a byte in the first block's headers
                    g[int(rng.integers(4, min(40, len(g))))] = int(rng.integers(0, 256))
                elif kind == 2:  # truncated
                    g = g[:int(rng.integers(8, len(g)))]
                elif kind == 3:  # bytes appended
                    g += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tolist())
                elif kind == 4:  # a run of zeros
                    at = int(rng.integers(0, max(1, len(g) - 16))); g[at:at + 16] = bytes(min(16, len(g) - at))
                else:            # the last bytes (block tail / checksum)
                    g[len(g) - 1 - int(rng.integers(0, 6))] ^= 0x10
                cases.append(bytes(g)); caps.append(len(p) if k % 3 else len(p) - int(rng.integers(1, 5000)))
        outs, status, fb = run(cases, caps, pass_blocks=2048)
        for i, (c, cap) in enumerate(zip(cases, caps)):
            total += 1
            try:
                want = o.decompress("zstd", c, cap)
            except oracle_lib.OracleError:
                want = None
            if i in fb:
                continue
            kept += 1
            if want is None or status[i] != 0 or outs[i] != want:
                bad += 1
                print("MISMATCH %s damaged case %d: oracle %s, stages status %d len %s" % (name, i, "refuses" if want is None else len(want), status[i], None if outs[i] is None else len(outs[i])))
    print("zstd multi-block stages, damaged frames: %d cases, %d decoded by the stages, %d mismatches" % (total, kept, bad))
    return bad


def fuzz(n_cases, seed):
    """--fuzz N [SEED]: N damaged multi-block frames (several mutations each, all encoders) through the stages; whatever is not handed to the
    fallback list must be exactly what the oracle's decoder returns.  For the record: profiles/r02_zstd_mb_emulator_fuzz.txt"""
    rng = np.random.default_rng(seed)
    plains = [p for p in common.multi_block_plains() if 131072 < len(p) <= 800000]
    encs = [lambda p: o.compress("zstd", p)] + ([lambda p: libzstd(p, 1), lambda p: libzstd(p, 3), lambda p: libzstd(p, 19)] if HAVE_LIBZSTD else [])
    frames = [(bytes(e(p)), len(p)) for p in plains for e in encs]
    bad = kept = refused_ok = 0
    done = 0
    while done < n_cases:
        cases = []; caps = []
        for _ in range(min(48, n_cases - done)):
            f, n = frames[int(rng.integers(0, len(frames)))]
            g = bytearray(f)
            for _ in range(int(rng.integers(1, 4))):
                kind = int(rng.integers(0, 7))
                if kind == 0:
                    g[int(rng.integers(0, len(g)))] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:
                    g[int(rng.integers(4, min(64, len(g))))] = int(rng.integers(0, 256))
                elif kind == 2 and len(g) > 16:
                    g = g[:int(rng.integers(8, len(g)))]
                elif kind == 3:
                    g += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tolist())
                elif kind == 4:
                    at = int(rng.integers(0, max(1, len(g) - 16))); g[at:at + 8] = bytes(rng.integers(0, 256, min(8, len(g) - at), dtype=np.uint8).tolist())
                elif kind == 5:
                    g[len(g) - 1 - int(rng.integers(0, min(8, len(g))))] ^= 0x10
                else:  # a block header: find one by walking, then flip a bit of it
                    at = int(rng.integers(0, len(g)))
                    g[at] ^= int(rng.integers(1, 256))
            cases.append(bytes(g)); caps.append(n if rng.random() < 0.7 else max(1, n - int(rng.integers(1, 70000))))
        outs, status, fb = run(cases, caps, pass_blocks=2048)
        for i, (c, cap) in enumerate(zip(cases, caps)):
            try:
                want = o.decompress("zstd", c, cap)
            except oracle_lib.OracleError:
                want = None
            if i in fb:
                refused_ok += 1
                continue
            kept += 1
            if want is None or status[i] != 0 or outs[i] != want:
                bad += 1
                print("MISMATCH case %d of batch at %d: oracle %s, stages status %d len %s" % (i, done, "refuses" if want is None else len(want), status[i], None if outs[i] is None else len(outs[i])))
        done += len(cases)
    print("zstd multi-block stages, fuzz seed %d: %d damaged frames, %d decoded by the stages (all equal to the oracle's output: %s), %d handed to the fallback list, %d mismatches" % (
        seed, done, kept, bad == 0, refused_ok, bad))
    return bad


def main():
    if "--fuzz" in sys.argv:
        k = sys.argv.index("--fuzz")
        sys.exit(1 if fuzz(int(sys.argv[k + 1]), int(sys.argv[k + 2]) if len(sys.argv) > k + 2 else 1) else 0)
    expect_fast = "--expect-fast" in sys.argv
    part = sys.argv[sys.argv.index("--part") + 1] if "--part" in sys.argv else "all"  # single | multi | damaged | all
    if part == "multi":
        sys.exit(1 if multi_block(expect_fast) else 0)
    if part == "damaged":
        sys.exit(1 if mutations(expect_fast) else 0)
    plains = [d for _, d in common.HAND_CASES if len(d) > 0] + [d[:131072] for _, d, _ in common.corpus_sample()[:8]] + common.synthetic_blocks(5, 6)
    plains = [p for p in plains if len(p) <= 131072]
    bad = 0; slow = 0; total = 0
    quick = "--quick" in sys.argv
    for name, enc in (("oracle", lambda p: o.compress("zstd", p)), ("libzstd-1", lambda p: libzstd(p, 1)), ("libzstd-3", lambda p: libzstd(p, 3)),
                      ("libzstd-9", lambda p: libzstd(p, 9))):
        if name.startswith("libzstd") and (not HAVE_LIBZSTD or (quick and name != "libzstd-3")):
            continue
        frames = [bytes(enc(p)) for p in plains]
        # (second run: the 8-items-per-wavefront instantiations of the literal and sequence stages -- mode bits 2 and 3)
        # (... and the sequence stage in full workgroups of 1, 2 and 4 wavefronts: mode bits 4 .. 6, lanes without an item beside lanes with one)
        # (... and the literal stage at 8 and 10 items per wavefront: mode bits 2, 3 -- option zstd.decompress.lit_items)
        for pad, mode in ((0, 1), (37, 1), (0, 1 | (1 << 4)), (0, 1 | (2 << 4)), (0, 1 | (4 << 4)), (0, 1 | (1 << 2)), (0, 1 | (2 << 2)), (0, 1 | (3 << 2)), (0, 1 | (1 << 7))):  # (3 << 2: 13 items, symbols and length nibbles apart; 1 << 7: 16 items, lengths by symbol)
            outs, status, fb = run(frames, [len(p) + pad for p in plains], exec_mode=mode)
            for i, p in enumerate(plains):
                total += 1
                if i in fb:
                    slow += 1
                elif status[i] != 0 or outs[i] != p:
                    bad += 1
                    print("MISMATCH %s item %d (len %d): status %d" % (name, i, len(p), status[i]))
        print("%s: %d frames, fallback list %s" % (name, len(frames), fb))
    print("zstd pipeline: %d cases, %d mismatches, %d on the fallback list" % (total, bad, slow))
    if part == "all":
        bad += multi_block(expect_fast)
        bad += mutations(expect_fast)
    if bad or (expect_fast and slow):
        sys.exit(1)


def stream_cases():
    """launch_zstd_stream_step under the emulator: frames cut into steps of a few blocks, tables / repeat offsets / window / checksum carried from
    step to step; this script plays the host's part (achip_abi.cpp: achip_zstdstream_decompress_feed) -- the walk over the block headers, the
    stand-in frame header in front of a step, the history kept behind the output.  Damaged frames: the blocks in front of the damage are
    delivered, the first damaged block is where the oracle's decoder fails too."""
    import struct
    emu.emu_zstd_stream_carry_bytes.restype = ctypes.c_int64
    cb = emu.emu_zstd_stream_carry_bytes()

    def decode(frame, step_blocks, window_cap=None):
        """returns (plaintext delivered, index of the first bad block or None)"""
        assert frame[:4] == b"\x28\xb5\x2f\xfd"
        fhd = frame[4]
        single = (fhd & 0x20) != 0
        cs = fhd >> 6
        pos = 5
        window = None
        if not single:
            wd = frame[pos]; pos += 1
            base = 1 << (10 + (wd >> 3)); window = base + (base // 8) * (wd & 7)
        n = (1 if single else 0) if cs == 0 else (1 << cs)
        content = int.from_bytes(frame[pos:pos + n], "little") + (256 if cs == 1 else 0) if n else None
        pos += n
        look = window if content is None else (content if window is None else min(window, content))
        if window_cap:
            look = min(look, window_cap)
        has_checksum = (fhd & 4) != 0
        carry = np.zeros(cb, dtype=np.uint8)
        emu.emu_zstd_stream_carry_init(P(carry))
        W = max(look, 1 << 16)
        S = step_blocks * 131072
        hist = np.full(W + S + 64, 0xEE, dtype=np.uint8)
        hist_len = 0
        out = bytearray()
        block_no = 0
        while True:
            blocks = []
            closing = False
            at = pos
            while len(blocks) < step_blocks:
                hd = int.from_bytes(frame[at:at + 3], "little")
                typ, size = (hd >> 1) & 3, hd >> 3
                st = 1 if typ == 1 else size
                blocks.append(frame[at:at + 3 + st])
                at += 3 + st
                if hd & 1:
                    closing = True
                    break
            body = bytearray(b"\x28\xb5\x2f\xfd\x20\x00")
            for i, b in enumerate(blocks):
                b = bytearray(b)
                b[0] = (b[0] & 0xFE) | (1 if i == len(blocks) - 1 else 0)
                body += b
            src = np.frombuffer(bytes(body), dtype=np.uint8).copy()
            expected = int.from_bytes(frame[at:at + 4], "little") if closing and has_checksum else 0
            result = np.zeros(3, dtype=np.int32)
            base = hist[W - hist_len:]
            rc = emu.emu_zstd_stream_step(P(carry), P(src), len(src), len(blocks), P(base), hist_len, hist_len + S, 1 if closing else 0, 1 if has_checksum else 0,
                                          ctypes.c_uint32(expected), P(result))
            assert rc == 0, rc
            good, produced, verdict = int(result[0]), int(result[1]), int(result[2])
            out += hist[W:W + produced].tobytes()
            keep = min(look, hist_len + produced)
            hist[W - keep:W] = hist[W + produced - keep:W + produced].copy()
            hist_len = keep
            if good < len(blocks):
                return bytes(out), block_no + good
            block_no += len(blocks)
            if closing:
                if has_checksum and verdict != 1:
                    return bytes(out), "checksum"
                return bytes(out), None
            pos = at

    bad = 0
    plains = common.multi_block_plains()
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    plains = [p for p in plains if len(p) > 131072][:6] + [whole[:700000], (whole * 3)[:1500000]]
    encs = [("oracle", lambda p: o.compress("zstd", p)), ("oracle-stream", lambda p: o.zstd_stream_compress(p))]
    if HAVE_LIBZSTD:
        encs += [("libzstd-3", lambda p: libzstd(p, 3)), ("libzstd-19", lambda p: libzstd(p, 19))]
    total = 0
    for name, enc in encs:
        for k, p in enumerate(plains):
            f = bytes(enc(p))
            for step_blocks in ((1, 3, 32) if "--quick" not in sys.argv else (2,)):
                got, where = decode(f, step_blocks)
                total += 1
                if where is not None or got != p:
                    bad += 1
                    print("MISMATCH stream %s plain %d (%d bytes) steps of %d: stopped at %s, %d bytes" % (name, k, len(p), step_blocks, where, len(got)))
    print("zstd stream steps: %d decodes, %d mismatches" % (total, bad), flush=True)
    # damage: what is delivered is a prefix of the plaintext made of whole blocks, and the oracle's decoder fails as well
    rng = np.random.default_rng(5)
    cases = 0
    for name, enc in encs[:3]:
        for p in plains[:3]:
            f = bytearray(enc(p))
            for _ in range(4 if "--quick" not in sys.argv else 1):
                g = bytearray(f)
                g[int(rng.integers(12, len(g) - 4))] ^= 1 << int(rng.integers(0, 8))
                try:
                    got, where = decode(bytes(g), 4)
                except Exception as e:  # (a damaged block header makes this script's own walk run off the frame)
                    continue
                cases += 1
                try:
                    ref_plain = o.decompress("zstd", bytes(g), len(p) + 1024)
                    ref_ok = True
                except oracle_lib.OracleError:
                    ref_ok = False
                if where is None:
                    if not ref_ok or got != ref_plain:
                        bad += 1
                        print("MISMATCH damaged %s: the steps decoded %d bytes, the oracle %s" % (name, len(got), "fails" if not ref_ok else "differs"))
                else:
                    # (a flipped bit in literal or sequence DATA decodes to other bytes and is caught by the checksum at the frame's end, here as in the
                    # Java stream; a stop at a block must deliver whole blocks of the plaintext in front of it)
                    if ref_ok or (where != "checksum" and got != p[:len(got)]):
                        bad += 1
                        print("MISMATCH damaged %s: stopped at %s with %d bytes (a prefix: %s), the oracle %s" % (name, where, len(got), got == p[:len(got)], "decodes" if ref_ok else "fails"))
    print("zstd stream steps, damaged frames: %d cases, %d mismatches in all" % (cases, bad), flush=True)
    return bad


if __name__ == "__main__":
    if "--stream" in sys.argv:
        sys.exit(1 if stream_cases() else 0)
    main()
