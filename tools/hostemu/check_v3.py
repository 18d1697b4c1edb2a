"""Runs the ring decoders and the two-pass decoders on the CPU (tools/hostemu/libemu.so) over the GPU parity suite's cases and compares
with the oracle: plaintext, status, error offset, and no write outside the block's output (guard bands)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib
from tests.oracle_lib import OracleError

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", os.environ.get("HOSTEMU_LIB", "libemu.so")))  # (HOSTEMU_LIB=libemu_lockstep.so: emu_lockstep.cpp)
o = oracle_lib.load()


def run(op, blocks, caps, misalign=0):
    n = len(blocks)
    GUARD = 64
    src_off, dst_off = [], []
    pos = misalign
    for b in blocks:
        src_off.append(pos)
        pos += len(b) + 7 + (len(b) % 5)
    src = np.zeros(pos + 64, dtype=np.uint8)
    for b, so in zip(blocks, src_off):
        src[so:so + len(b)] = np.frombuffer(b, dtype=np.uint8)
    pos = GUARD + misalign
    for c in caps:
        dst_off.append(pos)
        pos += c + GUARD + (c % 3)
    dst = np.full(pos + 64, 0xA5, dtype=np.uint8)
    so = np.array(src_off, dtype=np.int64); sl = np.array([len(b) for b in blocks], dtype=np.int32)
    do = np.array(dst_off, dtype=np.int64); dc = np.array(caps, dtype=np.int32)
    ol = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32); eo = np.zeros(n, dtype=np.int64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    r = emu.emu_batch(op, P(src), P(so), P(sl), P(dst), P(do), P(dc), P(ol), P(st), P(eo), n)
    assert r == 0
    outs = []
    for i in range(n):
        outs.append(dst[dst_off[i]:dst_off[i] + ol[i]].tobytes())
        lo = dst_off[i] + caps[i]
        assert (dst[lo:lo + GUARD] == 0xA5).all(), "block %d wrote past its output" % i
        assert (dst[dst_off[i] - GUARD:dst_off[i]] == 0xA5).all() or i > 0, "block %d wrote before its output" % i
    return outs, st.tolist(), eo.tolist()


def expect(codec, data, cap):
    try:
        return 0, 0, o.decompress(codec, data, cap)
    except OracleError as e:
        return e.status, e.offset, None


def cases_for(codec):
    rng = np.random.default_rng(99)
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()] + common.synthetic_blocks(5, 36)
    base = common.corpus_sample()[0][1]
    blocks += [base[:n] for n in range(1, 256, 7)]
    cases = [(o.compress(codec, b), len(b)) for b in blocks] + [(o.compress(codec, b), len(b) + 37) for b in blocks[:40]]
    if codec == "lz4":
        cases += [(bytes([15, 0, 0, 255, 255, 0x8A, 49, 255, 255, 0]), 1024), (b"", 10), (b"\x00", 0), (b"\x10a", 0),
                  (bytes([0xF0]) + b"\xff" * 4000, 1 << 16), (bytes([0x1F, ord("a"), 1, 0]) + b"\xff" * 4000, 1 << 16)]
    else:
        cases += [(bytes([16, 1, 0, 1, 0, 1, 0, 1, 0]), 1024), (bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F, 0]), 10), (bytes([0xFF] * 5), 10), (bytes([0x80]), 10), (b"", 10),
                  (bytes([10, 0xFC, 0xFF, 0xFF, 0xFF, 0x7F]) + b"abc", 100)]
    sample = [d for _, d, _ in common.corpus_sample()[:4]] + common.synthetic_blocks(8, 6)[:6]
    for b in sample:
        c = bytearray(o.compress(codec, b))
        cases += [(bytes(c), len(b) - 1), (bytes(c[:len(c) // 2]), len(b)), (bytes(c[:-1]), len(b))]
        for _ in range(12):
            m = bytearray(c)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            cases += [(bytes(m), len(b)), (bytes(m), len(b) + 64)]
    return cases


def main():
    # (44 / 46 / 48, 54 / 56 / 58: the ring decoders with 4 / 16 / 64 lanes per block -- the product's default is 4 --: the lanes of a group meet
    #  at the emulator-only lockstep points of achip_rings.h)
    only = [int(x) for x in sys.argv[sys.argv.index("--ops") + 1].split(",")] if "--ops" in sys.argv else None
    for codec, ops in (("lz4", (16, 17, 24, 25, 26, 27, 44, 46, 48, 49)), ("snappy", (12, 13, 34, 35, 36, 37, 54, 56, 58, 59))):
        if only is not None:
            ops = tuple(op for op in ops if op in only)
        elif "--quick" in sys.argv:
            ops = tuple(op for op in ops if op not in (46, 48, 56, 58))  # (49 / 59: the latency class stays in)  # (the product's 4 lanes per block only)
        cases = cases_for(codec)
        for op in ops:
            if emu.emu_batch(op, None, None, None, None, None, None, None, None, None, 0) != 0:
                print(codec, "op", op, "not built: skipped")
                continue
            for mis in ((3,) if "--quick" in sys.argv else (0, 3, 13)):
                outs, status, err = run(op, [c for c, _ in cases], [max(cap, 0) for _, cap in cases], mis)
                bad = 0
                for i, (c, cap) in enumerate(cases):
                    est, eoff, eout = expect(codec, c, cap)
                    ok = status[i] == est and (err[i] == eoff if est != 0 else outs[i] == eout)
                    if not ok:
                        bad += 1
                        if bad <= 5:
                            print("  MISMATCH case %d (len %d cap %d): emu status %d off %d len %d | oracle status %d off %d len %s" % (
                                i, len(c), cap, status[i], err[i], len(outs[i]), est, eoff, len(eout) if eout is not None else None))
                print("%s op %d misalign %d: %d cases, %d mismatches" % (codec, op, mis, len(cases), bad))


main()
