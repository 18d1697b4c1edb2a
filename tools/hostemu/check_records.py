"""Runs the executor for records of any length (achip_seqexec2.h exec_records -- the Zstd pipeline's execute stage) on the CPU
(tools/hostemu/libemu.so): LZ4 blocks of the oracle are parsed into {literal length, match length, offset} records + one literal
buffer, the way the Zstd sequence stage hands them over, executed, and compared with the plaintext; records that run outside their
buffers must be refused."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu.so"))
o = oracle_lib.load()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def sequences(c):
    """LZ4 block -> [(lit, ml, off)], literal bytes, tail literals"""
    ip = 0; seqs = []; lits = bytearray()
    while ip < len(c):
        t = c[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                v = c[ip]; ip += 1; lit += v
                if v != 255: break
        lits += c[ip:ip + lit]; ip += lit
        if ip >= len(c):
            return seqs, bytes(lits), lit
        off = c[ip] | (c[ip + 1] << 8); ip += 2
        ml = t & 15
        if ml == 15:
            while True:
                v = c[ip]; ip += 1; ml += v
                if v != 255: break
        seqs.append((lit, ml + 4, off))
    return seqs, bytes(lits), 0


def run(seqs, lits, cap):
    rec = np.array([l | (m << 18) | (f << 36) for l, m, f in seqs] or [0], dtype=np.uint64)
    lit = np.frombuffer(lits + b"\0" * 0, dtype=np.uint8).copy() if lits else np.zeros(1, dtype=np.uint8)
    guard = 64
    out = np.full(cap + 2 * guard, 0xA5, dtype=np.uint8)
    res = np.zeros(2, dtype=np.int32)
    emu.emu_exec_records(P(rec), len(seqs), P(lit), len(lits), ctypes.c_void_p(out.ctypes.data + guard), cap, P(res))
    assert (out[:guard] == 0xA5).all() and (out[guard + cap:] == 0xA5).all(), "wrote outside the output"
    return out[guard:guard + max(res[0], 0)].tobytes(), int(res[0]), bool(res[1])


def main():
    blocks = [d for _, d in common.HAND_CASES] + [d for _, d, _ in common.corpus_sample()[:10]] + common.synthetic_blocks(4, 24)
    base = common.corpus_sample()[0][1]
    blocks += [base[:n] for n in (1, 5, 13, 15, 16, 17, 31, 33, 100, 1000)] + [b"\0" * 70000, b"ab" * 40000, bytes(range(256)) * 200, b"x" * 20 + b"abcdefghijklmnopqrstuvwxyz" * 3000]
    bad = 0
    for i, b in enumerate(blocks):
        if len(b) == 0:
            continue
        seqs, lits, tail = sequences(o.compress("lz4", b))
        got, n, refused = run(seqs, lits, len(b))
        if refused or got != b:
            bad += 1
            print("MISMATCH block %d (len %d): produced %d refused %s" % (i, len(b), n, refused))
        # the same with room to spare
        got, n, refused = run(seqs, lits, len(b) + 37)
        bad += 1 if (refused or got != b) else 0
        # refusals: capacity one short, a literal buffer too short for the sequences, an offset before the start, an empty offset
        if seqs:
            _, _, r1 = run(seqs, lits, len(b) - 1)
            _, _, r2 = run(seqs, lits[:len(lits) - tail - 1], len(b)) if len(lits) - tail - 1 >= 0 and any(l for l, _, _ in seqs) else (None, None, True)  # the sequences want more literals than there are
            k = len(seqs) // 2
            l, m, f = seqs[k]
            _, _, r3 = run(seqs[:k] + [(l, m, len(b) + 5)] + seqs[k + 1:], lits, len(b))
            _, _, r4 = run(seqs[:k] + [(l, m, 0)] + seqs[k + 1:], lits, len(b))
            if not (r1 and r2 and r3 and r4):
                bad += 1
                print("NOT REFUSED block %d: %s" % (i, (r1, r2, r3, r4)))
    print("exec_records: %d blocks, %d mismatches" % (len(blocks), bad))


main()
