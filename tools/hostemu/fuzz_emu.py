"""Differential fuzz of the two-pass decoders ON THE CPU (tools/hostemu/libemu.so: the kernel sources under the fiber emulator) against
the oracle -- status, error offset, plaintext, guard bands -- over random mutations, truncations and capacity changes of real and
synthetic streams.  usage: python tools/hostemu/fuzz_emu.py [cases] [seed]   (the GPU twin of this is tools/fuzz_decoders.py)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ARGS = sys.argv[1:]
src = open(os.path.join(ROOT, "tools", "hostemu", "check_v3.py")).read().replace("\nmain()\n", "\n")
exec(compile(src, os.path.join(ROOT, "tools", "hostemu", "check_v3.py"), "exec"))   # run(), expect(), o, emu, common


def fuzz(n_cases, seed):
    rng = np.random.default_rng(seed)
    only = [int(x) for x in ARGS[ARGS.index("--ops") + 1].split(",")] if "--ops" in ARGS else None
    blocks = [d for _, d, _ in common.corpus_sample()[:8]] + common.synthetic_blocks(9, 12) + [d for _, d in common.HAND_CASES if len(d) > 0]
    blocks += [b[:n] for b in blocks[:4] for n in (17, 300, 5000)]
    total_bad = 0
    for codec, ops in (("lz4", (24, 25, 26)), ("snappy", (34, 35, 36))):
        if only is not None:
            ops = tuple(op for op in ops if op in only)
        if not ops:
            continue
        comp = [o.compress(codec, b) for b in blocks]
        caps = [len(b) for b in blocks]
        if codec == "snappy":
            for target in (40, 500, 3000, 20000, 70000, 150000):
                for _ in range(3):
                    c, n = common.snappy_random_stream(rng, target)
                    assert len(o.decompress("snappy", c, n)) == n
                    comp.append(c)
                    caps.append(n)
        cases = []
        for _ in range(n_cases):
            i = int(rng.integers(0, len(comp)))
            c = bytearray(comp[i])
            cap = caps[i]
            kind = int(rng.integers(0, 8))
            if kind >= 6:
                pass  # as it is
            elif kind <= 2 and len(c) > 0:
                for _ in range(int(rng.integers(1, 5))):
                    c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
            elif kind == 3 and len(c) > 1:
                c = c[:int(rng.integers(0, len(c)))]
            elif kind == 4:
                c += bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            else:
                cap = max(0, cap + int(rng.integers(-8, 9)))
            cases.append((bytes(c), cap))
        want = [expect(codec, c, cap) for c, cap in cases]
        for op in ops:
            outs, status, err = run(op, [c for c, _ in cases], [cap for _, cap in cases], 3)
            wrong = 0
            for i, (est, eoff, eout) in enumerate(want):
                ok = status[i] == est and (err[i] == eoff if est != 0 else outs[i] == eout)
                if not ok:
                    wrong += 1
                    if wrong <= 5:
                        print("MISMATCH", codec, "op", op, "case", i, "emu", status[i], err[i], "oracle", est, eoff, flush=True)
            total_bad += wrong
            print("%s op %d: %d cases (%d malformed), %d mismatches" % (codec, op, len(cases), sum(1 for w in want if w[0] != 0), wrong), flush=True)
    print("TOTAL MISMATCHES", total_bad)
    return total_bad


if __name__ == "__main__":
    sys.exit(1 if fuzz(int(ARGS[0]) if len(ARGS) > 0 else 2000, int(ARGS[1]) if len(ARGS) > 1 else 1) else 0)
