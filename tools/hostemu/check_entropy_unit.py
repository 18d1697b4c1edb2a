"""The Zstd encoder's wave-parallel entropy helpers (aircompressor_amd/csrc/zstd_compress_body.h), one function at a time under the fiber
emulator (libemu_entropy_unit.so), against the oracle's restatement of the Java methods (libunit_oracle_enc.so) on randomised count sets:
  huf    HuffmanCompressionTable.initialize (buildTree + setMaxHeight + the values): code lengths, values, table height
  norm   FiniteStateEntropy.normalizeCounts, and normalizeCounts2 alone (forced) with its three endings
  write  FiniteStateEntropy.writeNormalizedCounts: bytes and size, tight capacities included
    python tools/hostemu/check_entropy_unit.py [cases]          (build lines: the two sources' headers)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_entropy_unit.so"))
ref = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libunit_oracle_enc.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(11)
bad = 0


def huf_case(counts, max_bits):
    global bad
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    ms = len(counts) - 1
    out = []
    for lib, name in ((emu, "unit_huf"), (ref, "unit_ref_huf")):
        bits = np.zeros(256, dtype=np.uint8); vals = np.zeros(256, dtype=np.int16); mb = ctypes.c_int32()
        getattr(lib, name)(P(counts), ms, max_bits, P(bits), P(vals), ctypes.byref(mb))
        out.append((bits[:ms + 1].tolist(), vals[:ms + 1].tolist(), mb.value))
    if out[0] != out[1]:
        bad += 1
        print("MISMATCH huf maxSymbol %d maxBits %d counts %s\n  gpu %s\n  ref %s" % (ms, max_bits, counts.tolist(), out[0], out[1]))
    return out[1][2], max(out[1][0])


def huf_counts(kind, n):
    if kind == 0:   c = rng.integers(0, 2000, n)                           # flat-ish
    elif kind == 1: c = (rng.pareto(0.6, n) * 3).astype(np.int64) % 60000  # heavy tail: deep trees, the height limiter at work
    elif kind == 2: c = np.array([max(1, int(1.6 ** k)) for k in range(n)])[rng.permutation(n)] % 100000  # Fibonacci-like: maximal depth
    elif kind == 3: c = rng.integers(0, 3, n) * rng.integers(0, 50, n)     # many zeros, many ties
    elif kind == 4: c = np.where(rng.random(n) < 0.1, rng.integers(1000, 30000, n), rng.integers(0, 4, n))
    else:           c = rng.integers(1, 3, n)                              # all ties
    c = c.astype(np.int64)
    if (c > 0).sum() < 2:
        c[rng.integers(0, n)] = 5; c[(rng.integers(1, n) + 0) % n] += 7; c[0] += 1 if n > 1 else 0
    if (c[1:] > 0).sum() == 0:
        c[-1] = 3
    return c


limited = 0
for k in range(N):
    n = int(rng.choice([2, 3, 5, 13, 36, 64, 65, 100, 128, 200, 256]))
    c = huf_counts(k % 6, n)
    while c[-1] == 0:   # the callers' maxSymbol is the last symbol with a count
        c = c[:-1]
    if len(c) < 2 or (c > 0).sum() < 2 or (c[1:] > 0).sum() == 0:
        continue
    mb = int(rng.choice([5, 6, 7, 8, 9, 10, 11]))
    # (the leaves are the symbols with a count AND symbol 0, which keeps the table's first position whatever its count: more leaves than codes of
    # mb bits is what optimalNumberOfBits never asks for -- both sides then index their rank tables with length -1)
    if (1 << mb) < (c > 0).sum() + (1 if c[0] == 0 else 0):
        mb = 11
    got_mb, deepest = huf_case(c, mb)
    limited += 1 if deepest == mb else 0
print("huf: %d cases (%d at the height limit), %d mismatches so far" % (N, limited, bad), flush=True)


def norm_case(counts, table_log, force):
    global bad
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    ms = len(counts) - 1
    total = int(counts.sum())
    a = np.zeros(64, dtype=np.int16); b = np.zeros(64, dtype=np.int16)
    emu.unit_norm(P(counts), total, ms, table_log, force, P(a))
    ref.unit_ref_norm(P(counts), total, ms, table_log, force, P(b))
    if a[:ms + 1].tolist() != b[:ms + 1].tolist():
        bad += 1
        print("MISMATCH norm (second method forced: %d) tableLog %d counts %s\n  gpu %s\n  ref %s" % (force, table_log, counts.tolist(), a[:ms + 1].tolist(), b[:ms + 1].tolist()))
    return b[:ms + 1].copy()


norms = []
endings = [0, 0, 0]
for k in range(N):
    n = int(rng.choice([2, 4, 13, 29, 36, 53]))
    kind = k % 5
    if kind == 0:   c = rng.integers(0, 300, n)
    elif kind == 1: c = (rng.pareto(0.7, n) * 2).astype(np.int64) % 20000
    elif kind == 2: c = np.where(rng.random(n) < 0.15, rng.integers(500, 9000, n), rng.integers(0, 3, n))
    elif kind == 3: c = rng.integers(1, 4, n)
    else:           c = rng.integers(0, 2, n) * rng.integers(1, 40, n)
    c = c.astype(np.int64)
    while len(c) > 1 and c[-1] == 0:
        c = c[:-1]
    if (c > 0).sum() < 2:
        continue
    total = int(c.sum())
    lo = max(5, int(np.ceil(np.log2((c > 0).sum() + 1))))
    tl = int(rng.integers(lo, 10))
    if total < (1 << tl) // 2 and kind != 3:   # what the callers never pass: a table far larger than the input (optimalTableLog caps it)
        tl = max(5, int(np.log2(max(total, 2))) )
        if (1 << tl) < (c > 0).sum():
            continue
    norms.append((norm_case(c, tl, 0), tl))
    if k % 2 == 0 and total > (1 << tl):  # the second method alone (its own preconditions: more input than table)
        nm = norm_case(c, tl, 1)
        norms.append((nm, tl))
print("norm: %d count sets, %d mismatches so far" % (len(norms), bad), flush=True)


def write_case(norm, table_log, cap):
    global bad
    norm = np.ascontiguousarray(norm, dtype=np.int16)
    ms = len(norm) - 1
    a = np.full(600, 0xA5, dtype=np.uint8); b = np.full(600, 0xA5, dtype=np.uint8)
    sa = ctypes.c_int32()
    emu.unit_write(P(norm), ms, table_log, P(a), cap, ctypes.byref(sa))
    sb = ref.unit_ref_write(P(norm), ms, table_log, P(b), cap)
    ok = sa.value == sb and (sb < 0 or a[:sb].tolist() == b[:sb].tolist()) and (sb < 0 or (a[sb:] == 0xA5).all())
    if not ok:
        bad += 1
        print("MISMATCH write tableLog %d cap %d norm %s\n  gpu %d %s\n  ref %d %s" % (table_log, cap, norm.tolist(), sa.value, a[:max(sa.value, 0)].tolist(), sb, b[:max(sb, 0)].tolist()))
    return sb


writes = 0
for nm, tl in norms:
    if int(np.abs(nm).sum()) != (1 << tl):
        continue  # (a count set the normalisation cannot serve: the callers never produce one)
    size = write_case(nm, tl, 512)
    writes += 1
    if size > 0 and writes % 3 == 0:
        for cap in (size, size - 1, size + 1, 2, 1, 0):
            write_case(nm, tl, max(cap, 0))
# zero runs of every length up to 60 between two symbols that carry the table
for run in range(0, 61):
    for tl in (6, 9):
        nm = np.zeros(run + 3, dtype=np.int16)
        nm[0] = (1 << tl) - 5; nm[run + 1] = 4; nm[run + 2] = 1
        write_case(nm, tl, 512); writes += 1
        nm2 = np.zeros(run + 2, dtype=np.int16)  # the run in front
        nm2[run] = (1 << tl) - 1; nm2[run + 1] = -1
        if run > 0:
            write_case(nm2, tl, 512); writes += 1
print("write: %d tables, %d mismatches so far" % (writes, bad), flush=True)


def init_case(norm, table_log):
    global bad
    norm = np.ascontiguousarray(norm, dtype=np.int16)
    ms = len(norm) - 1
    res = []
    for lib, name in ((emu, "unit_fse_init"), (ref, "unit_ref_fse_init")):
        ns = np.zeros(512, dtype=np.int16); db = np.zeros(56, dtype=np.int32); df = np.zeros(56, dtype=np.int32)
        getattr(lib, name)(P(norm), ms, table_log, P(ns), P(db), P(df))
        res.append((ns[:1 << table_log].tolist(), db[:ms + 1].tolist(), df[:ms + 1].tolist()))
    if res[0] != res[1]:
        bad += 1
        print("MISMATCH fse_initialize tableLog %d norm %s\n  gpu %s\n  ref %s" % (table_log, norm.tolist(), res[0], res[1]))


inits = 0
for nm, tl in norms:
    if int(np.abs(nm).sum()) == (1 << tl) and len(nm) <= 53:
        init_case(nm, tl); inits += 1
for nm, tl in (([4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1], 6),
               ([1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1], 5),
               ([1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1], 6)):
    init_case(nm, tl); inits += 1  # the predefined distributions
print("fse_initialize: %d tables, %d mismatches in all" % (inits, bad))
sys.exit(1 if bad else 0)
