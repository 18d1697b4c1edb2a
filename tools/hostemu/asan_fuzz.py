"""Memory-safety fuzz of the DECODERS on the CPU: the kernel sources under tools/hostemu, built with AddressSanitizer (libemu_all_asan.so), one
item per call, the item's source and destination each in an allocation of their own -- source padded only to the aligned 32-byte granules
its first and last byte lie in (an aligned piece that holds one byte of the item cannot leave the item's pages; anything further out
can), destination exactly its capacity -- so that every read or write a kernel makes outside what the ABI hands it is reported, for valid
streams, truncated streams, bit flips and garbage.  Run through tools/hostemu/run_asan_fuzz.sh (LD_PRELOAD of the ASan runtime).

  asan_fuzz.py <seed> <rounds> [lz4|snappy|zstd|zstdmb|containers ...]     (zstdmb: frames of several blocks)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import common, oracle_lib

o = oracle_lib.load()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", os.environ.get("HOSTEMU_LIB", "libemu_all_asan.so")))  # (HOSTEMU_LIB=libemu_all_ubsan.so: the same fuzz under UBSan)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
counters = np.zeros(64, dtype=np.int32)


GRANULE = 32  # the decoders read their input in aligned pieces of up to 32 bytes (achip_seqexec.h LaneFeed: two 16-byte loads per granule)


def envelope(size):
    """a uint8 array of `size` bytes (a multiple of GRANULE) that starts at a GRANULE-aligned address and ENDS where its allocation ends"""
    keep = []
    for _ in range(200):
        raw = np.zeros(size, dtype=np.uint8)
        if raw.ctypes.data % GRANULE == 0:
            return raw, raw
        raw2 = np.zeros(size + 16, dtype=np.uint8)
        if raw2.ctypes.data % GRANULE == 16:
            return raw2, raw2[16:]
        keep += [raw, raw2]  # (held, so that the allocator hands out other chunks)
    raise RuntimeError("no allocation with the wanted alignment")


def run1(call, data, cap, lead):
    """one item: source at offset `lead` (0..31) of a granule-aligned allocation that ends with the granule of the item's last byte"""
    n = len(data)
    raw, src = envelope(max((lead + n + GRANULE - 1) // GRANULE * GRANULE, GRANULE))
    if n:
        src[lead:lead + n] = np.frombuffer(bytes(data), dtype=np.uint8)
    dst = np.full(max(cap, 1), 0xA5, dtype=np.uint8)
    so = np.array([lead], dtype=np.int64); do = np.zeros(1, dtype=np.int64)
    sl = np.array([n], dtype=np.int32); cp = np.array([cap], dtype=np.int32)
    ol = np.full(1, -7, dtype=np.int32); st = np.full(1, -7, dtype=np.int32); eo = np.zeros(1, dtype=np.int64)
    r = call(P(src), P(so), P(sl), P(dst), P(do), P(cp), P(ol), P(st), P(eo), 1)
    assert r == 0, r
    return int(ol[0]), int(st[0]), dst


# Round 2's finding (the two-pass parsers' input feed anchored an EMPTY payload at `in` itself, so a Snappy stream that is its length prefix
# only and ends on a 32-byte boundary had the 32 bytes BEHIND it read) is fixed in achip_seqexec.h LaneFeed::init; such streams are part of
# the fuzz with no slack (prefix_only_cases below runs every one of them at the lead that puts their end on a granule boundary).
def prefix_only_cases():
    for n in (0, 1, 127, 128, 300, 65536, 1 << 21, (1 << 31) - 1):
        c = bytearray()
        v = n
        while v >= 0x80:
            c.append((v & 0x7F) | 0x80)
            v >>= 7
        c.append(v)
        yield bytes(c)


def mutate(rng, c):
    c = bytearray(c)
    kind = rng.integers(0, 6)
    if kind == 0 or len(c) < 4:
        return bytes(c)
    if kind == 1:
        return bytes(c[:int(rng.integers(0, len(c)))])
    if kind == 2:
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        return bytes(c)
    if kind == 3:
        i = int(rng.integers(0, len(c)))
        c[i:i + int(rng.integers(1, 9))] = bytes([255]) * int(rng.integers(1, 9))
        return bytes(c)
    if kind == 4:
        i = int(rng.integers(0, len(c)))
        return bytes(c[:i]) + rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes() + bytes(c[i:])
    return rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8).tobytes()


def plains(rng):
    sample = [d for _, d, _ in common.corpus_sample()]
    out = [d for _, d in common.HAND_CASES if len(d) < 5000]
    out += [s[:int(rng.integers(20, 6000))] for s in sample]
    out += [b[:int(rng.integers(100, 4000))] for b in common.synthetic_blocks(int(rng.integers(1, 1000)), 3)]
    out += [bytes(int(rng.integers(1, 3000))), rng.integers(0, 256, int(rng.integers(1, 2000)), dtype=np.uint8).tobytes(), b"ab" * int(rng.integers(1, 900))]
    return out


def batch_op(op):
    return lambda *a: lib.emu_batch(op, *a)


def family(name):
    if name == "lz4":
        return "lz4", [("rings 4 lanes", batch_op(44)), ("rings 16 lanes", batch_op(46)), ("rings 64 lanes", batch_op(48)), ("two-pass, a lane parsing", batch_op(24)), ("two-pass, a wavefront parsing", batch_op(26)), ("latency class", batch_op(49))]
    if name == "snappy":
        return "snappy", [("rings 4 lanes", batch_op(54)), ("rings 16 lanes", batch_op(56)), ("rings 64 lanes", batch_op(58)), ("two-pass, a lane parsing", batch_op(34)), ("two-pass, a wavefront parsing", batch_op(36)), ("latency class", batch_op(59))]
    if name == "zstd":
        return "zstd", [("pipeline + one-kernel decoder", lambda *a: lib.emu_zstd_full(*a, 1, 65536, P(counters))), ("one-kernel decoder", lambda *a: lib.emu_zstd_full(*a, 0, 65536, P(counters))),
                        ("pipeline, smallest passes", lambda *a: lib.emu_zstd_full(*a, 1, 16, P(counters)))]
    raise SystemExit("unknown family " + name)


def containers():
    for v in (0, 1, 2):
        yield "lz4frame", "LZ4 frame reader %d" % v, (lambda *a, v=v: lib.emu_lz4frame(v, *a)), lambda b: o.compress("lz4frame", b)
    for v in (1, 0, 2):
        yield "snappyframed", "x-snappy-framed reader %d" % v, (lambda *a, v=v: lib.emu_snappyframed(v, *a)), lambda b: o.compress("snappyframed", b)
    for codec, sn in (("lz4", 0), ("snappy", 1)):
        for v in (1, 0, 2):
            yield "hadoop-" + codec, "Hadoop %s reader %d" % (codec, v), (lambda *a, sn=sn, v=v: lib.emu_hadoop(0, sn, 1024, v, *a)), (lambda b, codec=codec: o.hadoop_compress(codec, b, 1024))


def zstd_multi_block(rng, rounds):
    """frames of SEVERAL blocks (the pipeline's multi-block stages: walk with links, per-block parse, one wavefront per frame executes): the
    shared multi-block inputs (tests/common.py: shaped data -- RLE-mode tables, treeless literals, raw / RLE blocks) up to 800 KB, as the
    oracle's encoder and as libzstd (pyarrow: many short blocks) write them, valid and damaged"""
    import pyarrow as pa
    frames = []
    for p in common.multi_block_plains():
        if len(p) <= 800000:
            frames.append((len(p), o.compress("zstd", p)))
            frames.append((len(p), pa.Codec("zstd", compression_level=3).compress(p, asbytes=True)))
    calls = 0
    decoders = family("zstd")[1]
    for _ in range(rounds):
        for n, good in frames:
            title, call = decoders[int(rng.integers(0, len(decoders)))]
            c = bytearray(good)
            kind = int(rng.integers(0, 4))
            if kind == 1:
                c = c[:int(rng.integers(0, len(c)))]
            elif kind == 2:
                for _ in range(int(rng.integers(1, 4))):
                    c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 3:  # damage near a block header: the walk's territory
                i = int(rng.integers(0, min(len(c), 400)))
                c[i] = int(rng.integers(0, 256))
            run1(call, bytes(c), n if rng.integers(0, 3) else int(rng.integers(0, n)), int(rng.integers(0, 32)))
            calls += 1
    return calls


def main():
    seed, rounds = int(sys.argv[1]), int(sys.argv[2])
    names = sys.argv[3:] or ["lz4", "snappy", "zstd", "containers"]
    rng = np.random.default_rng(seed)
    t = time.time()
    calls = 0
    for _ in range(rounds):
        ps = plains(rng)
        for name in names:
            if name == "zstdmb":
                calls += zstd_multi_block(rng, 1)
                continue
            if name == "containers":
                for _, title, call, enc in containers():
                    for b in ps[::3]:
                        c = mutate(rng, enc(b))
                        run1(call, c, max(len(b) + int(rng.integers(-3, 40)), 0), int(rng.integers(0, 32)))
                        calls += 1
                continue
            codec, decoders = family(name)
            if codec == "snappy":  # the prefix-only streams, their end on a granule boundary (and one byte off it)
                for c in prefix_only_cases():
                    for title, call in decoders:
                        for lead in ((-len(c)) % GRANULE, (-len(c) - 1) % GRANULE):
                            for cap in (0, 1, 64):
                                run1(call, c, cap, lead)
                                calls += 1
            for b in ps:
                good = o.compress(codec, b)
                for title, call in decoders:
                    c = mutate(rng, good)
                    cap = max(len(b) + int(rng.integers(-3, 40)), 0) if rng.integers(0, 4) else int(rng.integers(0, len(b) + 2))
                    run1(call, c, cap, int(rng.integers(0, 32)))
                    calls += 1
    print("asan fuzz seed %d: %d calls over %s, no report (%.0f s)" % (seed, calls, ", ".join(names), time.time() - t))


if __name__ == "__main__":
    main()
