"""x-snappy-framed streams (snappy_frame.hip) on the CPU (tools/hostemu/libemu.so): reader variant 2 -- walk, the chunks through the two-pass
Snappy decoder with an arena asked for after the chunk count is known, CRC-32C verification, fold -- against the plaintext, on the streams the
oracle's restatement of SnappyFramedOutputStream writes (64 KiB chunks, stored when compression does not pay)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu.so"))
o = oracle_lib.load()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def run(variant, streams, caps):
    n = len(streams)
    src_off = np.zeros(n, dtype=np.int64); src_len = np.zeros(n, dtype=np.int32)
    dst_off = np.zeros(n, dtype=np.int64); dst_cap = np.array(caps, dtype=np.int32)
    pos = 64
    for i, f in enumerate(streams):
        src_off[i] = pos; src_len[i] = len(f); pos += len(f) + 5
    src = np.full(pos + 64, 0x5A, dtype=np.uint8)
    for i, f in enumerate(streams):
        src[src_off[i]:src_off[i] + len(f)] = np.frombuffer(f, dtype=np.uint8)
    pos = 64
    for i, c in enumerate(caps):
        dst_off[i] = pos; pos += c + 64
    dst = np.full(pos + 64, 0xA5, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.int32); status = np.full(n, -999, dtype=np.int32); err = np.zeros(n, dtype=np.int64)
    emu.emu_snappyframed(variant, P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(dst_cap), P(out_len), P(status), P(err), n)
    outs = [dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)]
    for i in range(n):
        hi = dst_off[i + 1] if i + 1 < n else len(dst)
        assert (dst[dst_off[i] + caps[i]:hi] == 0xA5).all(), "stream %d: wrote beyond its capacity" % i
    return outs, [int(x) for x in status]


def main():
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, 150000, dtype=np.uint8).tobytes()
    plains = [b"", b"x", whole[:1000], whole[:65536], whole[:65537], whole[:300000], whole[200000:900000], noise, b"ab" * 100000, whole[:100000] + noise + whole[:50000]]
    streams = [o.compress("snappyframed", p) for p in plains]
    bad = 0
    for pad in (0, 21):
        outs, status = run(2, streams, [len(p) + pad for p in plains])
        for i, p in enumerate(plains):
            if status[i] != 0 or outs[i] != p:
                bad += 1
                print("MISMATCH stream %d (len %d, pad %d): status %d, %d bytes" % (i, len(p), pad, status[i], len(outs[i])))
    print("snappy-framed reader, variant 2: %d streams x 2 capacities, %d mismatches" % (len(plains), bad))
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
