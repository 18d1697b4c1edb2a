"""Hadoop LZ4 / Snappy block streams (hadoop_streams.hip) on the CPU (tools/hostemu/libemu.so): the reader's variant 2 -- walk, the chunks
through the two-pass decoders with an arena asked for after the chunk count is known, fold, the serial kernel for everything else --
against the oracle, on the streams every writer produces.  (Variant 1 runs the chunks through the ring decoders at 4 / 16 lanes per chunk,
which lean on the hardware's in-order memory pipeline and do not run here.)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu.so"))
o = oracle_lib.load()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def run(snappy, buffer_size, variant, streams, caps):
    n = len(streams)
    src_off = np.zeros(n, dtype=np.int64); src_len = np.zeros(n, dtype=np.int32)
    dst_off = np.zeros(n, dtype=np.int64); dst_cap = np.array(caps, dtype=np.int32)
    pos = 64
    for i, f in enumerate(streams):
        src_off[i] = pos; src_len[i] = len(f); pos += len(f) + 7
    src = np.full(pos + 64, 0x5A, dtype=np.uint8)
    for i, f in enumerate(streams):
        src[src_off[i]:src_off[i] + len(f)] = np.frombuffer(f, dtype=np.uint8)
    pos = 64
    for i, c in enumerate(caps):
        dst_off[i] = pos; pos += c + 64
    dst = np.full(pos + 64, 0xA5, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.int32); status = np.full(n, -999, dtype=np.int32); err = np.zeros(n, dtype=np.int64)
    emu.emu_hadoop(0, 1 if snappy else 0, buffer_size, variant, P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(dst_cap), P(out_len), P(status), P(err), n)
    outs = [dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)]
    for i in range(n):
        hi = dst_off[i + 1] if i + 1 < n else len(dst)
        assert (dst[dst_off[i] + caps[i]:hi] == 0xA5).all(), "stream %d: wrote beyond its capacity" % i
    return outs, [int(x) for x in status], [int(x) for x in err]


def main():
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, 90000, dtype=np.uint8).tobytes()
    plains = [b"", b"x", whole[:1000], whole[:70000], whole[:300000], whole[100000:700000], noise, b"ab" * 150000, whole[:218422], whole[:259523 * 2]]
    bad = 0
    for codec, snappy in (("lz4", False), ("snappy", True)):
        for buf in (262144, 70000, 4096):
            streams = [o.hadoop_compress(codec, p, buf) for p in plains]
            caps = [len(p) for p in plains]
            cases = list(zip(streams, caps))
            # (valid streams only: anything irregular goes to the wavefront-per-stream kernel, whose block decoder moves bytes between lanes in
            # hardware order and does not run here -- tests/test_gpu_hadoop.py covers it on the GPU)
            outs, status, err = run(snappy, buf, 2, [c for c, _ in cases], [cap for _, cap in cases])
            m = 0
            for i, (c, cap) in enumerate(cases):
                try:
                    want = o.hadoop_decompress(codec, c, cap, buf); est = 0; eoff = 0
                except oracle_lib.OracleError as e:
                    want = None; est = e.status; eoff = e.offset
                if status[i] != est or (est == 0 and outs[i] != want) or (est != 0 and err[i] != eoff):
                    m += 1
                    print("MISMATCH %s buffer %d case %d: status %d@%d, oracle %d@%d" % (codec, buf, i, status[i], err[i], est, eoff))
            bad += m
            print("hadoop %s reader, variant 2, buffer %d: %d streams, %d mismatches" % (codec, buf, len(cases), m))
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
