"""Memory-safety check of the ENCODERS and writers on the CPU: libemu_enc_asan.so (access-granular lockstep + AddressSanitizer), one item per
call, source exactly its length and destination exactly its capacity in allocations of their own: every byte an encoder reads outside
[src, src + n) or touches outside [dst, dst + capacity) is reported.  Run through tools/hostemu/run_asan_fuzz.sh --enc.
Every encoder and writer: LZ4 and Snappy (the window encoders -- the defaults -- and the batch-probe ones), Zstd (window match finder, batch probes, one-kernel, the stream
writer), the LZ4 frame, x-snappy-framed and Hadoop writers; the bytes are compared with the oracle's as well.  (The library binds its
tracing callbacks to itself, -Bsymbolic-functions: the preloaded ASan runtime has no-op callbacks of the same names, and with those
taking the calls there is no lockstep.)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import common, oracle_lib

o = oracle_lib.load()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_enc_asan.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def run1(op, option, data, cap, buffer_size=262144):
    n = len(data)
    src = np.frombuffer(bytes(data), dtype=np.uint8).copy() if n else np.zeros(0, dtype=np.uint8)
    dst = np.full(max(cap, 1), 0xA5, dtype=np.uint8)
    z = np.zeros(1, dtype=np.int64)
    sl = np.array([n], dtype=np.int32); cp = np.array([cap], dtype=np.int32)
    ol = np.full(1, -7, dtype=np.int32); st = np.full(1, -7, dtype=np.int32); eo = np.zeros(1, dtype=np.int64)
    r = lib.emu_encode(op, P(src), P(z), P(sl), P(dst), P(z), P(cp), P(ol), P(st), P(eo), 1, option, buffer_size)
    assert r == 0, r
    return dst[:max(int(ol[0]), 0)].tobytes(), int(st[0])


def main():
    seed, rounds = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    sample = [d for _, d, _ in common.corpus_sample()]
    t = time.time()
    calls = bad = 0
    for _ in range(rounds):
        ps = [d for _, d in common.HAND_CASES if len(d) < 3000]
        ps += [s[int(rng.integers(0, 1000)):][:int(rng.integers(1, 5000))] for s in sample]
        ps += [b[:int(rng.integers(13, 3000))] for b in common.synthetic_blocks(int(rng.integers(1, 1000)), 2)]
        ps += [bytes(int(rng.integers(1, 2000))), rng.integers(0, 256, int(rng.integers(1, 1500)), dtype=np.uint8).tobytes(), b"abc" * int(rng.integers(1, 600)), b""]
        for b in ps:
            for title, op, option, bound, ref in (
                    ("lz4 (many matches per window)", 1, 4, lambda n: o.max_compressed_length("lz4", n), lambda b: o.compress("lz4", b)),
                    ("lz4 batch probes", 1, 1, lambda n: o.max_compressed_length("lz4", n), lambda b: o.compress("lz4", b)),
                    ("snappy (many matches per window)", 3, 4, lambda n: o.max_compressed_length("snappy", n), lambda b: o.compress("snappy", b)),
                    ("snappy batch probes", 3, 2, lambda n: o.max_compressed_length("snappy", n), lambda b: o.compress("snappy", b)),
                    ("snappy LDS input window", 3, 3, lambda n: o.max_compressed_length("snappy", n), lambda b: o.compress("snappy", b)),
                    ("zstd (window match finder)", 5, 3, lambda n: o.max_compressed_length("zstd", n), lambda b: o.compress("zstd", b)),
                    ("zstd batch probes", 5, 0, lambda n: o.max_compressed_length("zstd", n), lambda b: o.compress("zstd", b)),
                    ("zstd one kernel", 5, 2, lambda n: o.max_compressed_length("zstd", n), lambda b: o.compress("zstd", b)),
                    ("zstd stream", 14, 1, lambda n: o.lib.orc_zstd_stream_max_compressed_length(n), lambda b: o.zstd_stream_compress(b)),
                    ("lz4 frame", 7, 0, lambda n: o.max_compressed_length("lz4frame", n), lambda b: o.compress("lz4frame", b)),
                    ("snappy framed", 9, 1, lambda n: o.max_compressed_length("snappyframed", n), lambda b: o.compress("snappyframed", b)),
                    ("hadoop lz4", 11, 0, lambda n: o.hadoop_max_compressed_length("lz4", n, 1024), lambda b: o.hadoop_compress("lz4", b, 1024)),
                    ("hadoop snappy", 13, 0, lambda n: o.hadoop_max_compressed_length("snappy", n, 1024), lambda b: o.hadoop_compress("snappy", b, 1024))):
                out, st = run1(op, option, b, bound(len(b)), 1024)
                calls += 1
                if st != 0 or out != ref(b):
                    bad += 1
                    print("  MISMATCH %s: len %d status %d" % (title, len(b), st))
    print("asan encoders seed %d: %d calls, %d mismatches, no report (%.0f s)" % (seed, calls, bad, time.time() - t))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
