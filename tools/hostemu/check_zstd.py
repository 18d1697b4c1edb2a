"""Runs the Zstd decode pipeline (zstd_decompress_pipe.hip) on the CPU (tools/hostemu/libemu_zstd.so) over frames of the oracle's encoder
(the Java compressor restated) and of libzstd, and compares with the plaintext.  Items the pipeline hands to its fallback list (the
one-kernel decoder, which the emulator does not run) are counted; `--expect-fast` makes any of them a failure."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib, native_libs

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_zstd.so"))
o = oracle_lib.load()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def run(frames, caps, tile=0, exec_mode=1):
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
    k = emu.emu_zstd_pipe(P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(dst_cap), P(out_len), P(status), P(err), n, tile, exec_mode, P(fb))
    outs = []
    for i in range(n):
        outs.append(dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() if status[i] == 0 else None)
        lo = dst_off[i] + caps[i]
        hi = dst_off[i + 1] if i + 1 < n else len(dst)
        # (a few bytes beyond the capacity may be scribbled by whole-vector stores only where the product allows it: nowhere)
        assert (dst[lo:hi] == 0xA5).all(), "item %d: wrote beyond its capacity" % i
    return outs, status, sorted(int(x) for x in fb[:k])


def main():
    expect_fast = "--expect-fast" in sys.argv
    plains = [d for _, d in common.HAND_CASES if len(d) > 0] + [d[:131072] for _, d, _ in common.corpus_sample()[:8]] + common.synthetic_blocks(5, 6)
    plains = [p for p in plains if len(p) <= 131072]
    bad = 0; slow = 0; total = 0
    for name, enc in (("oracle", lambda p: o.compress("zstd", p)), ("libzstd-1", lambda p: native_libs.zstd_compress(p, 1)), ("libzstd-3", lambda p: native_libs.zstd_compress(p, 3)),
                      ("libzstd-9", lambda p: native_libs.zstd_compress(p, 9))):
        if name.startswith("libzstd") and not native_libs.available():
            continue
        frames = [bytes(enc(p)) for p in plains]
        for pad in (0, 37):
            outs, status, fb = run(frames, [len(p) + pad for p in plains])
            for i, p in enumerate(plains):
                total += 1
                if i in fb:
                    slow += 1
                elif status[i] != 0 or outs[i] != p:
                    bad += 1
                    print("MISMATCH %s item %d (len %d): status %d" % (name, i, len(p), status[i]))
        print("%s: %d frames, fallback list %s" % (name, len(frames), fb))
    print("zstd pipeline: %d cases, %d mismatches, %d on the fallback list" % (total, bad, slow))
    if bad or (expect_fast and slow):
        sys.exit(1)


if __name__ == "__main__":
    main()
