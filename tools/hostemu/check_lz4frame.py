"""LZ4 frames (lz4_frame.hip) on the CPU (tools/hostemu/libemu.so): reader variant 1 -- walk, the frames' blocks as one batch through the two-pass
block decoder with an arena asked for after the block count is known, fold -- against the plaintext, on frames as writers produce them
(the oracle's restatement of the Java writer: 4 MiB blocks, stored when not smaller; hand-built frames with content size / content checksum
and smaller block sizes).  Irregular items go to the wavefront-per-item kernel, which does not run here (tests/test_gpu_lz4_frame.py)."""
import ctypes, os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common, oracle_lib

emu = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu.so"))
o = oracle_lib.load()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def frame(content, size_id=7, content_size=False, content_checksum=False, stored=()):
    """one frame, independent blocks of the BD's size; block k stored when k in `stored` or when compression does not shrink it"""
    flg = (1 << 6) | (1 << 5) | (8 if content_size else 0) | (4 if content_checksum else 0)
    desc = bytes([flg, size_id << 4]) + (struct.pack("<q", len(content)) if content_size else b"")
    out = bytearray(struct.pack("<I", 0x184D2204)) + desc + bytes([(o.xxh32(desc) >> 8) & 0xFF])
    bmax = 1 << (8 + 2 * size_id)
    for k, off in enumerate(range(0, len(content), bmax)):
        b = content[off:off + bmax]
        c = o.compress("lz4", b)
        if k in stored or len(c) >= len(b):
            out += struct.pack("<I", len(b) | 0x80000000) + b
        else:
            out += struct.pack("<I", len(c)) + c
    out += struct.pack("<I", 0)
    if content_checksum:
        out += struct.pack("<I", o.xxh32(content))
    return bytes(out)


def run(variant, frames, caps):
    n = len(frames)
    src_off = np.zeros(n, dtype=np.int64); src_len = np.zeros(n, dtype=np.int32)
    dst_off = np.zeros(n, dtype=np.int64); dst_cap = np.array(caps, dtype=np.int32)
    pos = 64
    for i, f in enumerate(frames):
        src_off[i] = pos; src_len[i] = len(f); pos += len(f) + 3
    src = np.full(pos + 64, 0x5A, dtype=np.uint8)
    for i, f in enumerate(frames):
        src[src_off[i]:src_off[i] + len(f)] = np.frombuffer(f, dtype=np.uint8)
    pos = 64
    for i, c in enumerate(caps):
        dst_off[i] = pos; pos += c + 64
    dst = np.full(pos + 64, 0xA5, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.int32); status = np.full(n, -999, dtype=np.int32); err = np.zeros(n, dtype=np.int64)
    emu.emu_lz4frame(variant, P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(dst_cap), P(out_len), P(status), P(err), n)
    outs = [dst[dst_off[i]:dst_off[i] + max(int(out_len[i]), 0)].tobytes() for i in range(n)]
    for i in range(n):
        hi = dst_off[i + 1] if i + 1 < n else len(dst)
        assert (dst[dst_off[i] + caps[i]:hi] == 0xA5).all(), "frame %d: wrote beyond its capacity" % i
    return outs, [int(x) for x in status]


def main():
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    rng = np.random.default_rng(8)
    noise = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    cases = []
    for p in (whole[:1], whole[:1000], whole[:65536], whole[:300000], whole, noise, b"ab" * 100000, whole[:100000] + noise + whole[:50000]):
        cases.append((o.compress("lz4frame", p), p))                       # the Java writer's frame
        for size_id in (4, 5):                                               # 64 KiB and 256 KiB blocks: several per frame, some stored (noise)
            cases.append((frame(p, size_id, content_size=True, content_checksum=True), p))
            cases.append((frame(p, size_id, stored=(1,)), p))
    total = 0
    # variant 1: the listed blocks through the two-pass decoder; variant 2: by the probe -- whose statistics stay zero under the emulator, which reads as
    # "long sequences": the listed blocks through the ring decoders at 64 lanes per block (what fragments-like frames take on the device since round 5)
    for variant in (1, 2):
        bad = 0
        for pad in (0, 33):
            outs, status = run(variant, [f for f, _ in cases], [len(p) + pad for _, p in cases])
            for i, (f, p) in enumerate(cases):
                if status[i] != 0 or outs[i] != p:
                    bad += 1
                    print("MISMATCH variant %d case %d (len %d, pad %d): status %d, %d bytes" % (variant, i, len(p), pad, status[i], len(outs[i])))
        print("lz4 frame reader, variant %d: %d frames x 2 capacities, %d mismatches" % (variant, len(cases), bad))
        total += bad
    if total:
        sys.exit(1)


if __name__ == "__main__":
    main()
