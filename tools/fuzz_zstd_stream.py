"""Differential fuzz of the INCREMENTAL Zstd stream twins on the GPU (achip_zstdstream_decompress_begin / _feed / _end, achip_zstdstream_compress_begin / _feed / _finish).

reader: streams of the oracle's ZstdOutputStream writer and of libzstd (one frame, frames back to back, with and without checksum / content size), valid, with flipped
        bytes, cut short or with bytes appended, read through ZstdHipInputStream from a source that hands out random piece sizes into buffers of random sizes -- against
        the reference's own ZstdInputStream transliterated (oracle/_ref/libref.so, ref_zstd_stream_decompress_partial): where it decodes the whole stream this reader
        must return the same bytes; where it fails this reader must fail too, and everything the reference delivered before failing must be a prefix of what this
        reader delivered (the reference keeps a window's worth of decoded bytes back, this reader delivers every block in front of the bad one).
writer: random plaintexts written through ZstdHipOutputStream in random piece sizes -- the sink must hold exactly the oracle writer's bytes.

    python tools/fuzz_zstd_stream.py [reader cases] [writer cases] [seed]"""
import ctypes, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyarrow as pa
import aircompressor_amd as A
from tests import common, oracle_lib

n_read = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n_write = int(sys.argv[2]) if len(sys.argv) > 2 else 150
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
o = oracle_lib.load()
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref.so"))
ref.ref_zstd_stream_decompress_partial.restype = ctypes.c_int64
ref.ref_zstd_stream_decompress_partial.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]

whole = b"".join(d for _, d, _ in common.corpus_sample())
tiled = whole * 8
noise = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()


class Dribble(io.RawIOBase):
    def __init__(self, data, most):
        self.data, self.at, self.most = data, 0, most

    def read(self, n=-1):
        n = self.most if n is None or n < 0 else min(n, self.most)
        piece = self.data[self.at:self.at + n]
        self.at += len(piece)
        return piece


def plaintext():
    kind = int(rng.integers(0, 6))
    n = int(rng.choice([0, 1, 300, 70000, 131072, 131073, 400000, 1 << 20, 3000000, (4 << 20) + 5]))
    if kind == 0:
        return tiled[int(rng.integers(0, 1000)):][:n]
    if kind == 1:
        return (noise * 12)[:n]
    if kind == 2:
        return bytes(n)
    if kind == 3:
        return (tiled[:n // 2] + noise[:n // 4] + bytes(n - n // 2 - n // 4))[:n]
    if kind == 4:
        return (whole[:5000] * (n // 5000 + 1))[:n]
    return tiled[:n]


def reference(stream, cap):
    out = ctypes.create_string_buffer(cap + 1)
    delivered, eo = ctypes.c_int64(0), ctypes.c_int64(0)
    src = ctypes.create_string_buffer(bytes(stream), len(stream))
    r = ref.ref_zstd_stream_decompress_partial(src, len(stream), out, cap + 1, 65536, ctypes.byref(delivered), ctypes.byref(eo))
    return r, out.raw[:max(delivered.value, 0)] if r < 0 else out.raw[:r]


zc = [pa.Codec("zstd", compression_level=l) for l in (1, 3, 9)]
bad = 0
fails = 0
deviations = 0
for case in range(n_read):
    frames, plain = [], b""
    for _ in range(int(rng.integers(1, 4))):
        p = plaintext()
        if len(plain) + len(p) > 9 << 20:
            break
        w = int(rng.integers(0, 5))
        frames.append(o.zstd_stream_compress(p) if w <= 1 else (o.compress("zstd", p) if w == 2 and len(p) <= 1 << 20 else zc[int(rng.integers(0, 3))].compress(p, asbytes=True)))
        plain += p
    z = bytearray(b"".join(frames))
    m = int(rng.integers(0, 8))
    if m <= 2 and len(z) > 0:
        for _ in range(int(rng.integers(1, 4))):
            z[int(rng.integers(0, len(z)))] = int(rng.integers(0, 256))
    elif m == 3 and len(z) > 1:
        z = z[:int(rng.integers(0, len(z)))]
    elif m == 4:
        z += bytes(rng.integers(0, 256, int(rng.integers(1, 12)), dtype=np.uint8))
    cap = len(plain) + (2 << 20)
    r, want = reference(z, cap)
    truncated = r == cap + 1  # (a mutated size field: the stream decodes to more than this harness lets the reference write -- what it delivered is a prefix)
    got = bytearray()
    failed = False
    size = int(rng.choice([7, 1000, 65536, 77777, 1 << 20, 3 << 20]))
    if size == 7 and len(plain) > 200000:
        size = 77777
    try:
        with A.ZstdHipInputStream(Dribble(bytes(z), int(rng.choice([1, 13, 4096, 70001, 1 << 20, 8 << 20])))) as s:
            buf = bytearray(size)
            while True:
                n = s.read_into(buf, 0, size)
                if n < 0:
                    break
                got += buf[:n]
                if len(got) > len(plain) + (2 << 20):
                    raise IOError("runaway")
    except (A.MalformedInputException, IOError, ValueError):
        failed = True
    fails += 1 if r < 0 else 0
    ok = (r >= 0 and not failed and bytes(got) == want) or (r < 0 and failed and bytes(got[:len(want)]) == want)
    if not ok and truncated and bytes(got[:len(want)]) == want:
        ok = True  # (both decode beyond the harness' cap: this reader's "runaway" guard stopped it behind the reference's last byte)
    if not ok and r < 0 and not failed and bytes(got[:len(want)]) == want:
        # The documented deviation: a RAW / RLE block that says more than 128 KiB.  ZstdIncrementalFrameDecompressor.java:204-226 copies / fills what
        # the block says IF its window buffer happens to have the room (its resize :291-340 only guarantees 128 KiB: "window buffer is too small" otherwise -- which
        # depends on how the caller has been reading); the one-shot ZstdFrameDecompressor decodes such a frame, and so does this reader.  Counted, not a mismatch,
        # where the ONE-SHOT Java decoder (the oracle's restatement) reads the stream to exactly this reader's bytes.
        try:
            if o.decompress("zstd", bytes(z), len(got) + 16) == bytes(got):
                deviations += 1
                ok = True
        except oracle_lib.OracleError:
            pass
    if not ok:
        bad += 1
        if bad <= 8:
            print("MISMATCH reader case %d: mutation %d, %d stream bytes, reference r=%d delivered %d | this reader failed=%s delivered %d" % (case, m, len(z), r, len(want), failed, len(got)), flush=True)
            if os.environ.get("ACHIP_FUZZ_DUMP"):  # (keep the stream for a closer look)
                os.makedirs(os.environ["ACHIP_FUZZ_DUMP"], exist_ok=True)
                open(os.path.join(os.environ["ACHIP_FUZZ_DUMP"], "stream_case_%d.zst" % case), "wb").write(bytes(z))
print("reader: %d streams (%d the reference refuses, %d of them for an oversized RAW / RLE block that the one-shot reference decoder and this reader decode alike), %d mismatches" % (
    n_read, fails, deviations, bad), flush=True)

wbad = 0
for case in range(n_write):
    p = plaintext() + (plaintext() if rng.integers(0, 2) else b"")
    class Sink(io.BytesIO):  # (the stream closes its sink: keep the bytes)
        def close(self):
            self.final = self.getvalue()
            super().close()
    sink = Sink()
    with A.ZstdHipOutputStream(sink) as s:
        at = 0
        while at < len(p):
            n = int(rng.choice([1, 100, 4096, 131072, 131073, 1 << 20, 5 << 20]))
            if n < 4096 and len(p) - at > 300000:
                n = 131072
            s.write(p[at:at + n])
            at += n
    want = o.zstd_stream_compress(p)
    if sink.final != want:
        wbad += 1
        if wbad <= 8:
            print("MISMATCH writer case %d: %d plaintext bytes, %d stream bytes against the oracle's %d" % (case, len(p), len(sink.final), len(want)), flush=True)
print("writer: %d streams, %d mismatches" % (n_write, wbad), flush=True)
print("TOTAL MISMATCHES", bad + wbad)
sys.exit(1 if bad + wbad else 0)
