"""Targeted differential fuzz of the Zstd sequence stage's END-OF-STREAM behaviour (round 6): frames whose sequence bit streams are damaged where they are read LAST
-- their first bytes --, or whose sequence counts are off by a few: extra bits that run past a stream's start (the Java reader executes such a sequence from what
its wrapped shifts return and only its NEXT load() notices -- which is no error when no sequence is left: ZstdFrameDecompressor.java:395-399, BitInputStream.java:171-204),
overflow with sequences left, exact exhaustion.
  * the GPU pipeline (variant 1) and the one-kernel decoder (variant 0) against the oracle: status, error offset, plaintext;
  * the incremental reader (ZstdHipInputStream, which has no fallback decoder behind its stages): whatever the oracle decodes it must read, to the same bytes.
    python tools/fuzz_zstd_tail.py [cases] [seed]"""
import io
import sys
import numpy as np
sys.path.insert(0, ".")


def sections(f):
    """[(offset of the sequence count, offset behind the sequences header, block end, compression modes)] of a frame's compressed blocks"""
    fhd = f[4]
    at = 5 + (0 if fhd & 0x20 else 1) + [0, 1, 2, 4][fhd & 3] + [1 if fhd & 0x20 else 0, 2, 4, 8][fhd >> 6]
    out = []
    while at + 3 <= len(f):
        h = int.from_bytes(f[at:at + 3], "little")
        last, t, size = h & 1, (h >> 1) & 3, h >> 3
        body = at + 3
        if t == 2 and size > 8:
            b = f[body:body + size]
            lt, sf = b[0] & 3, (b[0] >> 2) & 3
            if lt < 2:
                hs = 1 if sf in (0, 2) else (2 if sf == 1 else 3)
                regen = (b[0] >> 3) if hs == 1 else ((b[0] >> 4) | (b[1] << 4) if hs == 2 else (b[0] >> 4) | (b[1] << 4) | (b[2] << 12))
                comp = regen if lt == 0 else 1
            else:
                hs = 3 if sf < 2 else (4 if sf == 2 else 5)
                v = int.from_bytes(b[0:hs], "little")
                bits = 10 if hs == 3 else (14 if hs == 4 else 18)
                comp = (v >> (4 + bits)) & ((1 << bits) - 1)
            sq = hs + comp
            if sq < size:
                n0 = b[sq]
                sh = 1 if n0 < 128 else (2 if n0 < 255 else 3)
                if n0 != 0 and sq + sh + 1 < size:
                    out.append((body + sq, body + sq + sh + 1, body + size, b[sq + sh]))
        at = body + (1 if t == 1 else size)
        if last:
            break
    return out


def run(n_cases, seed, stream_reader=True):
    """returns the number of mismatches"""
    import pyarrow as pa
    import aircompressor_amd as A
    from tests import common, oracle_lib
    from tests.gpu_harness import GpuBatch
    from tests.oracle_lib import OracleError
    rng = np.random.default_rng(seed)
    o = oracle_lib.load()
    whole = b"".join(d for _, d, _ in common.corpus_sample())
    zc = pa.Codec("zstd", compression_level=3)
    gb = {1: GpuBatch(0, options={"zstd.decompress.variant": 1}), 0: GpuBatch(0, options={"zstd.decompress.variant": 0})}
    bad = done = fast = read = 0
    kinds = [0, 0, 0, 0, 0]
    while done < n_cases:
        items, caps = [], []
        for _ in range(min(256, n_cases - done)):
            n = int(rng.choice([3000, 20000, 70000, 131072]))
            at = int(rng.integers(0, len(whole) - n))
            p = whole[at:at + n] if rng.integers(0, 4) else (whole[at:at + 700] * (n // 700 + 1))[:n]
            f = bytearray(zc.compress(p, asbytes=True) if rng.integers(0, 2) else o.compress("zstd", p))
            secs = sections(f)
            if secs:
                cnt, tbl, end, modes = secs[int(rng.integers(0, len(secs)))]
                k = int(rng.integers(0, 5))
                kinds[k] += 1
                if k == 0:    # a bit in the bit stream's first bytes (table descriptions may lie between: reach a little further when the modes say so)
                    q = tbl + int(rng.integers(0, 6 if modes == 0 else 40))
                    if q < end:
                        f[q] ^= 1 << int(rng.integers(0, 8))
                elif k == 1:  # the sequence count, up or down a little
                    if 1 < f[cnt] < 127:
                        f[cnt] = int(np.clip(f[cnt] + int(rng.integers(-2, 4)), 1, 127))
                    elif 128 <= f[cnt] < 255:
                        f[cnt + 1] = (f[cnt + 1] + int(rng.integers(-2, 4))) & 0xFF
                elif k == 2:  # the stream's last byte (its padding marker: where reading STARTS)
                    f[end - 1] = int(rng.integers(1, 256))
                elif k == 3:  # several bytes right behind the sequences header
                    for _ in range(3):
                        q = tbl + int(rng.integers(0, 48))
                        if q < end - 1:
                            f[q] = int(rng.integers(0, 256))
                # k == 4: untouched
            items.append(bytes(f))
            caps.append(len(p) + int(rng.integers(0, 2)) * 100)
        want = []
        for f, c in zip(items, caps):
            try:
                want.append((0, 0, o.decompress("zstd", f, c)))
            except OracleError as e:
                want.append((e.status, e.offset, None))
        for v in (1, 0):
            outs, st, eo = gb[v].run(A.OP_ZSTD_DECOMPRESS, items, caps)
            if v == 1:
                fast += len(items) - gb[v].codec.native.get_stat("zstd.decompress.fallback_items")
            for i, (w, out, s, e) in enumerate(zip(want, outs, st, eo)):
                if not (s == w[0] and (s != 0 or out == w[2]) and (s == 0 or e == w[1])):
                    bad += 1
                    if bad <= 10:
                        print("MISMATCH variant %d item %d: oracle (%d, %d, %s) GPU (%d, %d, %d bytes)" % (v, done + i, w[0], w[1], None if w[2] is None else len(w[2]), s, e, len(out)), flush=True)
        if stream_reader:
            for i, (w, f) in enumerate(zip(want, items)):
                if w[0] == 0 and (done + i) % 4 == 0:
                    read += 1
                    try:
                        got = A.ZstdHipInputStream(io.BytesIO(f)).read()
                    except (A.MalformedInputException, IOError, ValueError) as e:
                        got = repr(e)
                    if got != w[2]:
                        bad += 1
                        if bad <= 10:
                            print("MISMATCH incremental reader, item %d: oracle %d bytes, reader %s" % (done + i, len(w[2]), got if isinstance(got, str) else "%d bytes" % len(got)), flush=True)
        done += len(items)
    print("zstd end-of-stream fuzz: %d damaged frames (by kind %s), %d of them decoded on the pipeline's fast path, %d read through the incremental reader, %d mismatches" % (
        done, kinds, fast, read, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
