"""Differential fuzz of the block decoders against the oracle (status, error offset, plaintext): random mutations, truncations
and capacity changes of real and synthetic streams, every decoder variant.  usage: python tools/fuzz_decoders.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import common, oracle_lib
from tests.gpu_harness import GpuBatch
from tests.oracle_lib import OracleError

def run(n_cases, seed, codecs=("lz4", "snappy"), big=False):
    rng = np.random.default_rng(seed)
    o = oracle_lib.load()
    gb = GpuBatch(0)
    blocks = [d for _, d, _ in common.corpus_sample()] + common.synthetic_blocks(9, 24) + [d for _, d in common.HAND_CASES if len(d) > 0]
    blocks += [b[:n] for b in blocks[:6] for n in (17, 300, 5000)]
    if big:  # long runs, long periods, blocks beyond 64 KiB: the wide / doubling / memcpy paths of the decoders
        text = b"".join(d for _, d, _ in common.corpus_sample()[:5])
        blocks += [bytes(1 << 20), bytes(range(256)) * 2048, (b"abc" * 100000)[:250001], text, text[:100000] + bytes(50000) + text[:70000],
                   bytes(rng.integers(0, 256, 300000, dtype=np.uint8)), (bytes(rng.integers(0, 256, 1000, dtype=np.uint8)) * 400)]
    OPS = {"lz4": 0, "snappy": 2, "zstd": 4, "lz4frame": 6, "snappyframed": 8, "lz4hadoop": 10, "snappyhadoop": 12}  # (round 6: the Hadoop block-stream readers)
    HADOOP = {"lz4hadoop": "lz4", "snappyhadoop": "snappy"}
    VARIANTS = {"lz4": [1, 13, 7, 71], "snappy": [1, 13, 7, 71],  # (71: variant 7 with the lane-per-block parser -- batches of this size take the wavefront-per-block one;
                                                                # 13: the ring decoders' latency class -- a wavefront and 128 KiB of LDS history per block -- for every batch size)
                 "zstd": [1, 0], "lz4frame": [None], "snappyframed": [None], "lz4hadoop": [3, 1, 2, 0], "snappyhadoop": [3, 1, 2, 0]}  # (hadoop.decompress.variant)


    def expect(codec, data, cap):
        try:
            if codec in HADOOP:
                return 0, 0, o.hadoop_decompress(HADOOP[codec], data, cap)
            return 0, 0, o.decompress(codec, data, cap)
        except OracleError as e:
            return e.status, e.offset, None


    bad = 0
    for codec in codecs:
        comp = [o.hadoop_compress(HADOOP[codec], b, buffer_size=(262144, 8192, 1024)[i % 3]) if codec in HADOOP else o.compress(codec, b) for i, b in enumerate(blocks)]  # (several chunks per stream)
        caps = [len(b) for b in blocks]
        if codec == "snappy":  # streams of random elements of every kind (what the Java encoder never writes: 4-byte offsets, runs with length bytes, runs behind runs)
            for target in (40, 500, 3000, 20000, 70000) + ((150000, 400000) if big else ()):
                for _ in range(4):
                    c, n = common.snappy_random_stream(rng, target)
                    comp.append(c)
                    caps.append(n)
        if codec == "zstd":  # third-party frames as well (no checksum, other header forms, treeless / repeat modes never from the Java encoder)
            try:
                import pyarrow as pa
                z = pa.Codec("zstd", compression_level=3)
                comp = comp + [z.compress(b, asbytes=True) for b in blocks]
                blocks = blocks + blocks
                caps = caps + caps
            except ImportError:
                pass
        cases = []
        for _ in range(n_cases):
            k = int(rng.integers(0, len(comp)))
            c = bytearray(comp[k])
            cap = caps[k]
            kind = int(rng.integers(0, 6)) if len(c) else 3  # (an empty stream -- what the Hadoop writers make of no input -- can only be extended)
            if kind <= 2:      # 1..4 byte mutations
                for _ in range(int(rng.integers(1, 5))):
                    c[int(rng.integers(0, len(c)))] = int(rng.integers(0, 256))
            elif kind == 3:    # truncation / extension
                if rng.integers(0, 2) and len(c) > 1:
                    c = c[:int(rng.integers(1, len(c)))]
                else:
                    c += bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
            elif kind == 4:    # capacity
                cap = max(0, cap + int(rng.integers(-40, 41)))
            else:              # a mutation near the start (headers, first tokens) and a capacity change
                c[int(rng.integers(0, min(len(c), 24)))] ^= 1 << int(rng.integers(0, 8))
                cap = max(0, cap + int(rng.integers(-8, 9)))
            cases.append((bytes(c), cap))
        want = [expect(codec, c, cap) for c, cap in cases]
        for variant in VARIANTS[codec]:
            if variant is not None and codec in HADOOP:
                gb.set_option("hadoop.decompress.variant", variant)
            elif variant is not None:
                gb.set_option("%s.decompress.variant" % codec, 7 if variant == 71 else (1 if variant == 13 else variant))
                gb.set_option("decompress.latency_max_blocks", 65536 if variant == 13 else 0)
                if codec in ("lz4", "snappy"):
                    gb.set_option("%s.decompress.parse" % codec, 1 if variant == 71 else 0)
            outs, status, err = gb.run(OPS[codec], [c for c, _ in cases], [cap for _, cap in cases], unaligned=True)
            wrong = 0
            for i, (est, eoff, eout) in enumerate(want):
                ok = status[i] == est and (err[i] == eoff if est != 0 else outs[i] == eout)
                if not ok:
                    wrong += 1
                    if wrong <= 5:
                        print("MISMATCH", codec, "variant", variant, "case", i, "gpu", status[i], err[i], "oracle", est, eoff, "len", len(cases[i][0]), "cap", cases[i][1], flush=True)
            bad += wrong
            n_err = sum(1 for w in want if w[0] != 0)
            print("%s variant %s: %d cases (%d malformed), %d mismatches" % (codec, variant, len(cases), n_err, wrong), flush=True)
        if codec in ("lz4", "snappy"):
            gb.set_option("%s.decompress.parse" % codec, 0)
        if codec in HADOOP:
            gb.set_option("hadoop.decompress.variant", 3)
        gb.set_option("decompress.latency_max_blocks", 256)
    print("TOTAL MISMATCHES", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 6000, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                      tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("lz4", "snappy"), big=len(sys.argv) > 4) else 0)
