"""Differential fuzz of the encoders against the oracle: byte-identical streams on inputs of many shapes (random bytes, runs, spliced
text, periodic data, ragged lengths), every encoder variant.  usage: python tools/fuzz_encoders.py [cases] [seed] [codecs]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import common, oracle_lib
from tests.gpu_harness import GpuBatch

OPS = {"lz4": 1, "snappy": 3, "zstd": 5, "lz4frame": 7, "snappyframed": 9}
# (defaults first; the loop leaves the default set.  A pair: the variant and further options -- LZ4's window encoder in two tiers whatever the batch size, with one
# and two memory-tier wavefronts per workgroup: batches below 5 120 blocks otherwise take the wavefront-per-block kernel)
VARIANTS = {"lz4": [4, (4, {"lz4.compress.tier_min_blocks": 1, "lz4.compress.mem_waves": 1}), (4, {"lz4.compress.tier_min_blocks": 1, "lz4.compress.mem_waves": 2}), 1, 0],
            "snappy": [4, 2, 1, 0], "zstd": [3, 0, 2], "lz4frame": [None], "snappyframed": [None]}
DEFAULTS = {"lz4.compress.tier_min_blocks": 5120, "lz4.compress.mem_waves": 1}


def make_inputs(rng, n, max_len):
    texts = [d for _, d, _ in common.corpus_sample()]
    out = []
    for _ in range(n):
        kind = int(rng.integers(0, 7))
        length = int(rng.integers(0, max_len)) if rng.integers(0, 4) else int(rng.integers(0, 300))
        if kind == 0:
            b = rng.integers(0, 256, length, dtype=np.uint8).tobytes()
        elif kind == 1:    # runs of few symbols
            b = np.repeat(rng.integers(0, 4, length // 7 + 1, dtype=np.uint8), rng.integers(1, 14, length // 7 + 1))[:length].tobytes()
        elif kind == 2:    # periodic with a random period, a few defects
            period = rng.integers(0, 256, int(rng.integers(1, 70)), dtype=np.uint8)
            a = np.resize(period, length).copy()
            for _ in range(int(rng.integers(0, 5))):
                if length:
                    a[int(rng.integers(0, length))] ^= 0x55
            b = a.tobytes()
        elif kind == 3:    # text with random splices
            t = texts[int(rng.integers(0, len(texts)))]
            s = int(rng.integers(0, max(1, len(t) - 1)))
            b = (t[s:] + t[:s])[:length]
        elif kind == 4:    # pieces of different slices glued together
            parts = []
            while sum(map(len, parts)) < length:
                t = texts[int(rng.integers(0, len(texts)))]
                s = int(rng.integers(0, len(t)))
                parts.append(t[s:s + int(rng.integers(1, 5000))])
            b = b"".join(parts)[:length]
        elif kind == 5:    # low-entropy random
            b = (rng.integers(0, 256, length, dtype=np.uint8) & int(rng.choice([1, 3, 7, 15, 0x11]))).astype(np.uint8).tobytes()
        else:              # the reference's generator at a random ratio
            b = oracle_lib.load().random_generator(float(rng.choice([0.1, 0.25, 0.5, 0.75, 1.0])))[:length].tobytes()
        out.append(b)
    return out


def run(n_cases, seed, codecs=("lz4", "snappy", "zstd", "lz4frame", "snappyframed"), max_len=200000):
    rng = np.random.default_rng(seed)
    o = oracle_lib.load()
    gb = GpuBatch(0)
    bad = 0
    for codec in codecs:
        inputs = make_inputs(rng, n_cases, max_len)
        want = [o.compress(codec, b) for b in inputs]
        caps = [o.max_compressed_length(codec, len(b)) for b in inputs]
        for variant in VARIANTS[codec]:
            extra = {}
            if isinstance(variant, tuple):
                variant, extra = variant
            if variant is not None:
                gb.set_option("%s.compress.variant" % codec, variant)
            for k, v in extra.items():
                gb.set_option(k, v)
            outs, status, _ = gb.run(OPS[codec], inputs, caps, unaligned=True)
            wrong = sum(1 for i in range(len(inputs)) if status[i] != 0 or outs[i] != want[i])
            for i in range(len(inputs)):
                if (status[i] != 0 or outs[i] != want[i]) and wrong <= 5:
                    print("MISMATCH", codec, "variant", variant, "case", i, "len", len(inputs[i]), "status", status[i], flush=True)
            bad += wrong
            print("%s variant %s%s: %d inputs, %d mismatches" % (codec, variant, " %s" % extra if extra else "", len(inputs), wrong), flush=True)
            for k in extra:
                gb.set_option(k, DEFAULTS[k])
        if VARIANTS[codec][0] is not None:
            gb.set_option("%s.compress.variant" % codec, VARIANTS[codec][0])
    print("TOTAL MISMATCHES", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 1500, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                      tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("lz4", "snappy", "zstd", "lz4frame", "snappyframed")) else 0)
