"""Differential fuzz of an ENCODER variant on the CPU emulator (libemu_enc.so) against the oracle:  fuzz_enc.py <seed> <count> [lz4|snappy|zstd] [variant]
Structured inputs -- short and long incompressible stretches (the skip schedule grows, batches leave the LDS window and come back), near and
far copies of 4..1500 bytes, byte runs, small alphabets -- of 13 bytes to 40 KB, plus, for seed 0, the sizes around the window's chunk
boundaries and around 64 KiB (Snappy sub-blocks, LZ4's wide table)."""
import sys, os, time, ctypes
ROOT_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT_); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from emu_harness import EmuBatch, P, ROOT
from tests import oracle_lib, common
o = oracle_lib.load()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "hostemu", "libemu_enc.so"))
class EncBatch(EmuBatch):
    def __init__(self, option): self.lib = lib; self.options = {}; self.option = option
    def _call(self, op, src, src_off, src_len, dst, dst_off, caps, out_len, status, err, n):
        return self.lib.emu_encode(op, P(src), P(src_off), P(src_len), P(dst), P(dst_off), P(caps), P(out_len), P(status), P(err), n, self.option, 262144)
seed = int(sys.argv[1]); count = int(sys.argv[2]); codec = sys.argv[3] if len(sys.argv) > 3 else "lz4"; opnum = {"lz4": 1, "snappy": 3, "zstd": 5}[codec]; variant = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.default_rng(seed)
def gen(n):
    out = bytearray()
    while len(out) < n:
        kind = rng.integers(0, 6)
        if kind == 0:   out += rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
        elif kind == 1: out += rng.integers(0, 256, int(rng.integers(200, 6000)), dtype=np.uint8).tobytes()   # long incompressible stretch: skip steps grow
        elif kind == 2 and len(out) > 8:
            d = int(rng.integers(1, min(len(out), 70000) + 1)); l = int(rng.integers(4, 40))
            for _ in range(l): out.append(out[-d])
        elif kind == 3 and len(out) > 8:
            d = int(rng.integers(1, min(len(out), 3000) + 1)); l = int(rng.integers(16, 1500))
            for _ in range(l): out.append(out[-d])
        elif kind == 4: out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 300))
        else: out += bytes(rng.integers(97, 101, int(rng.integers(5, 200)), dtype=np.uint8))
    return bytes(out[:n])
sizes = [0, 1, 5, 13, 14, 15, 16, 1023, 1024, 1025, 1040, 2047, 2048, 2049, 3071, 3072, 3100, 4096, 65535, 65536, 65537, 70000, 131072, 131073, 150000]
blocks = [gen(s) for s in sizes] if seed == 0 else []
blocks += [gen(int(rng.integers(13, 40000))) for _ in range(count)]
caps = [o.max_compressed_length(codec, len(b)) for b in blocks]
t = time.time()
outs, status, _ = EncBatch(variant).run(opnum, blocks, caps)
bad = [i for i, (b, c, s) in enumerate(zip(blocks, outs, status)) if s != 0 or c != o.compress(codec, b)]
print("%s variant %d, seed %d: %d inputs, %d bytes, mismatches %s (%.0f s)" % (codec, variant, seed, len(blocks), sum(map(len, blocks)), [(i, len(blocks[i])) for i in bad], time.time() - t))
sys.exit(1 if bad else 0)
