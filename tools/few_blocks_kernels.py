"""Per-kernel times of the two-pass decoders on FEW blocks (the wavefront-per-block parsers): run under
   rocprofv3 --kernel-trace --stats -- python tools/few_blocks_kernels.py [blocks] [block bytes]
and read the kernel_stats; prints the wall time per call beside it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import aircompressor_amd as A
from tests import common, oracle_lib
from tests.gpu_harness import GpuBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
o = oracle_lib.load()
text = b"".join(d for _, d, _ in common.corpus_sample())
text = (text * (1 + n * bs // len(text) + 1))
blocks = [text[i * bs:(i + 1) * bs] for i in range(n)]
gb = GpuBatch(0)
for codec, dop in (("lz4", A.OP_LZ4_DECOMPRESS), ("snappy", A.OP_SNAPPY_DECOMPRESS)):
    comp = [o.compress(codec, b) for b in blocks]
    gb.set_option("%s.decompress.variant" % codec, 7)
    gb.set_option("%s.decompress.parse" % codec, 2)
    for _ in range(3):
        outs, st, _ = gb.run(dop, comp, [bs] * n)
    assert outs == blocks
    t0 = time.perf_counter()
    for _ in range(10):
        gb.run(dop, comp, [bs] * n)
    print("%s: %d blocks of %d bytes, two passes with the wavefront parser: %.2f ms per call (wall, copies included)" % (codec, n, bs, (time.perf_counter() - t0) * 100), flush=True)
