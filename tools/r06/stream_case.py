"""One dumped stream of tools/fuzz_zstd_stream.py: the incremental reader, the batched decoder (with the pipeline's fallback statistics) and the oracle side by side.
python tools/r06/stream_case.py <file.zst>"""
import io, sys
import numpy as np
sys.path.insert(0, ".")
import aircompressor_amd as A
from tests import oracle_lib
from tests.gpu_harness import GpuBatch

z = open(sys.argv[1], "rb").read()
o = oracle_lib.load()
want = o.decompress("zstd", z, 16 << 20)
print("oracle (one-shot Java decoder restated): %d bytes" % len(want))
g = GpuBatch(0)
outs, st, eo = g.run(A.OP_ZSTD_DECOMPRESS, [z], [len(want)])
print("batched GPU decoder: status %d, %d bytes, equal %s; fallback items %d, by stage %s; multiblock items %d fast %d" % (
    st[0], len(outs[0]), outs[0] == want, g.codec.native.get_stat("zstd.decompress.fallback_items"),
    [g.codec.native.get_stat("zstd.decompress.fallback_stage%d" % k) for k in range(1, 7)],
    g.codec.native.get_stat("zstd.decompress.multiblock_items"), g.codec.native.get_stat("zstd.decompress.multiblock_fast_items")))
for size in (1 << 20, 65536):
    got = bytearray()
    try:
        with A.ZstdHipInputStream(io.BytesIO(z)) as s:
            buf = bytearray(size)
            while True:
                n = s.read_into(buf, 0, size)
                if n < 0:
                    break
                got += buf[:n]
        print("incremental reader (%d-byte reads): %d bytes, equal %s" % (size, len(got), bytes(got) == want))
    except Exception as e:
        print("incremental reader (%d-byte reads): FAILED after %d bytes (a prefix: %s): %r" % (size, len(got), bytes(got) == want[:len(got)], e))
