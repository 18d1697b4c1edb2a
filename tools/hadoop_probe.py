"""development aid: one small Hadoop-stream compress / decompress through the batch ABI (run under `timeout`)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.gpu_harness import GpuBatch
from tests import oracle_lib
o = oracle_lib.load()
codec = sys.argv[1]
size = int(sys.argv[2])
buf = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
ops = {"lz4": (10, 11), "snappy": (12, 13)}[codec]
gb = GpuBatch(0)
gb.set_option("hadoop.buffer_size", buf)
print("start", codec, size, buf, flush=True)
data = (b"hello hello world " * (size // 18 + 1))[:size]
cap = o.hadoop_max_compressed_length(codec, size, buf)
outs, st, eo = gb.run(ops[1], [data], [cap])
print("compress", st, len(outs[0]), outs[0] == o.hadoop_compress(codec, data, buf), flush=True)
outs2, st2, eo2 = gb.run(ops[0], [outs[0]], [size])
print("decompress", st2, outs2[0] == data, flush=True)
