import numpy as np, sys
sys.path.insert(0,'.')
from tests import common, oracle_lib
from tests.gpu_harness import GpuBatch
o=oracle_lib.load()
codec='snappy'
rng = np.random.default_rng(99)
cases=[None]*6
sample = [d for _, d, _ in common.corpus_sample()[:4]] + common.synthetic_blocks(8, 6)[:6]
for b in sample:
    c = bytearray(o.compress(codec, b))
    cases.append((bytes(c), len(b) - 1)); cases.append((bytes(c[:len(c) // 2]), len(b))); cases.append((bytes(c[:-1]), len(b)))
    for _ in range(6):
        m = bytearray(c)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        cases.append((bytes(m), len(b))); cases.append((bytes(m), len(b) + 64))
c,cap=cases[19]
gb=GpuBatch(0)
for g in (1,2,4,8,16,32,64):
    gb.set_option('snappy.decompress.group', g)
    for reps in (1,3):
        outs,st,err=gb.run(2,[c]*reps,[cap]*reps)
        print(g,reps,st,err)
