"""N>1 path on CPU: two gloo ranks shard one batch with the product's partition rule (achip_partition_blocks), each rank
processes only its slice (the oracle stands in for the GPU here), and the union must equal the single-process result.
Also checks the throughput aggregation bench.py uses (sum of bytes / max of time)."""
import hashlib
import os
import socket

import numpy as np
import pytest

from tests import common


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, blocks, q):
    import torch.distributed as dist
    from aircompressor_amd.sharding import aggregate_throughput, shard_for_rank
    from tests import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_lib.load()
    comp = [o.compress("lz4", b) for b in blocks]
    weights = [len(b) + len(c) for b, c in zip(blocks, comp)]
    lo, hi = shard_for_rank(weights, world, rank)
    digests = []
    for i in range(lo, hi):  # this rank's slice only; no collective on the data path
        digests.append((i, hashlib.sha256(o.decompress("lz4", comp[i], len(blocks[i]))).hexdigest()))
    dist.barrier()
    rate, total_bytes, tmax = aggregate_throughput(dist, sum(len(blocks[i]) for i in range(lo, hi)), 1.0 + rank)
    q.put((rank, lo, hi, digests, rate, total_bytes, tmax))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_shard_a_batch_without_collectives():
    import torch.multiprocessing as mp
    blocks = [d for _, d, _ in common.corpus_sample()] + [b for b in common.synthetic_blocks(2, 12) if len(b)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, blocks, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, lo0, hi0, d0, rate0, tb0, tm0), (r1, lo1, hi1, d1, rate1, tb1, tm1) = results
    assert lo0 == 0 and hi0 == lo1 and hi1 == len(blocks)           # contiguous, disjoint, covering
    got = dict(d0 + d1)
    assert sorted(got) == list(range(len(blocks)))
    for i, b in enumerate(blocks):
        assert got[i] == hashlib.sha256(b).hexdigest()
    total = sum(len(b) for b in blocks)
    assert tb0 == tb1 == total and tm0 == tm1 == 2.0                # sum of bytes, max of time, identical on every rank
    assert abs(rate0 - total / 2.0) < 1e-6 and rate0 == rate1
    # byte balance of the split
    w = [len(b) for b in blocks]
    assert abs(sum(w[:hi0]) - sum(w[hi0:])) <= 2 * max(w)
