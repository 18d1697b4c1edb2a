"""Block sharding for multi-GPU runs: one process per GPU, contiguous slices, no data-path collective.

The only cross-rank traffic is control: a barrier before/after the timed region and a MAX reduction of the
elapsed time (torch.distributed; RCCL on GPUs, gloo in the CPU tests).  SURVEY.md 8e.
"""
import numpy as np

from .batch import partition_blocks


def shard_for_rank(weights, world_size, rank):
    """(first, last) block indices of `rank`'s contiguous slice, balanced by `weights` (bytes per block)."""
    starts = partition_blocks(np.asarray(weights, dtype=np.int64), world_size)
    return int(starts[rank]), int(starts[rank + 1])


def aggregate_throughput(dist, local_bytes, local_seconds):
    """Whole-job bytes/s = sum over ranks of bytes / max over ranks of time.  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_bytes / local_seconds, local_bytes, local_seconds
    import torch
    t = torch.tensor([float(local_seconds)], dtype=torch.float64)
    b = torch.tensor([float(local_bytes)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return float(b.item()) / float(t.item()), float(b.item()), float(t.item())
