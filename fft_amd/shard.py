"""Batch sharding of the spectral mix across the GPUs of one node.

Every (batch, channel) column is independent (spectre.py:506, :545, :551 have no cross-batch term and the
gate is indexed by b, :515), so the path shards along B with NO collective in the data path: rank r owns a
contiguous run of batch elements, `memory_fft` (12.6 MB at the headline shape) is replicated.
"""
from __future__ import annotations

from typing import Tuple


def batch_shard(B: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of the batch elements rank `rank` owns; sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    q, r = divmod(B, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)
