"""Batch data parallelism for the coupling-flow training step.

The reference has no distributed code; samples are independent (every op of hint.py:62-101 is
row-wise), so the batch is sharded into equal contiguous row blocks, one rank per GPU, weights
replicated, and the ONLY exchange per step is one all-reduce (sum) of the flat fp32 gradient
arena — RCCL over xGMI on the GPU box (`backend="nccl"`), gloo in the CPU tests.  The sum is
turned into the global-batch mean (train_unconditional.py:128-129 uses .mean() over the batch)
by the 1/world factor that the optimizer kernel applies BEFORE the +-5 clamp
(train_unconditional.py:140-141 clamps the final gradient).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous equal shards; the global batch must divide evenly so that the mean of the
    per-rank means equals the global mean"""
    if n_rows % world != 0:
        raise ValueError(f"global batch {n_rows} is not divisible by world size {world}")
    per = n_rows // world
    return rank * per, (rank + 1) * per


def allreduce_sum_(flat: torch.Tensor, group=None) -> float:
    """one collective over the whole flat gradient bucket; returns the scale (1/world) that
    turns the sum into the mean (applied later, fused into the optimizer kernel)"""
    rank, world = world_info(group)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def allreduce_buckets_(flat: torch.Tensor, cuts, group=None) -> float:
    """the same sum in several collectives: flat[cuts[i]:cuts[i+1]] each (last bucket first - the order in which the
    backward pass finishes them); element for element the single-bucket result"""
    edges = [0] + list(cuts) + [flat.numel()]
    scale = 1.0
    for a, b in reversed(list(zip(edges[:-1], edges[1:]))):
        if b > a:
            scale = allreduce_sum_(flat[a:b], group)
    return scale


def dp_step(local_grads: Callable[[], torch.Tensor], optimizer_step: Callable[[torch.Tensor, float], None],
            group=None) -> None:
    """The step protocol, independent of where the compute runs: local backward into the flat
    arena -> ONE all-reduce -> (scale, clamp, Adam) identically on every rank."""
    flat = local_grads()
    scale = allreduce_sum_(flat, group)
    optimizer_step(flat, scale)
