"""Utterance sharding across the GPUs of one node and the final gather.

Every utterance is independent in all four stages, so ranks never exchange data on the data path.  The
only collective is the gather of results at the end; it is written against torch.distributed so the same
code runs over RCCL ("nccl" backend on ROCm, xGMI links) on the GPUs and over gloo in the CPU tests.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def partition(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Static longest-processing-time partition of utterance indices over `world` ranks: utterances sorted by
    length (ties by index) are dealt to the currently lightest rank.  Deterministic, identical on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(lengths[i])
    for p in parts:
        p.sort()
    return parts


def gather_ragged(local: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather 1-D tensors whose length differs per rank (pad to the longest, gather once, trim)."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    padded = torch.zeros(m, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[:s] for o, s in zip(out, sizes)]


def scatter_back(parts: List[List[int]], gathered: List[torch.Tensor], lengths_per_item: Sequence[int]) -> List[torch.Tensor]:
    """Undo `partition`: gathered[r] is the concatenation of rank r's items (in parts[r] order); returns the
    items in original utterance order."""
    out: List[torch.Tensor] = [None] * sum(len(p) for p in parts)  # type: ignore
    for r, idx in enumerate(parts):
        o = 0
        for i in idx:
            n = int(lengths_per_item[i])
            out[i] = gathered[r][o:o + n]
            o += n
    return out
