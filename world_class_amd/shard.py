"""Utterance sharding across the GPUs of one node and the final gather.

Every utterance is independent in all four stages, so ranks never exchange data on the data path.  The
only collective is the gather of results at the end; it is written against torch.distributed so the same
code runs over RCCL ("nccl" backend on ROCm, xGMI links) on the GPUs and over gloo in the CPU tests.
"""
import ctypes as C
from typing import List, Sequence

import torch
import torch.distributed as dist

# the C-ABI of include/world_class_shard.h (C / C++ hosts shard and gather through it; this module is the torch.distributed twin)
SHARD_SIGNATURES = {
    "wc_shard_partition": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "wc_gather_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p]),
    "wc_gather_to_root_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p]),
}


def partition_c(lengths: Sequence[int], world: int) -> List[List[int]]:
    """`partition` through the C-ABI (wc_shard_partition): the same deal, for hosts without Python"""
    from . import _check, lib
    L = lib()
    for name, (res, args) in SHARD_SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    n = len(lengths)
    arr = (C.c_int * max(n, 1))(*[int(v) for v in lengths])
    out = (C.c_int * max(n, 1))()
    _check(L.wc_shard_partition(arr, n, world, out))
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in range(n):
        parts[out[i]].append(i)
    return parts


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


def gather_ragged_to_root(local: torch.Tensor, sizes: Sequence[int], root: int = 0, group=None):
    """Gather 1-D tensors of known, rank-dependent lengths on ONE rank: every other rank posts one send, the root the
    world - 1 receives in one batch (RCCL: grouped ncclSend / ncclRecv, each peer over its own xGMI link at once) -- no
    padding to the longest shard, no copy of the total on the other ranks.  Returns the per-rank list on the root, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert len(sizes) == world and int(sizes[rank]) == local.numel()
    if rank != root:
        if local.numel() > 0:
            for w_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), root, group)]):
                w_.wait()
        return None
    out = [local if r == root else torch.empty(int(sizes[r]), dtype=local.dtype, device=local.device) for r in range(world)]
    ops = [dist.P2POp(dist.irecv, out[r], r, group) for r in range(world) if r != root and int(sizes[r]) > 0]
    if ops:
        for w_ in dist.batch_isend_irecv(ops):
            w_.wait()
    return out


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


class ShardLayout:
    """What one rank holds of a sharded batch: the utterance indices `partition` dealt to it and the packed layout of their
    samples, frames and output samples (include/world_class_c.h: utterance u's samples start at sum(x_length[<u]), its frames at
    row sum(f0_length[<u]), its output at sum(out_length[<u])).  Sizes come from the library's own host arithmetic
    (wc_get_samples = Harvest::getSamples, wc_synthesis_out_length), so the gloo tests shard exactly what the GPUs shard."""

    def __init__(self, x_lengths: Sequence[int], fs: int, frame_period: float, world: int, rank: int):
        from . import get_samples, synthesis_out_length
        self.world, self.rank = world, rank
        self.parts = partition(x_lengths, world)
        self.all_x_len = [int(n) for n in x_lengths]
        self.all_f_len = [get_samples(fs, n, frame_period) for n in self.all_x_len]
        self.all_y_len = [synthesis_out_length(f, frame_period, fs) for f in self.all_f_len]
        self.mine = self.parts[rank]
        self.x_len = [self.all_x_len[i] for i in self.mine]
        self.f_len = [self.all_f_len[i] for i in self.mine]
        self.y_len = [self.all_y_len[i] for i in self.mine]

    def gather_frames(self, local: torch.Tensor, width: int = 1, group=None) -> List[torch.Tensor]:
        """per-frame rows of `width` values of every rank -> list over ALL utterances in their original order"""
        return scatter_back(self.parts, gather_ragged(local.reshape(-1), group), [f * width for f in self.all_f_len])

    def gather_samples(self, local: torch.Tensor, group=None) -> List[torch.Tensor]:
        """output waveforms of every rank -> list over all utterances in their original order"""
        return scatter_back(self.parts, gather_ragged(local.reshape(-1), group), self.all_y_len)

    def gather_frames_to_root(self, local: torch.Tensor, width: int = 1, root: int = 0, group=None):
        """per-frame rows of every rank on the ROOT only (None elsewhere), all utterances in their original order"""
        sizes = [sum(self.all_f_len[i] for i in p) * width for p in self.parts]
        got = gather_ragged_to_root(local.reshape(-1), sizes, root, group)
        return None if got is None else scatter_back(self.parts, got, [f * width for f in self.all_f_len])

    def gather_samples_to_root(self, local: torch.Tensor, root: int = 0, group=None):
        """output waveforms of every rank on the ROOT only (None elsewhere), all utterances in their original order"""
        sizes = [sum(self.all_y_len[i] for i in p) for p in self.parts]
        got = gather_ragged_to_root(local.reshape(-1), sizes, root, group)
        return None if got is None else scatter_back(self.parts, got, self.all_y_len)


# ---- the few collectives of a sharded run besides the final gather: used by bench.py over RCCL and by the gloo tests ----------
def verify_group(world: int, local_id: int, device, group=None) -> None:
    """the process group really spans `world` ranks, each with its own device / local id (a launcher mistake would otherwise
    produce an N-GPU-labelled number from fewer GPUs)"""
    probe = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(probe, group=group)
    ids = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([local_id], dtype=torch.int64, device=device), group=group)
    seen = {int(t.item()) for t in ids}
    if int(probe.item()) != world or dist.get_world_size(group) != world or len(seen) != world:
        raise RuntimeError(f"process group does not span {world} distinct devices (ranks answering: {int(probe.item())}, local ids: {sorted(seen)})")


def max_over_ranks(values: Sequence[float], device, group=None) -> List[float]:
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(v) for v in t.tolist()]


def sum_over_ranks(value: int, device, group=None) -> int:
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())
