"""The N>1 path on CPU: utterance partition + final gather over torch.distributed with the gloo backend,
world_size 2 (the same code runs over RCCL on the GPUs, see bench.py / world_class_amd/shard.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from world_class_amd.shard import gather_ragged, partition, scatter_back


def test_partition_is_balanced_and_deterministic():
    rng = np.random.default_rng(0)
    lengths = rng.integers(1000, 500000, size=37).tolist()
    parts = partition(lengths, 4)
    assert sorted(i for p in parts for i in p) == list(range(37))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lengths)
    assert parts == partition(lengths, 4)
    assert partition([5, 5, 5], 1) == [[0, 1, 2]]
    assert partition([], 2) == [[], []]


def _stage(x):
    """Stand-in for the per-utterance device pipeline (independent per utterance, ragged output)."""
    return np.cumsum(x)[::80] * 0.5 + len(x)  # (len + 79) // 80 values


def _worker(rank, world, port, lengths, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = partition(lengths, world)
    mine = parts[rank]
    outs = [_stage(np.random.default_rng(100 + i).normal(size=lengths[i])) for i in mine]
    local = torch.from_numpy(np.concatenate(outs)) if outs else torch.zeros(0, dtype=torch.float64)
    gathered = gather_ragged(local)
    items = scatter_back(parts, gathered, [(lengths[i] + 79) // 80 for i in range(len(lengths))])
    if rank == 0:
        ret.put([t.numpy().copy() for t in items])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_run_equals_serial_run_gloo_world2():
    lengths = [1600, 8000, 240, 4800, 3200]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for i, n in enumerate(lengths):
        want = _stage(np.random.default_rng(100 + i).normal(size=n))
        assert np.array_equal(got[i], want)
