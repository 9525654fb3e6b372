"""The N>1 path on CPU: utterance partition + final gather over torch.distributed with the gloo backend,
world_size 2 (the same code runs over RCCL on the GPUs, see bench.py / world_class_amd/shard.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from world_class_amd.shard import ShardLayout, gather_ragged, partition, scatter_back


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
    # the C-ABI twin for hosts without Python (include/world_class_shard.h) deals identically
    from world_class_amd.shard import partition_c
    for world in (1, 2, 3, 8):
        assert partition_c(lengths, world) == partition(lengths, world)
    assert partition_c([7, 7, 7, 7, 7], 2) == partition([7, 7, 7, 7, 7], 2)


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


# ---- the layout the GPUs shard: real batch descriptors (frames and output samples per utterance from the library's own
# size arithmetic), per-frame rows and waveforms gathered back into utterance order ----------------------------------------
FS, HOP = 16000, 5.0


def _fake_outputs(i, f_len, y_len, width):
    """stand-in for what the device pipeline leaves in HBM for utterance i: an F0 contour, `width`-wide rows, a waveform"""
    rng = np.random.default_rng(500 + i)
    return rng.uniform(70, 400, f_len), rng.uniform(0, 1, (f_len, width)), rng.normal(size=y_len)


def _layout_worker(rank, world, port, x_lengths, width, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from world_class_amd.shard import max_over_ranks, sum_over_ranks, verify_group
    verify_group(world, rank, torch.device("cpu"))          # the checks bench.py makes on its RCCL group
    try:
        verify_group(world, 0, torch.device("cpu"))         # every rank claiming device 0: must be refused
        raise AssertionError("duplicate local ids went unnoticed")
    except RuntimeError:
        pass
    assert max_over_ranks([float(rank), 1.5], torch.device("cpu")) == [float(world - 1), 1.5]
    lay = ShardLayout(x_lengths, FS, HOP, world, rank)
    assert sum_over_ranks(sum(lay.f_len), torch.device("cpu")) == sum(lay.all_f_len)
    outs = [_fake_outputs(i, f, y, width) for i, f, y in zip(lay.mine, lay.f_len, lay.y_len)]
    cat = lambda k, shape: torch.from_numpy(np.concatenate([o[k].ravel() for o in outs])) if outs else torch.zeros(shape, dtype=torch.float64)
    f0_all = lay.gather_frames(cat(0, 0))
    rows_all = lay.gather_frames(cat(1, 0), width=width)
    y_all = lay.gather_samples(cat(2, 0))
    # gather to ONE rank: ragged sends to the root, which alone holds the total; root 1 as well as root 0
    for root in (0, world - 1):
        f0_root = lay.gather_frames_to_root(cat(0, 0), root=root)
        rows_root = lay.gather_frames_to_root(cat(1, 0), width=width, root=root)
        y_root = lay.gather_samples_to_root(cat(2, 0), root=root)
        if rank == root:
            assert all(torch.equal(a, b) for a, b in zip(f0_root, f0_all)) and all(torch.equal(a, b) for a, b in zip(y_root, y_all))
            assert all(torch.equal(a, b) for a, b in zip(rows_root, rows_all))
        else:
            assert f0_root is None and rows_root is None and y_root is None
    if rank == 0:
        ret.put(([t.numpy().copy() for t in f0_all], [t.numpy().copy() for t in rows_all], [t.numpy().copy() for t in y_all],
                 lay.parts, lay.all_f_len, lay.all_y_len))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_layout_and_final_gather_gloo_world2():
    import world_class_amd as w
    x_lengths = [16000, 48001, 3200, 80000, 24000, 801, 31999]  # ragged: 0.05 s to 5 s
    width = 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_layout_worker, args=(r, 2, port, x_lengths, width, q)) for r in range(2)]
    for p in procs:
        p.start()
    f0_all, rows_all, y_all, parts, all_f, all_y = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the layout is the library's own size arithmetic, and every utterance went to exactly one rank, longest first
    assert all_f == [w.get_samples(FS, n, HOP) for n in x_lengths]
    assert all_y == [w.synthesis_out_length(f, HOP, FS) for f in all_f]
    assert sorted(i for p in parts for i in p) == list(range(len(x_lengths)))
    loads = [sum(x_lengths[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= max(x_lengths)
    for i, (f, y) in enumerate(zip(all_f, all_y)):
        want = _fake_outputs(i, f, y, width)
        assert np.array_equal(f0_all[i], want[0])
        assert np.array_equal(rows_all[i].reshape(f, width), want[1])
        assert np.array_equal(y_all[i], want[2])


def test_bench_refuses_a_gpu_count_the_box_does_not_have():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing to report a 64-GPU number" in r.stdout
    # and a rank count that contradicts --gpus is an error, not a silent 1-GPU run
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stdout
