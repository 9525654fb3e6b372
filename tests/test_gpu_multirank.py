"""-m gpu: the real pipeline on more than one rank / more than one host thread (SURVEY.md section 8(e), 8(b) "Threading").

* two PROCESSES on GPU 0 under a gloo group: `ShardLayout` deals a ragged 7-utterance batch, each rank runs the fused pipeline
  (wc_pipeline_run_device) on its shard, the results come together through `gather_frames` / `gather_*_to_root`, and equal the
  one-rank `run_batch` of the whole batch (a one-GPU box cannot form a two-rank RCCL group -- RCCL refuses two ranks on one
  device -- so the collective runs over gloo on host copies; the sharding, the packed layouts and the per-rank pipeline are
  the ones bench.py uses);
* bench.py itself with two ranks sharing the device (`--share-gpu`, gloo) and `--gather y`: BASELINE config 4's gather path runs
  with world > 1;
* four host THREADS, each with its own stage objects and its own stream, looping compute on different utterances with explicit
  noise positions: the same bits as the serial run; two threads on ONE object are serialised by the device's call lock and
  give the serial result too (the reference allows distinct objects on distinct threads: per-object scratch, reference
  src/harvest.cpp:69-103; only randn is shared, src/world_matlabfunctions.cpp:243-264);
* a C++ host (tests/cpp/threads.cpp): two threads x wc_set_device, wc_shard_partition, a pipeline handle and a stream per thread.
"""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from world_class_amd.synth import make_utterance

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS, HOP = 16000, 5.0
SECONDS = (0.8, 0.35, 1.2, 0.5, 0.27, 1.0, 0.61)  # ragged


def _batch():
    return [make_utterance(FS, sec, 7100 + i) for i, sec in enumerate(SECONDS)]


def _rank(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import world_class_amd as w
    from world_class_amd.shard import ShardLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w.lib().wc_set_device(0)  # both ranks on the one device of the box
    xs_all = _batch()
    lay = ShardLayout([len(x) for x in xs_all], FS, HOP, world, rank)
    pipe = w.Pipeline(FS, frame_period=HOP)
    xs = [xs_all[i] for i in lay.mine]
    d_x = w.DeviceArray.from_host(np.concatenate(xs))
    nf, ny = sum(lay.f_len), sum(lay.y_len)
    d_t, d_f, d_y = w.DeviceArray(nf), w.DeviceArray(nf), w.DeviceArray(ny)
    d_sp, d_ap = w.DeviceArray(nf * pipe.bins), w.DeviceArray(nf * pipe.bins)
    pipe.run_device(d_x, lay.x_len, d_t, d_f, d_sp, d_ap, d_y)
    host = lambda d: torch.from_numpy(d.to_host())
    f0_all = lay.gather_frames(host(d_f))                        # every rank gets the contours (bench.py's all-gather)
    t_root = lay.gather_frames_to_root(host(d_t), root=0)
    sp_root = lay.gather_frames_to_root(host(d_sp), width=pipe.bins, root=0)
    ap_root = lay.gather_frames_to_root(host(d_ap), width=pipe.bins, root=0)
    y_root = lay.gather_samples_to_root(host(d_y), root=0)       # BASELINE config 4's gather
    assert len(f0_all) == len(xs_all)
    if rank == 0:
        ret.put(dict(parts=lay.parts, bins=pipe.bins, f0=[t.numpy().copy() for t in f0_all], tpos=[t.numpy().copy() for t in t_root],
                     sp=[t.numpy().copy() for t in sp_root], ap=[t.numpy().copy() for t in ap_root], y=[t.numpy().copy() for t in y_root]))
    else:
        assert t_root is None and sp_root is None and y_root is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_one_rank_run():
    import torch.multiprocessing as mp
    import world_class_amd as w
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert sorted(i for p in got["parts"] for i in p) == list(range(len(SECONDS))) and all(len(p) >= 3 for p in got["parts"])
    w.lib().wc_set_device(0)
    one = w.Pipeline(FS, frame_period=HOP).run_batch(_batch())
    for u, r in enumerate(one):
        assert np.array_equal(got["f0"][u], r["f0"]) and np.array_equal(got["tpos"][u], r["tpos"])
        assert np.array_equal(got["sp"][u].reshape(-1, got["bins"]), r["sp"])
        assert np.array_equal(got["ap"][u].reshape(-1, got["bins"]), r["ap"])
        assert np.abs(got["y"][u] - r["y"]).max() < 1e-12


def test_bench_rank_path_with_two_ranks_sharing_the_gpu_and_gather_y():
    """bench.py's own rank code (ShardLayout, step, final gather with --gather y, max over ranks) with world 2 on one device"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--utts", "4", "--distinct", "2", "--seconds", "1.0",
           "--steps", "2", "--warmup", "1", "--gather", "y", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks"] == 2 and out["n_gpus"] == 1 and out["gather"] == "y" and out["gather_ms"] > 0
    assert out["config"]["utterances_per_gpu"] == 4 and out["value"] > 0
    assert "sharing one device" in out["config"]["parallelism"]


def _analysis_synthesis(w, objs, x, start):
    hv, ct, d4, sy = objs
    (t, f), = hv.compute_batch([x])
    sps, p1 = ct.compute_batch([x], [t], [f], rng_pos=[start])
    aps, p2 = d4.compute_batch([x], [t], [f], ct.fft_size, rng_pos=p1)
    ys, p3 = sy.compute_batch([f], sps, aps, rng_pos=p2)
    return f, sps[0], aps[0], ys[0], p3[0]


def test_four_host_threads_with_their_own_objects_and_streams_equal_the_serial_run():
    import torch
    import world_class_amd as w
    L = w.lib()
    L.wc_set_device(0)
    xs = [make_utterance(FS, 0.4 + 0.1 * i, 7300 + i) for i in range(4)]
    starts = [0, 999, 123456, 31]
    make = lambda: (w.Harvest(FS), w.CheapTrick(FS), w.D4C(FS), w.Synthesis(FS, w.cheaptrick_fft_size(FS, 71.0), HOP))
    serial = [_analysis_synthesis(w, make(), x, s0) for x, s0 in zip(xs, starts)]
    errors, rounds = [], [0] * 4

    def worker(k):
        try:
            L.wc_set_device(0)
            stream = torch.cuda.Stream()
            L.wc_set_stream(stream.cuda_stream)  # this thread's calls enqueue on its own stream
            objs = make()
            import time
            t_end = time.perf_counter() + 2.0
            while time.perf_counter() < t_end or rounds[k] < 3:
                got = _analysis_synthesis(w, objs, xs[k], starts[k])
                for a, b in zip(got, serial[k]):
                    assert np.array_equal(a, b), f"thread {k}, round {rounds[k]}: result differs from the serial run"
                rounds[k] += 1
            L.wc_set_stream(None)
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {k}: {e!r}")

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r >= 3 for r in rounds)

    # two threads on ONE set of objects: serialised behind the device's call lock, never corrupted
    shared = make()
    errors2 = []

    def hammer(k):
        try:
            L.wc_set_device(0)
            for _ in range(6):
                got = _analysis_synthesis(w, shared, xs[0], starts[0])
                for a, b in zip(got, serial[0]):
                    assert np.array_equal(a, b)
        except Exception as e:  # noqa: BLE001
            errors2.append(f"thread {k}: {e!r}")
    ths = [threading.Thread(target=hammer, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not errors2, errors2


def test_cpp_host_with_two_threads_shards_and_runs_the_pipeline(tmp_path):
    """tests/cpp/threads.cpp: two threads x wc_set_device(0), wc_shard_partition over a ragged batch, a pipeline handle and a HIP
    stream per thread (wc_set_stream), wc_pipeline_run_device on the thread's shard; the shards put together equal one
    wc_pipeline_run_device over the whole batch (the program compares and exits non-zero otherwise)."""
    from world_class_amd import build
    lib = build.build()
    exe = tmp_path / "threads"
    subprocess.run(["g++", "-std=c++14", "-O1", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "threads.cpp"), "-o", str(exe), "-L" + os.path.dirname(lib), "-lworldclass_hip",
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    inp = tmp_path / "x.f64"
    xs = _batch()
    np.concatenate(xs).tofile(inp)
    r = subprocess.run([str(exe), str(inp), str(FS)] + [str(len(x)) for x in xs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "threads ok" in r.stdout
