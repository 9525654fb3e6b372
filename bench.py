#!/usr/bin/env python
"""Headline benchmark: analysis+synthesis frames/sec, 48 kHz, 5 ms hop (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W [--gather f0|y]

One "step" = one pass of the whole hot path Harvest -> CheapTrick -> D4C -> Synthesis (demo order of
reference test/test.cpp:288-384) over this rank's batch of synthetic 48 kHz 10 s utterances, inputs
already resident in HBM, every stage on the device through the C-ABI (libworldclass_hip.so).

Multi-GPU: one process per GPU over RCCL.  Started under torch.distributed.run the ranks come from the
environment; started plainly with --gpus N > 1 the script re-executes itself under torch.distributed.run
with N ranks (and fails loudly when the box has fewer than N devices).  The global list of N x --utts
utterances is dealt to the ranks by world_class_amd.shard.partition; utterances are independent, so there is
no data-path collective (weak scaling), and the only collective is the final RCCL all-gather of the F0
contours and output checksums (--gather y: of the waveforms too, BASELINE config 4's gather), timed inside
the measured region and reported separately as `gather_ms`.  `value` = frames of all ranks / max-over-ranks
wall time.

At N = 1 rank 0 adds, outside the timed region:
  stages          the other BASELINE configs on this GPU: config 3 (256 x 48 kHz 10 s, CheapTrick only, against the HBM
                  roofline on B_ct), config 2 (64 x 16 kHz 10 s, full pipeline), config 4 (one GPU's share of 1024
                  utterances, Synthesis only), config 5 (one GPU's share of 4096 streams, 1 ms hop, chunked Harvest +
                  CheapTrick: push time and algorithmic latency, whole windows and incremental), and
                  dropin_single_utterance: what an unchanged caller of the four classes sees (host pointers, one
                  48 kHz 10 s utterance, constructor / compute() split like reference test/test.cpp:76-264)
  with_transfers  the same batch through the host front-end (wc_pipeline_run_batch_host): H2D of x and D2H of
                  the outputs inside the clock; reported as `value_with_transfers`, never
                  as `value`
  cpu_baseline    the real reference's OpenMP build (oracle/_ref, kind "reference") on a bounded sample of the
                  same workload, as many concurrent processes as the host's cores allow; our CPU restatement
                  (kind "port") when the reference build is absent
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 48000
SECONDS = 10.0
FRAME_PERIOD = 5.0
# algorithmic HBM bytes per 5 ms frame at 48 kHz (restated in DESIGN.md)
STAGE_BYTES = {
    "harvest": 1920 + 16,          # hop samples in, (tpos, f0) out
    "cheaptrick": 8 * 2048 + 8 * 1025,  # "batched-FFT-stage" view B_ct: windowed frame in, envelope out
    "d4c": 1920 + 16 + 8200,       # hop samples + (tpos, f0) in, aperiodicity row out
    "synthesis": 8 + 8200 + 8200 + 1920,
}
KERNEL_STAGE = {
    "harvest_decimate": "harvest", "harvest_bandpass": "harvest", "harvest_raw": "harvest",
    "harvest_refine": "harvest", "harvest_contour": "harvest",
    "cheaptrick_frames": "cheaptrick", "d4c_lovetrain": "d4c", "d4c_frames": "d4c", "d4c_bands": "d4c",
    "synthesis_timebase": "synthesis", "synthesis_pulses": "synthesis",
}
SEQUENTIAL_SCANS = ("synthesis_timebase", "harvest_contour")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def _ref_call(args):
    """one fresh reference process over a list of utterances, one after the other; returns (frames, start stamp, end stamp), the
    stamps taken inside the process (oracle/ref.py: pipeline_timed), so interpreter start-up and pickling stay outside"""
    from oracle import ref
    xs, threads = args
    os.environ["OMP_NUM_THREADS"] = str(threads)
    return ref.run_fresh("pipeline_timed", xs, FS, harvest_floor=71.0, omp=True)


def usable_cores():
    """cores this process may actually run on: the affinity mask, cut down by a cgroup CPU quota if there is one (a container on
    a 256-core host often owns far fewer; os.cpu_count() reports the host's)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(xs, budget_s=24.0, min_utts=16, reps=3):
    """CPU path on a bounded sample (whole utterances of the same workload), rank 0 / N=1 only.

    The reference's OpenMP build does not scale to all cores of a large host inside one process (its parallel
    loops allocate and plan FFTs per iteration), but utterances are as independent on the CPU as on the GPU: the
    thread count T with the best rate per thread is found first (one utterance each, T in {2, 4, 8, 16}), then
    floor(cores / T) processes run concurrently, each over its share of at least `min_utts` utterances one after the
    other, and THAT whole-host rate is the baseline: frames of all processes / (last end - first start), the stamps
    taken inside the processes.  Repeated `reps` times; `value` is the median, the spread is reported.  Every run is a
    fresh process: the reference's noise state is process-global, and its Synthesis overflows its pulse arrays on
    some inputs, which a subprocess isolates.
    """
    from concurrent.futures import ThreadPoolExecutor
    from oracle import port, ref
    cores = usable_cores()

    if ref.available(omp=True):
        try:
            best, trials = None, {}
            for th in sorted({min(cores, t) for t in (2, 4, 8, 16)}):
                n, t0, t1 = _ref_call(([xs[0]], th))
                rate = n / max(t1 - t0, 1e-3)
                trials[th] = rate
                if best is None or rate / th > 1.02 * best[1] / best[0]:  # (the larger count must earn its threads)
                    best = (th, rate)
            threads, one_rate = best
            procs = max(1, min(64, cores // threads))
            per = max(1, -(-min_utts // procs))
            jobs = [([xs[(i * per + k) % len(xs)] for k in range(per)], threads) for i in range(procs)]
            rates = []
            with ThreadPoolExecutor(procs) as ex:  # (one discarded pass of an utterance per process: page cache, core clocks)
                list(ex.map(_ref_call, [([xs[i % len(xs)]], threads) for i in range(procs)]))
            for _ in range(reps):
                with ThreadPoolExecutor(procs) as ex:
                    res = list(ex.map(_ref_call, jobs))
                wall = max(max(r[2] for r in res) - min(r[1] for r in res), 1e-3)
                rates.append(sum(r[0] for r in res) / wall)
            value = float(np.median(rates))
            return {"value": value, "unit": "frames/s", "cores": procs * threads, "kind": "reference",
                    "sample": f"{procs * per} x 48 kHz 10 s utterances of the same synthetic workload, full pipeline, OpenMP build of the "
                              f"reference (oracle/_ref): {procs} concurrent processes x {threads} threads x {per} utterances each on a host "
                              f"with {cores} cores; median of {reps} repetitions",
                    "repetitions": rates, "spread": (max(rates) - min(rates)) / value,
                    "one_process": {"value": one_rate, "threads": threads, "by_threads": trials}, "host_cores": cores}
        except Exception as e:  # fall through to the restatement
            sys.stderr.write(f"cpu_baseline: reference run failed ({e}); using the restatement\n")
    P = port.Port()
    P.set_threads(cores)
    frames, t_used, n_done = 0, 0.0, 0
    for x in xs:
        if t_used > budget_s * 0.5 and n_done >= 1:
            break
        t0 = time.perf_counter()
        r = P.pipeline(x, FS)
        t_used += time.perf_counter() - t0
        frames += len(r["f0"])
        n_done += 1
    P.set_threads(0)
    return {"value": frames / t_used, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_done} x 48 kHz 10 s utterance(s) of the same synthetic workload, full pipeline, CPU restatement "
                      f"(oracle/), OpenMP, {cores} threads"}


def spawn_ranks(a, argv):
    """plain `python bench.py --gpus N`, N > 1: re-execute under torch.distributed.run with one rank per GPU"""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: this box has {have} HIP device(s); refusing to report a {a.gpus}-GPU number")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def stage_cheaptrick(w, L, torch, dev, d_x64, x_len64, d_t64, d_f64, f_len64, reps=4, iters=5):
    """BASELINE config 3: 256 x 48 kHz x 10 s, CheapTrick only (contour = this batch's own Harvest output), resident in HBM"""
    ct = w.CheapTrick(FS)
    d_x = d_x64.repeat(reps)
    d_t, d_f = d_t64.repeat(reps), d_f64.repeat(reps)
    xl, fl = list(x_len64) * reps, list(f_len64) * reps
    frames = sum(fl)
    d_sp = torch.empty(frames * ct.bins, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ct.compute_device(d_x, xl, d_t, d_f, fl, d_sp)
    L.wc_synchronize()
    L.wc_set_kernel_timing(1)
    wall, kern = [], []
    for _ in range(iters):
        t0 = time.perf_counter()
        ct.compute_device(d_x, xl, d_t, d_f, fl, d_sp)
        L.wc_synchronize()
        wall.append(time.perf_counter() - t0)
        kern.append(float(L.wc_last_kernel_ms(b"cheaptrick_frames")) * 1e-3)
    L.wc_set_kernel_timing(0)
    t, k = float(np.median(wall)), float(np.mean(kern))
    b_ct = STAGE_BYTES["cheaptrick"]
    return {"workload": f"{len(xl)} x 48 kHz 10 s, CheapTrick only (2048-point FFT), contour given, resident in HBM (BASELINE config 3)",
            "frames": frames, "ms": t * 1e3, "frames_per_s": frames / t, "kernel": "ct_wave_kernel",
            "kernel_ms": k * 1e3, "bytes_per_frame": b_ct, "achieved_GBps": frames * b_ct / k / 1e9,
            "hbm_frac": frames * b_ct / k / (HBM_PEAK_GBS * 1e9), "_k": k}


def _timed(fn, L, iters=3):
    fn()
    L.wc_synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        L.wc_synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def stage_config2(w, L, torch, dev):
    """BASELINE config 2: 64 x 16 kHz x 10 s, 5 ms hop, full pipeline on one GPU, resident in HBM"""
    from world_class_amd.synth import make_utterance
    fs, n = 16000, 64
    base = [make_utterance(fs, SECONDS, 2000 + u) for u in range(8)]
    xs = [base[i % 8] for i in range(n)]
    p = w.Pipeline(fs)
    xl = [len(x) for x in xs]
    fl, yl = p.lengths(xl)
    d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
    d_t = torch.empty(sum(fl), dtype=torch.float64, device=dev)
    d_f = torch.empty_like(d_t)
    d_sp = torch.empty(sum(fl) * p.bins, dtype=torch.float64, device=dev)
    d_ap = torch.empty_like(d_sp)
    d_y = torch.empty(sum(yl), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t = _timed(lambda: p.run_device(d_x, xl, d_t, d_f, d_sp, d_ap, d_y), L)
    return {"workload": f"{n} x 16 kHz 10 s, 5 ms hop, full pipeline, resident in HBM (BASELINE config 2)", "frames": sum(fl),
            "ms": t * 1e3, "frames_per_s": sum(fl) / t}


def stage_config4(w, L, torch, dev, pipe):
    """BASELINE config 4, one GPU's share: Synthesis only from precomputed {f0, sp, ap} of 128 x 48 kHz x 10 s (1024 / 8)"""
    from world_class_amd.synth import make_utterance
    n = 128
    base = pipe.run_batch([make_utterance(FS, SECONDS, 4000 + u) for u in range(4)])
    sy = w.Synthesis(FS, pipe.fft_size, FRAME_PERIOD)
    fl = [len(base[i % 4]["f0"]) for i in range(n)]
    yl = [sy.out_length(v) for v in fl]
    d_f = torch.from_numpy(np.concatenate([base[i % 4]["f0"] for i in range(n)])).to(dev)
    sp4 = [torch.from_numpy(b["sp"].ravel()).to(dev) for b in base]
    ap4 = [torch.from_numpy(b["ap"].ravel()).to(dev) for b in base]
    d_sp = torch.cat([sp4[i % 4] for i in range(n)])
    d_ap = torch.cat([ap4[i % 4] for i in range(n)])
    d_y = torch.empty(sum(yl), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t = _timed(lambda: sy.compute_device(d_f, fl, d_sp, d_ap, yl, d_y), L)
    return {"workload": f"{n} x 48 kHz 10 s (one GPU's share of 1024), Synthesis only from {(d_sp.numel() + d_ap.numel()) * 8 / 1e9:.1f} GB of "
                        "resident f0 / sp / ap (BASELINE config 4)", "frames": sum(fl), "ms": t * 1e3, "frames_per_s": sum(fl) / t,
            "bytes_per_frame": STAGE_BYTES["synthesis"], "hbm_frac": sum(fl) * STAGE_BYTES["synthesis"] / t / (HBM_PEAK_GBS * 1e9)}


def stage_config5(w, L, torch, dev):
    """BASELINE config 5, one GPU's share: 512 concurrent 24 kHz streams, 1 ms frames, chunked Harvest + CheapTrick
    (include/world_class_stream.h), 200 ms chunks: whole windows and the incremental mode"""
    from world_class_amd.stream import StreamAnalyzer
    from world_class_amd.synth import make_utterance
    fs, n = 24000, 512
    sig = [make_utterance(fs, 4.0, 5000 + u) for u in range(8)]
    out = {"workload": f"{n} concurrent 24 kHz streams (one GPU's share of 4096), 1 ms frames, chunked Harvest + CheapTrick, 200 ms per push "
                       "(BASELINE config 5; the reference has no streaming mode: semantics in DESIGN.md section 10)"}
    for key, chunk_ms, back_ms, ahead_ms, ctx_ms in (("whole_windows", 200, 400, 400, 0), ("incremental", 200, 400, 560, 160)):
        sa = StreamAnalyzer(fs, n, frame_period=1.0, chunk_ms=chunk_ms, lookback_ms=back_ms, lookahead_ms=ahead_ms, context_ms=ctx_ms)
        cs = sa.chunk_samples
        cap = n * sa.max_frames
        d_t = torch.empty(cap, dtype=torch.float64, device=dev)
        d_f = torch.empty(cap, dtype=torch.float64, device=dev)
        d_sp = torch.empty(cap * sa.bins, dtype=torch.float64, device=dev)
        n_push = len(sig[0]) // cs
        chunks = [torch.from_numpy(np.concatenate([sig[u % 8][k * cs:(k + 1) * cs] for u in range(n)])).to(dev) for k in range(n_push)]
        torch.cuda.synchronize()
        times, frames = [], []
        for k in range(n_push):
            t0 = time.perf_counter()
            counts = sa.push_device(chunks[k], None, None, d_t, d_f, d_sp)
            L.wc_synchronize()
            times.append(time.perf_counter() - t0)
            frames.append(sum(counts))
        full = (back_ms + chunk_ms + ahead_ms) // chunk_ms + 1  # pushes until the history window is full
        t = float(np.median(times[full:]))
        out[key] = {"lookback_ms": back_ms, "lookahead_ms": ahead_ms, "context_ms": ctx_ms, "frames_per_push": frames[-1], "push_ms": t * 1e3,
                    "frames_per_s": frames[-1] / t, "algorithmic_latency_ms": ahead_ms + chunk_ms,
                    "latency_ms_incl_compute": ahead_ms + chunk_ms + t * 1e3, "real_time_factor": chunk_ms / (t * 1e3)}
        del sa, chunks, d_t, d_f, d_sp
    return out


def stage_dropin(w, L, x):
    """What an unchanged caller of the reference's four classes sees: host pointers, one 48 kHz 10 s utterance, constructor and
    compute() timed apart like the demo (reference test/test.cpp:76-264: Harvest floor 40 Hz, CheapTrick floor 71 Hz, D4C
    threshold 0.85).  `first`: fresh objects and fresh result arrays (workspaces and the noise table are created, the arrays'
    pages are touched for the first time); `steady`: the same objects and arrays again."""
    bufs = {}  # the caller's result arrays, allocated once like the demo's (fresh 16 MB numpy arrays are 4096 page faults per call)

    def once(objs):
        ms = {}
        t0 = time.perf_counter(); hv = objs.get("hv") or w.Harvest(FS, f0_floor=40.0, frame_period=FRAME_PERIOD); t1 = time.perf_counter()
        tpos, f0 = hv.compute(x); t2 = time.perf_counter()
        ms["harvest"] = {"ctor_ms": (t1 - t0) * 1e3, "compute_ms": (t2 - t1) * 1e3}
        t0 = time.perf_counter(); ct = objs.get("ct") or w.CheapTrick(FS); t1 = time.perf_counter()
        sp = ct.compute(x, tpos, f0, out=bufs.get("sp")); t2 = time.perf_counter()
        ms["cheaptrick"] = {"ctor_ms": (t1 - t0) * 1e3, "compute_ms": (t2 - t1) * 1e3}
        t0 = time.perf_counter(); d4 = objs.get("d4") or w.D4C(FS); t1 = time.perf_counter()
        ap = d4.compute(x, tpos, f0, ct.fft_size, out=bufs.get("ap")); t2 = time.perf_counter()
        ms["d4c"] = {"ctor_ms": (t1 - t0) * 1e3, "compute_ms": (t2 - t1) * 1e3}
        t0 = time.perf_counter(); sy = objs.get("sy") or w.Synthesis(FS, ct.fft_size, FRAME_PERIOD); t1 = time.perf_counter()
        y = sy.compute(f0, sp, ap, out=bufs.get("y")); t2 = time.perf_counter()
        bufs.update(sp=sp, ap=ap, y=y)
        ms["synthesis"] = {"ctor_ms": (t1 - t0) * 1e3, "compute_ms": (t2 - t1) * 1e3}
        ms["total_compute_ms"] = sum(v["compute_ms"] for v in ms.values())
        return ms, dict(hv=hv, ct=ct, d4=d4, sy=sy), len(f0), len(y)
    w.rng_set_position(0)
    first, objs, frames, _ = once({})
    runs = []
    for _ in range(3):
        w.rng_set_position(0)
        runs.append(once(objs)[0])
    steady = min(runs, key=lambda r: r["total_compute_ms"])
    return {"workload": "one 48 kHz 10 s utterance through Harvest::compute ... Synthesis::compute with host pointers (the reference demo's calls, "
                        "reference test/test.cpp:76-264; H2D / D2H of every stage's arguments included)", "frames": frames,
            "first": first, "steady": steady, "frames_per_s_steady": frames / (steady["total_compute_ms"] * 1e-3)}


def with_transfers(w, pipe, xs, frames, iters=3):
    """the host front-end on the same batch: H2D of x and D2H of the outputs inside the clock"""
    out = {}
    pcm = [np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16) for x in xs[:8]]
    pcm = [pcm[i % len(pcm)] for i in range(len(xs))]
    xl = [len(x) for x in xs]
    xs_pinned = pipe.host_inputs(xs)
    for key, inp, want, ypcm, label in (
            ("f64_in_all_five_out", xs, ("tpos", "f0", "sp", "ap", "y"), False,
             "x as float64 from the caller's page-locked host memory, tpos + f0 + spectrogram + aperiodicity + waveform back as float64 into "
             "the caller's page-locked rows (section 8(d) to the letter)"),
            ("pcm16_in_f0_pcm16_out", pcm, ("f0", "y"), True,
             "x as the int16 PCM of a WAV file, F0 + int16 waveform back; spectrogram and aperiodicity stay in HBM")):
        res = pipe.host_buffers(xl, want=want, y_pcm16=ypcm, pinned=True)  # the caller's result buffers, page-locked, reused
        if inp is xs:
            inp = xs_pinned  # ... and its float64 utterances in page-locked memory as well (read by the copy engine where they lie)
        pipe.run_batch_host(inp, want=want, y_pcm16=ypcm, out=res)  # warm-up: pinned staging, device buffers
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            pipe.run_batch_host(inp, want=want, y_pcm16=ypcm, out=res)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        out[key] = {"what": label, "ms": t * 1e3, "frames_per_s": frames / t,
                    "host_bytes_in": int(sum(v.nbytes for v in inp)),
                    "host_bytes_out": int(sum(a.nbytes for r in res for a in r.values()))}
        del res
    # the feature codec as the epilogue: F0 + 60 mel-cepstral coefficients + band aperiodicities + waveform back
    res = pipe.coded_host_buffers(xl, number_of_dimensions=60, pinned=True)
    pipe.run_batch_host_coded(xs_pinned, number_of_dimensions=60, out=res)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        pipe.run_batch_host_coded(xs_pinned, number_of_dimensions=60, out=res)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    out["coded_out"] = {"what": "x as float64 in; F0, 60 mel-cepstral coefficients and the band aperiodicities per frame (the reference's codec, "
                                "src/codec.cpp:211-325, as the epilogue of CheapTrick / D4C) and the float64 waveform back",
                        "ms": t * 1e3, "frames_per_s": frames / t, "host_bytes_in": int(sum(v.nbytes for v in xs_pinned)),
                        "host_bytes_out": int(sum(a.nbytes for r in res for a in r.values()))}
    return out


# wavefronts per SIMD the full-grid kernels of the 48 kHz step run at (registers / LDS: profiles/r05_*_kernel_resources.txt)
WAVES_PER_SIMD = {"harvest_bandpass": 3, "harvest_raw": 7, "harvest_refine": 4, "cheaptrick_frames": 2, "d4c_lovetrain": 2, "d4c_frames": 2,
                  "d4c_bands": 2, "synthesis_pulses": 2}


def tie_counts(L, reset=False):
    """(utterances that went through a Harvest refinement, utterances flagged for a tie) since the last reset -- the library's
    process-wide counters (wc_harvest_tie_counts, a development hook)"""
    import ctypes as C
    fn = L.wc_harvest_tie_counts
    fn.restype = None
    fn.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
    a, b = C.c_ulonglong(0), C.c_ulonglong(0)
    fn(C.byref(a), C.byref(b), 1 if reset else 0)
    return int(a.value), int(b.value)


def per_1000(seen, flagged):
    return {"utterances": seen, "flagged": flagged, "per_1000_utterances": (1000.0 * flagged / seen) if seen else None}


def rate_at(table, waves):
    """nanoseconds per instruction at `waves` wavefronts per SIMD: the measured occupancy, or the next one measured above it (the
    tables are non-increasing in the occupancy, so that is a lower bound on the cost)"""
    if waves in table:
        return table[waves]
    above = [k for k in table if k > waves]
    return table[min(above)] if above else table[max(table)]


def load_issue_rates():
    """profiles/issue_rates.json (tools/issue_rate.hip on the GPU box) -> ns per wave-instruction and SIMD by wavefronts per SIMD:
    'fp64' = the v_fma_f64 stream, 'int32' = the cheaper of the v_add_u32 / v_mov_b32 streams, made non-increasing in the occupancy
    (a lower bound of what the non-FP64 instructions of a kernel cost)"""
    path = os.path.join(ROOT, "profiles", "issue_rates.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    try:
        ws = (1, 2, 3, 4, 8)
        f64 = {wv: d["v_fma_f64"]["ns_per_inst_by_waves"][str(wv)] for wv in ws}
        i32, best = {}, float("inf")
        for wv in ws:
            best = min(best, d["v_add_u32"]["ns_per_inst_by_waves"][str(wv)], d["v_mov_b32"]["ns_per_inst_by_waves"][str(wv)])
            i32[wv] = best
        return {"fp64": f64, "int32": i32}
    except (KeyError, TypeError):
        return None


def serialised_kernels(w, L, d_x, x_len, d_t, d_f, f_len, d_sp, d_ap, y_len, d_y, fft_size, iters=3):
    """every kernel of a step alone on the chip: the four stage calls one after the other on the same resident batch (what
    tools/microbench.py does), HIP events on the library's stream around each kernel, the best of `iters`.  The results overwrite
    the step's own (the same values: the noise positions are chained like the pipeline chains them)."""
    n = len(x_len)
    hv, ct, d4 = w.Harvest(FS, frame_period=FRAME_PERIOD), w.CheapTrick(FS), w.D4C(FS)
    sy = w.Synthesis(FS, fft_size, FRAME_PERIOD)
    old = os.environ.get("WC_SYN_HALVES")
    os.environ["WC_SYN_HALVES"] = "0"  # (the stage call would otherwise overlap its second half's time base with the first half's pulses)

    def once():
        hv.compute_device(d_x, x_len, d_t, d_f)
        pos = ct.compute_device(d_x, x_len, d_t, d_f, f_len, d_sp, rng_pos=[0] * n)
        pos = d4.compute_device(d_x, x_len, d_t, d_f, f_len, fft_size, d_ap, rng_pos=pos)
        sy.compute_device(d_f, f_len, d_sp, d_ap, y_len, d_y, rng_pos=pos)
        L.wc_synchronize()
    best = {}
    try:
        once()
        L.wc_set_kernel_timing(1)
        for _ in range(iters):
            once()
            for k in KERNEL_STAGE:
                ms = float(L.wc_last_kernel_ms(k.encode()))
                if ms >= 0:
                    best[k] = min(best.get(k, 1e9), ms)
    finally:
        L.wc_set_kernel_timing(0)
        if old is None:
            del os.environ["WC_SYN_HALVES"]
        else:
            os.environ["WC_SYN_HALVES"] = old
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--utts", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic utterances per GPU (tiled to --utts)")
    ap.add_argument("--gather", choices=("f0", "y"), default="f0", help="what the final RCCL all-gather collects besides the checksums")
    ap.add_argument("--seconds", type=float, default=SECONDS, help="utterance length (the headline workload is 10 s; shorter only in tests)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test mode: all ranks on device 0 under a gloo group (RCCL does not form a group of several ranks on one device); "
                         "exercises the sharded code path on a one-GPU box and reports n_gpus = 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stages / with_transfers blocks (N = 1)")
    ap.add_argument("--no-serialised", action="store_true",
                    help="skip the serialised kernel pass (the counter passes of tools/profile_round.sh: every kernel launch of the process "
                         "must belong to a pipeline step, the counters are divided by the number of steps); roofline.kernel_ms then comes "
                         "from the overlapped step's events")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if a.share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if a.share_gpu else dev  # where the collectives' tensors live (gloo: host memory)
    if world > 1 and a.share_gpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        # RCCL really spans `world` ranks on distinct devices
        from world_class_amd.shard import verify_group
        if dist.get_backend() != "nccl":
            raise SystemExit("bench.py: the process group is not RCCL")
        try:
            verify_group(world, local_rank, dev)
        except RuntimeError as e:
            raise SystemExit(f"bench.py: {e}")

    import world_class_amd as w
    from world_class_amd.shard import ShardLayout
    from world_class_amd.synth import make_utterance
    L = w.lib()
    L.wc_set_device(local_rank)

    # ---- synthetic workload: world x utts utterances of 10 s; utterance i is seed 3000 + i % (distinct x world); the ranks
    # take their shares by the static longest-first partition of world_class_amd.shard (equal lengths: round robin) ----
    n_total = a.utts * world
    cache = {3000: make_utterance(FS, a.seconds, 3000)}
    n_samples = len(cache[3000])  # every utterance of the workload has this length
    lay = ShardLayout([n_samples] * n_total, FS, FRAME_PERIOD, world, rank)
    n_utt = len(lay.mine)
    distinct = max(1, min(a.distinct, n_utt)) * world
    xs = []
    for i in lay.mine:
        seed = 3000 + i % distinct
        if seed not in cache:
            cache[seed] = make_utterance(FS, a.seconds, seed)
        xs.append(cache[seed])
    assert [len(x) for x in xs] == lay.x_len, "utterance length differs from the layout's"
    x_len, f_len, y_len = lay.x_len, lay.f_len, lay.y_len
    frames = sum(f_len)
    pipe = w.Pipeline(FS, frame_period=FRAME_PERIOD)  # Harvest -> CheapTrick -> D4C -> Synthesis, reference defaults
    bins = pipe.bins
    d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
    d_t = torch.empty(frames, dtype=torch.float64, device=dev)
    d_f = torch.empty(frames, dtype=torch.float64, device=dev)
    d_sp = torch.empty(frames * bins, dtype=torch.float64, device=dev)
    d_ap = torch.empty(frames * bins, dtype=torch.float64, device=dev)
    d_y = torch.empty(sum(y_len), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step():
        # one fused call: all four stages, every utterance starting its noise stream at position 0
        pipe.run_device(d_x, x_len, d_t, d_f, d_sp, d_ap, d_y)

    def barrier():
        L.wc_synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    gather_s = [0.0]

    def final_gather():
        # the path's only collective: the F0 contours (and, --gather y, the waveforms) of every utterance in the original
        # utterance order on every rank, plus the output checksums
        L.wc_synchronize()
        summary = torch.stack([d_sp.sum(), d_ap.sum(), d_y.abs().sum()])
        if world > 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # F0 contours: all-gather (every rank gets the 2001-frame contours: 16 KB per utterance); waveforms (--gather y,
            # BASELINE config 4's gather): to rank 0 only, each peer over its own xGMI link (shard.gather_ragged_to_root)
            f0_all = lay.gather_frames(d_f.to(cdev))
            y_all = lay.gather_samples_to_root(d_y.to(cdev), root=0) if a.gather == "y" else None
            summary_c = summary.to(cdev)
            sums = [torch.empty_like(summary_c) for _ in range(world)]
            dist.all_gather(sums, summary_c)
            torch.cuda.synchronize()
            gather_s[0] = time.perf_counter() - t0
            assert len(f0_all) == n_total and (y_all is None or rank != 0 or len(y_all) == n_total)
        return summary

    for _ in range(a.warmup):
        step()
    if a.warmup > 0:
        final_gather()  # one-time costs of the epilogue too (torch reduction kernels, RCCL channel set-up)
    L.wc_set_kernel_timing(1)
    barrier()
    tie_counts(L, reset=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    final_gather()
    barrier()
    elapsed = time.perf_counter() - t0
    ties = {"headline": per_1000(*tie_counts(L, reset=True))}
    if world > 1:
        from world_class_amd.shard import max_over_ranks, sum_over_ranks
        elapsed, gather_s[0] = max_over_ranks([elapsed, gather_s[0]], cdev)
        total_frames = sum_over_ranks(frames, cdev)
    else:
        total_frames = frames

    # What a tie costs (round-5 verdict, item 3): the same step with ONE of its utterances forced onto the tie flag -- that
    # utterance goes through the whole path once more by itself, the band-pass as direct FIR sums; everything else stands
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            pipe.set_option("force_tie", str(n_utt // 3))
            step(); barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                step()
            barrier()
            ties["step_with_one_utterance_flagged_ms"] = (time.perf_counter() - t1) / 5 * 1e3
            ties["plain_step_ms"] = elapsed / a.steps * 1e3
            ties["ratio"] = ties["step_with_one_utterance_flagged_ms"] / ties["plain_step_ms"]
        except Exception as e:
            ties["step_with_one_utterance_flagged_ms"] = f"failed: {e}"
        finally:
            pipe.set_option("force_tie", None)
            tie_counts(L, reset=True)
    # per-kernel time of the last timed step, HIP events on the library's own streams: what the kernels take INSIDE the overlapped
    # schedule (a kernel that shares the chip with the other half batch's is stretched by the sharing; the sum exceeds the step)
    kern_overlapped = {}
    for name in KERNEL_STAGE:
        ms = float(L.wc_last_kernel_ms(name.encode()))
        if ms >= 0:
            kern_overlapped[name] = ms
    L.wc_set_kernel_timing(0)
    # The host-memory measurements come first, in a process that has done nothing else with host memory or large device
    # allocations (like the reference demo): behind the stage calls of the serialised pass below the same front-end run measured
    # 67 instead of 49 ms, behind the config 4 / config 5 stages 10 - 25 ms more, the drop-in caller 9 ms more behind the front-end's
    # 2.6 GB of page-locked buffers -- freeing tens of GB of device memory and pageable uploads leave this ROCm's copy path in a
    # slower state for the rest of the process (cause not found).
    host_first = {}
    if rank == 0 and world == 1 and not a.no_extras:
        try:  # (the unchanged caller's view)
            host_first["stages"] = {"dropin_single_utterance": stage_dropin(w, L, xs[0])}
        except Exception as e:
            host_first["stages"] = {"dropin_single_utterance": {"error": str(e)}}
        try:
            host_first["with_transfers"] = with_transfers(w, pipe, xs, frames)
            host_first["value_with_transfers"] = host_first["with_transfers"]["f64_in_all_five_out"]["frames_per_s"]
        except Exception as e:
            host_first["with_transfers"] = {"error": str(e)}
        try:  # (2.6 GB of page-locked buffers go back to the system: torch keeps them cached otherwise)
            torch._C._host_emptyCache()
        except Exception:
            pass
    # ... and the same kernels one after the other, each alone on the chip, one launch for the whole batch: the times the roofline
    # figures are priced on (no schedule effects, no exclusion list)
    kern = {}
    if rank == 0:
        kern = dict(kern_overlapped) if a.no_serialised else serialised_kernels(w, L, d_x, x_len, d_t, d_f, f_len, d_sp, d_ap, y_len, d_y, pipe.fft_size)

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = total_frames * a.steps / elapsed
        # dominant = the longest kernel of the serialised pass
        dom = max(kern, key=kern.get) if kern else None
        roofline = None
        pmc_all = None
        if dom:
            stage = KERNEL_STAGE[dom]
            achieved = frames * STAGE_BYTES[stage] / (kern[dom] * 1e-3) / 1e9
            traffic, fp64 = None, None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            pmc = {}
            if os.path.exists(tpath):
                with open(tpath) as f:
                    pmc = json.load(f)
                # counters of another build say nothing about this one: the file carries the source hash of the library it was
                # measured on (tools/profile_round.sh), and a stale one is refused
                if pmc.get("_build_hash") != L.wc_build_hash().decode():
                    sys.stderr.write("bench.py: profiles/pmc_traffic.json was measured on another build; traffic / FLOP figures left out\n")
                    pmc = {}
            pmc_all = pmc
            if pmc:
                traffic = pmc.get(dom)  # HBM bytes per step (PMC, see the file's _note)
                flops = pmc.get("_fp64_flops_per_step", {}).get(dom)
                if flops:  # FP64 FLOP/s from SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes (FMA counted twice)
                    rate = flops / (kern[dom] * 1e-3) / 1e12
                    fp64 = {"flops_per_step": flops, "tflops": rate, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS, "frac": rate / FP64_VECTOR_PEAK_TFLOPS}
            # What binds: none of the full-grid kernels is limited by bytes (20 KB per frame against ~6 MFLOP of FP64); they compete
            # for vector issue slots.  issue_frac = the time one SIMD needs to issue the kernel's vector instructions at the rates
            # tools/issue_rate.hip measured on this part for the kernel's occupancy (profiles/issue_rates.json: nanoseconds per
            # wave-instruction and SIMD of an FP64 stream and of a 32-bit integer stream at 1 / 2 / 3 / 4 / 8 wavefronts per SIMD,
            # HIP-event timed, so the clock the chip settles at is inside the figure) / the kernel's serialised time:
            #   (n_fp64 c_fp64(W) + (n_valu - n_fp64) c_int(W)) / (1024 SIMDs x t),
            # n from the counter passes (SQ_INSTS_VALU, SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64).  Round 4 charged every instruction
            # 4 cycles at 2.4 GHz; measured: an FP64 instruction costs 2.2 - 2.8 ns (4.5 - 5.5 cycles at the ~2 GHz an FP64 stream
            # runs at), a 32-bit one 1.05 - 1.6 ns, and two wavefronts per SIMD cannot issue faster than one instruction per
            # ~3.75 cycles whatever the class (one wavefront issues every 7.5 cycles).
            issue, issue_detail = {}, None
            rates = load_issue_rates()
            n_valu_all = pmc.get("_valu_insts_per_step") or {}
            n_f64_all = pmc.get("_fp64_insts_per_step") or {}
            if rates and n_valu_all:
                issue_detail = {}
                for k, n_valu in n_valu_all.items():
                    if k not in kern or kern[k] <= 0 or k not in WAVES_PER_SIMD:
                        continue
                    n64 = sum((n_f64_all.get(k) or {}).values())
                    wv = WAVES_PER_SIMD[k]
                    c64, c32 = rate_at(rates["fp64"], wv), rate_at(rates["int32"], wv)
                    t_issue = (n64 * c64 + max(0.0, n_valu - n64) * c32) * 1e-9 / 1024.0
                    issue[k] = t_issue / (kern[k] * 1e-3)
                    issue_detail[k] = {"waves_per_simd": wv, "valu_insts": n_valu, "fp64_insts": n64, "ns_fp64": c64, "ns_int32": c32,
                                       "issue_ms": t_issue * 1e3, "kernel_ms": kern[k]}
            ser = sum(kern.values())
            schedule = {"serialised_sum_ms": ser, "step_ms": ms_per_step, "gained_by_overlap_ms": ser - ms_per_step,
                        "sequential_scans_ms": {k: kern[k] for k in SEQUENTIAL_SCANS if k in kern},
                        "note": "serialised: every kernel alone on the chip, one launch per batch (stage calls one after the other); the step overlaps "
                                "the two half batches' chains -- the one-wavefront-per-utterance scans (Harvest contour logic, Synthesis time "
                                "base) run underneath the other half's full-grid kernels"}
            roofline = {"bound": "hbm", "binds": "fp64_issue", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "kernel_ms": kern[dom], "bytes_per_frame": STAGE_BYTES[stage],
                        "issue_frac": issue.get(dom), "issue_frac_by_kernel": issue or None, "issue_model": issue_detail,
                        "issue_frac_step": (sum(v["issue_ms"] for v in issue_detail.values()) / ms_per_step) if issue_detail else None,
                        "schedule": schedule,
                        "fp64_vector": fp64, "all_kernels_ms": kern, "kernels_in_the_overlapped_step_ms": kern_overlapped,
                        "pipeline": {"bytes_per_frame": 20248, "achieved": frames * 20248 / (ms_per_step * 1e-3) / 1e9,
                                     "frac": frames * 20248 / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        out = {
            "metric": "analysis+synthesis frames/sec (whole node), 48 kHz 5 ms hop",
            "value": value, "unit": "frames/s", "n_gpus": 1 if a.share_gpu else world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n_utt} synthetic 48 kHz {a.seconds:g} s utterances per GPU ({distinct // world} distinct, tiled), 5 ms hop, "
                                   "Harvest->CheapTrick->D4C->Synthesis, inputs resident in HBM when the clock starts (`value`; SURVEY.md section 8(d)'s "
                                   "headline with H2D of x and D2H of all outputs inside the clock is `value_with_transfers`)",
                       "utterances_per_gpu": n_utt, "frames_per_gpu": frames, "fs": FS, "frame_period_ms": FRAME_PERIOD,
                       "fft_size": pipe.fft_size,
                       "parallelism": (f"TEST MODE: {world} ranks sharing one device under gloo (no scaling number), utterance-sharded (shard.partition), final gather of "
                                       if a.share_gpu else f"utterance-sharded x{world} (shard.partition), final RCCL all-gather of ")
                                      + ("F0 + checksums, waveforms gathered to rank 0" if a.gather == "y" else "F0 + checksums")},
            "roofline": roofline,
        }
        if world > 1:
            out["gather_ms"] = gather_s[0] * 1e3
            out["gather"] = a.gather
            out["ranks"] = world
        if world == 1 and not a.no_extras:
            # (the host-memory measurements first: behind the config 4 / config 5 stages below the host front-end measured 10 - 25 ms
            # more in this process, and the drop-in caller 9 ms more behind the front-end's 2.6 GB of page-locked buffers -- large
            # host allocations and pageable uploads leave this ROCm's copy path in a slower state; cause not found)
            st = dict(host_first.get("stages", {}))
            for k in ("with_transfers", "value_with_transfers"):
                if k in host_first:
                    out[k] = host_first[k]
            for key, fn in (("cheaptrick_config3", lambda: stage_cheaptrick(w, L, torch, dev, d_x, x_len, d_t, d_f, f_len)),
                            ("config2_16k_full_pipeline", lambda: stage_config2(w, L, torch, dev)),
                            ("config4_synthesis_only_share", lambda: stage_config4(w, L, torch, dev, pipe)),
                            ("config5_streams_share", lambda: stage_config5(w, L, torch, dev))):
                if key in os.environ.get("WC_BENCH_SKIP", "").split(","):  # (development aid: leave stages out)
                    continue
                tie_counts(L, reset=True)
                try:
                    st[key] = fn()
                except Exception as e:
                    st[key] = {"error": str(e)}
                if key in ("config2_16k_full_pipeline", "config5_streams_share"):
                    ties[key.split("_")[0]] = per_1000(*tie_counts(L, reset=True))
                torch.cuda.empty_cache()
            c3 = st.get("cheaptrick_config3", {})
            if "_k" in c3:
                k = c3.pop("_k")
                fl3 = (pmc_all or {}).get("_config3_fp64_flops")  # FP64 operations of the config-3 launch (counter pass of this build)
                if fl3:
                    c3["fp64_tflops"] = fl3 / k / 1e12
                    c3["fp64_frac"] = fl3 / k / 1e12 / FP64_VECTOR_PEAK_TFLOPS
                # the north star's stage target, what is achieved, and what the kernel's instruction count allows at all: its vector
                # instructions at the rates of tools/issue_rate.hip (profiles/issue_rates.json) for the occupancy it runs at (two
                # wavefronts per SIMD) and for a full SIMD (eight) -- the ceiling of this formulation, whatever the schedule
                c3["target_hbm_frac"] = 0.60
                n_v, n_64, rates = (pmc_all or {}).get("_config3_valu_insts"), (pmc_all or {}).get("_config3_fp64_insts"), load_issue_rates()
                if n_v and n_64 and rates:
                    byt = c3["frames"] * c3["bytes_per_frame"]
                    for tag, wv in (("at_2_waves_per_simd", 2), ("at_8_waves_per_simd", 8)):
                        t_iss = (n_64 * rate_at(rates["fp64"], wv) + (n_v - n_64) * rate_at(rates["int32"], wv)) * 1e-9 / 1024.0
                        c3["issue_floor_" + tag] = {"ms": t_iss * 1e3, "hbm_frac": byt / t_iss / (HBM_PEAK_GBS * 1e9)}
                    c3["issue_frac"] = c3["issue_floor_at_2_waves_per_simd"]["ms"] / c3["kernel_ms"]
            out["stages"] = st
            # the figures beside `value` that SURVEY.md section 8(d) and BASELINE.json's north star name, repeated under keys the
            # driver's record keeps whole (`config`, `roofline`): the transfer-inclusive headline and the batched-CheapTrick stage
            wt = (host_first.get("with_transfers") or {}).get("f64_in_all_five_out") or {}
            c2, c4 = st.get("config2_16k_full_pipeline") or {}, st.get("config4_synthesis_only_share") or {}
            also = {"value_with_transfers": host_first.get("value_with_transfers"), "with_transfers_ms": wt.get("ms"),
                    "cheaptrick_config3_ms": c3.get("ms"), "cheaptrick_config3_hbm_frac": c3.get("hbm_frac"),
                    "cheaptrick_config3_target_hbm_frac": c3.get("target_hbm_frac"),
                    "config2_ms": c2.get("ms"), "config4_ms": c4.get("ms")}
            also["ties"] = ties
            out["config"]["also_measured"] = also
            if out["roofline"] is not None:
                out["roofline"]["also_measured"] = also
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(list(cache.values()))
            except Exception as e:  # the GPU number stays valid without it
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
