"""BASELINE.json `configs` 2-5 on ONE MI355X (per-GPU share of the multi-GPU ones), synthetic inputs resident in HBM.
Prints one JSON line per config.  Not the graded bench (that is bench.py, the headline metric); this is the
evidence table quoted in DESIGN.md section 5.

    python tools/run_configs.py [--configs 2,3,4,5] [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w  # noqa: E402
from world_class_amd.synth import make_utterance, true_f0  # noqa: E402


def timed(fn, L, iters=3):
    fn()
    L.wc_synchronize()
    best = 1e9
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        L.wc_synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def tile(items, n):
    return [items[i % len(items)] for i in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5,6,7")
    ap.add_argument("--scale", type=float, default=1.0, help="scale the utterance counts (for quick runs)")
    ap.add_argument("--stream-modes", default="", help="config 5: only these variants, e.g. 200:560:160 (chunk:lookahead:context ms)")
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    L = w.lib()
    L.wc_set_device(0)
    todo = [int(c) for c in a.configs.split(",")]

    if 2 in todo:  # 64 x 16 kHz x 10 s, full pipeline
        fs, n = 16000, max(2, int(64 * a.scale))
        xs = tile([make_utterance(fs, 10.0, 2000 + u) for u in range(8)], n)
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
        t = timed(lambda: p.run_device(d_x, xl, d_t, d_f, d_sp, d_ap, d_y), L)
        print(json.dumps({"config": 2, "what": f"{n} x 16 kHz 10 s, 5 ms hop, full pipeline, 1 GPU", "frames": sum(fl),
                          "ms": t * 1e3, "frames_per_s": sum(fl) / t}))
        del d_x, d_t, d_f, d_sp, d_ap, d_y, p

    if 3 in todo:  # 256 x 48 kHz x 10 s, CheapTrick only
        fs, n = 48000, max(2, int(256 * a.scale))
        xs = tile([make_utterance(fs, 10.0, 3000 + u) for u in range(8)], n)
        tf = tile([true_f0(fs, 10.0, 3000 + u) for u in range(8)], n)
        ct = w.CheapTrick(fs)
        xl, fl = [len(x) for x in xs], [len(t_) for t_, _ in tf]
        d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
        d_t = torch.from_numpy(np.concatenate([t_ for t_, _ in tf])).to(dev)
        d_f = torch.from_numpy(np.concatenate([f for _, f in tf])).to(dev)
        d_sp = torch.empty(sum(fl) * ct.bins, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        L.wc_set_kernel_timing(1)
        t = timed(lambda: ct.compute_device(d_x, xl, d_t, d_f, fl, d_sp), L)
        k = L.wc_last_kernel_ms(b"cheaptrick_frames") * 1e-3
        L.wc_set_kernel_timing(0)
        b_ct = 8 * ct.fft_size + 8 * ct.bins
        print(json.dumps({"config": 3, "what": f"{n} x 48 kHz 10 s, CheapTrick only (contour given), 1 GPU", "frames": sum(fl),
                          "ms": t * 1e3, "frames_per_s": sum(fl) / t, "kernel_ms": k * 1e3,
                          "B_ct_GBps": sum(fl) * b_ct / k / 1e9, "hbm_frac_on_B_ct": sum(fl) * b_ct / k / 8e12}))
        del d_x, d_t, d_f, d_sp, ct

    if 4 in todo:  # Synthesis only from precomputed {f0, sp, ap}; per-GPU share of 1024 utterances = 128
        fs, n = 48000, max(2, int(128 * a.scale))
        base = []
        p = w.Pipeline(fs)
        for u in range(4):  # real parameters of 4 utterances, tiled
            (r,) = p.run_batch([make_utterance(fs, 10.0, 4000 + u)])
            base.append(r)
        del p
        sy = w.Synthesis(fs, 2048, 5.0)
        fl = [len(base[i % 4]["f0"]) for i in range(n)]
        yl = [sy.out_length(v) for v in fl]
        d_f = torch.from_numpy(np.concatenate([base[i % 4]["f0"] for i in range(n)])).to(dev)
        sp4 = [torch.from_numpy(b["sp"].ravel()).to(dev) for b in base]
        ap4 = [torch.from_numpy(b["ap"].ravel()).to(dev) for b in base]
        d_sp = torch.cat([sp4[i % 4] for i in range(n)])
        d_ap = torch.cat([ap4[i % 4] for i in range(n)])
        d_y = torch.empty(sum(yl), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t = timed(lambda: sy.compute_device(d_f, fl, d_sp, d_ap, yl, d_y), L)
        print(json.dumps({"config": 4, "what": f"{n} x 48 kHz 10 s (1/8 of 1024), Synthesis only from resident f0/sp/ap "
                                                f"({(d_sp.numel() + d_ap.numel()) * 8 / 1e9:.1f} GB), 1 GPU",
                          "frames": sum(fl), "ms": t * 1e3, "frames_per_s": sum(fl) / t}))
        del d_f, d_sp, d_ap, d_y, sy, sp4, ap4

    if 5 in todo:  # streaming 1 ms hop Harvest + CheapTrick at 24 kHz; per-GPU share of 4096 streams = 512
        from world_class_amd.stream import StreamAnalyzer
        fs, n = 24000, max(2, int(512 * a.scale))
        sig = [make_utterance(fs, 4.0, 5000 + u) for u in range(8)]
        variants = ((200, 400, 400, 0), (200, 400, 560, 160), (400, 400, 400, 0), (400, 400, 560, 160), (80, 400, 400, 0), (80, 400, 560, 160))
        if a.stream_modes:
            variants = tuple((int(c), 400, int(ah), int(cx)) for c, ah, cx in (m.split(":") for m in a.stream_modes.split(",")))
        for chunk_ms, back_ms, ahead_ms, ctx_ms in variants:
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
            steady = times[full:]
            t = float(np.median(steady))
            print(json.dumps({"config": 5, "what": f"{n} concurrent 24 kHz streams (1/8 of 4096), 1 ms frames, chunked Harvest + CheapTrick "
                                                    f"(include/world_class_stream.h): chunk {chunk_ms} ms, lookback {back_ms} ms, lookahead {ahead_ms} ms, "
                                                    + (f"incremental (context {ctx_ms} ms)" if ctx_ms else "whole windows") + ", 1 GPU",
                              "frames_per_push": frames[-1], "push_ms": t * 1e3, "push_ms_max": max(steady) * 1e3,
                              "frames_per_s": frames[-1] / t, "algorithmic_latency_ms": ahead_ms + chunk_ms,
                              "latency_ms_incl_compute": ahead_ms + chunk_ms + t * 1e3, "real_time_factor": chunk_ms / (t * 1e3),
                              "streams_sustainable_in_real_time": int(n * chunk_ms / (t * 1e3))}))
            del sa, chunks, d_t, d_f, d_sp
        # for comparison: the same streams as whole 2 s utterances in one batch (no chunking)
        xs = tile([make_utterance(fs, 2.0, 5000 + u) for u in range(8)], n)
        hv, ct = w.Harvest(fs, frame_period=1.0), w.CheapTrick(fs)
        xl = [len(x) for x in xs]
        fl = [hv.get_samples(v) for v in xl]
        d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
        d_t = torch.empty(sum(fl), dtype=torch.float64, device=dev)
        d_f = torch.empty_like(d_t)
        d_sp = torch.empty(sum(fl) * ct.bins, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()

        def run():
            hv.compute_device(d_x, xl, d_t, d_f)
            ct.compute_device(d_x, xl, d_t, d_f, fl, d_sp)
        t = timed(run, L)
        print(json.dumps({"config": "5-whole", "what": f"{n} x 24 kHz 2 s as whole utterances in one batch, 1 ms hop, Harvest + CheapTrick, 1 GPU",
                          "frames": sum(fl), "ms": t * 1e3, "frames_per_s": sum(fl) / t}))

    if 7 in todo:  # PCIe-inclusive: host batch front-end, int16 PCM in, f0 + int16 waveform out (and everything out)
        fs, n = 48000, max(2, int(64 * a.scale))
        xs = tile([make_utterance(fs, 10.0, 3000 + u) for u in range(8)], n)
        pcm = [np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16) for x in xs]
        p = w.Pipeline(fs)
        frames = sum(p.lengths([len(v) for v in pcm])[0])
        for want, label in ((("f0", "y"), "f0 + int16 waveform back"), (("tpos", "f0", "sp", "ap", "y"), "all five outputs back (2.1 GB of sp + ap)")):
            res = p.run_batch_host(pcm, want=want, y_pcm16=True)  # the caller's result buffers, written again below (no fresh pages)
            t0 = time.perf_counter()
            p.run_batch_host(pcm, want=want, y_pcm16=True, out=res)
            t = time.perf_counter() - t0
            del res
            print(json.dumps({"config": "host", "what": f"{n} x 48 kHz 10 s from host int16 PCM through pinned staging, full pipeline, {label}",
                              "frames": frames, "ms": t * 1e3, "frames_per_s": frames / t}))
        del p

    if 6 in todo:  # section 8(f) kernels: int16 PCM expansion and the demo's parameter modification, HBM-bound
        from world_class_amd import io as wio
        n = 64 * 480000
        d_pcm = torch.randint(-32768, 32767, (n,), dtype=torch.int16, device=dev)
        d_x = torch.empty(n, dtype=torch.float64, device=dev)
        t = timed(lambda: wio.pcm16_to_double_device(d_pcm, n, d_x), L)
        print(json.dumps({"config": "pcm16->f64", "what": "64 x 48 kHz x 10 s of int16 PCM expanded on the device", "ms": t * 1e3,
                          "GBps": n * 10 / t / 1e9, "hbm_frac": n * 10 / t / 8e12}))
        frames, fft = 128064, 2048
        d_f0 = torch.full((frames,), 200.0, dtype=torch.float64, device=dev)
        d_sp = torch.rand(frames * (fft // 2 + 1), dtype=torch.float64, device=dev) + 1e-6
        t = timed(lambda: wio.modify_parameters_device(48000, fft, frames, d_f0, d_sp, 1.0, 1.1), L)
        b = frames * (fft // 2 + 1) * 16
        print(json.dumps({"config": "modify", "what": "spectral stretching of 128064 rows of 1025 bins in place (log, interp1, exp)",
                          "ms": t * 1e3, "GBps": b / t / 1e9, "hbm_frac": b / t / 8e12}))


if __name__ == "__main__":
    main()
