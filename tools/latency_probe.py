"""Wall time of one fused-pipeline call for small batches of 10 s / 48 kHz utterances (request latency rather than batch
throughput): python tools/latency_probe.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import world_class_amd as w
from world_class_amd.synth import make_utterance
L = w.lib(); L.wc_set_device(0)
fs = int(os.environ.get("LAT_FS", "48000"))
dev = torch.device("cuda", 0)
for n in [int(v) for v in os.environ.get("LAT_N", "1,2,4,8,16").split(",")]:
    xs = [make_utterance(fs, 10.0, 2000 + u) for u in range(n)]
    p = w.Pipeline(fs)
    xl = [len(x) for x in xs]
    fl, yl = p.lengths(xl)
    d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
    d_t = torch.empty(sum(fl), dtype=torch.float64, device=dev); d_f = torch.empty_like(d_t)
    d_sp = torch.empty(sum(fl) * p.bins, dtype=torch.float64, device=dev); d_ap = torch.empty_like(d_sp)
    d_y = torch.empty(sum(yl), dtype=torch.float64, device=dev)
    def run():
        p.run_device(d_x.data_ptr(), xl, d_t.data_ptr(), d_f.data_ptr(), d_sp.data_ptr(), d_ap.data_ptr(), d_y.data_ptr())
    run(); L.wc_synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(); L.wc_synchronize(); best = min(best, time.perf_counter() - t0)
    print("utterances %2d: fused pipeline %.2f ms" % (n, best * 1e3))
