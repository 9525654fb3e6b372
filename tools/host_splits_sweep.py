"""Group splits of the host front-end on the headline batch (64 x 48 kHz x 10 s, all five outputs, page-locked in and out), timed in
turn in ONE process (option "host_splits"): python tools/host_splits_sweep.py "5,8,12,18,25" "3,5,8,12,18,25" ..."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w
from world_class_amd.synth import make_utterance
L = w.lib(); L.wc_set_device(0)
fs = 48000
base = [make_utterance(fs, 10.0, 2000 + u) for u in range(8)]
xs = [torch.from_numpy(base[i % 8]).pin_memory().numpy() for i in range(64)]
pipe = w.Pipeline(fs)
xl = [len(x) for x in xs]
want = ("tpos", "f0", "sp", "ap", "y")
res = pipe.host_buffers(xl, want=want, pinned=True)
splits = sys.argv[1:] or ["default"]
for _ in range(2):
    pipe.run_batch_host(xs, want=want, out=res)
out = {s: [] for s in splits}
for rnd in range(4):
    for s in splits:
        pipe.set_option("host_splits", None if s == "default" else s)
        pipe.run_batch_host(xs, want=want, out=res)
        t0 = time.perf_counter()
        for _ in range(3):
            pipe.run_batch_host(xs, want=want, out=res)
        out[s].append((time.perf_counter() - t0) / 3 * 1e3)
for s in splits:
    print("host_splits=%-22s %s  median %.2f ms" % (s, " ".join("%.1f" % t for t in out[s]), float(np.median(out[s]))))
