"""A/B of pipeline options inside ONE process (the headline workload, resident): python tools/schedule_ab.py name=v1,v2,... [steps]
Every value is timed `rounds` times in turn (20 steps each), so that drifts of the box show up as such."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w
from world_class_amd.synth import make_utterance
name, vals = sys.argv[1].split("=")
vals = vals.split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fs, n = 48000, 64
L = w.lib()
dev = torch.device("cuda", 0)
base = [make_utterance(fs, 10.0, 3000 + u) for u in range(8)]
xs = [base[u % 8] for u in range(n)]
p = w.Pipeline(fs)
xl = [len(x) for x in xs]
fl, yl = p.lengths(xl)
d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
d_t = torch.empty(sum(fl), dtype=torch.float64, device=dev); d_f = torch.empty_like(d_t)
d_sp = torch.empty(sum(fl) * p.bins, dtype=torch.float64, device=dev); d_ap = torch.empty_like(d_sp)
d_y = torch.empty(sum(yl), dtype=torch.float64, device=dev)
def run():
    p.run_device(d_x.data_ptr(), xl, d_t.data_ptr(), d_f.data_ptr(), d_sp.data_ptr(), d_ap.data_ptr(), d_y.data_ptr())
for _ in range(5): run()
L.wc_synchronize()
res = {v: [] for v in vals}
for rnd in range(5):
    for v in vals:
        p.set_option(name, None if v == "default" else v)
        run(); L.wc_synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): run()
        L.wc_synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
for v in vals:
    print("%s=%s: %s  median %.3f ms" % (name, v, " ".join("%.2f" % t for t in res[v]), float(np.median(res[v]))))
