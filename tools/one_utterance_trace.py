"""One 48 kHz 10 s utterance through the fused pipeline, six times (development aid; under rocprofv3 --kernel-trace the last
call's kernels are the single-utterance time line: python tools/timeline.py <kernel_trace.csv> 45)"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import world_class_amd as w
from world_class_amd.synth import make_utterance
L = w.lib(); L.wc_set_device(0)
fs = 48000
dev = torch.device("cuda", 0)
xs = [make_utterance(fs, 10.0, 2000 + (u % 8)) for u in range(int(os.environ.get("TRACE_N", "1")))]
p = w.Pipeline(fs)
xl = [len(x) for x in xs]
fl, yl = p.lengths(xl)
d_x = torch.from_numpy(np.concatenate(xs)).to(dev)
d_t = torch.empty(sum(fl), dtype=torch.float64, device=dev); d_f = torch.empty_like(d_t)
d_sp = torch.empty(sum(fl) * p.bins, dtype=torch.float64, device=dev); d_ap = torch.empty_like(d_sp)
d_y = torch.empty(sum(yl), dtype=torch.float64, device=dev)
for _ in range(6):
    p.run_device(d_x.data_ptr(), xl, d_t.data_ptr(), d_f.data_ptr(), d_sp.data_ptr(), d_ap.data_ptr(), d_y.data_ptr()); L.wc_synchronize()
