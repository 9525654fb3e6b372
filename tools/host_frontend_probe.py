"""The host front-end on the headline batch (development aid): wc_pipeline_run_batch_host with page-locked rows, three timed runs;
WC_PIPELINE_TIMING=1 prints the phase marks, PIN_IN=1 page-locks the utterances too, CODED=1 adds the coded-output variant."""
import sys, time, os, numpy as np
sys.path.insert(0, '/root/repo')
import world_class_amd as w
from world_class_amd.synth import make_utterance
L = w.lib(); L.wc_set_device(0)
fs = 48000
base = [make_utterance(fs, 10.0, 2000 + u) for u in range(8)]
xs = [base[i % 8] for i in range(64)]
if os.environ.get("PIN_IN"):
    import torch
    xs = [torch.from_numpy(x).pin_memory().numpy() for x in xs]
pipe = w.Pipeline(fs)
xl = [len(x) for x in xs]
want = ("tpos", "f0", "sp", "ap", "y")
res = pipe.host_buffers(xl, want=want, pinned=True)
pipe.run_batch_host(xs, want=want, out=res)
for it in range(3):
    t0 = time.perf_counter()
    pipe.run_batch_host(xs, want=want, out=res)
    print("run %d: %.1f ms" % (it, (time.perf_counter() - t0) * 1e3), flush=True)
if os.environ.get("CODED"):
    r2 = pipe.run_batch_host_coded(xs, number_of_dimensions=60)
    for it in range(3):
        t0 = time.perf_counter()
        pipe.run_batch_host_coded(xs, number_of_dimensions=60, out=r2)
        print("coded %d: %.1f ms" % (it, (time.perf_counter() - t0) * 1e3), flush=True)
