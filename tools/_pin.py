import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WC_PIPELINE_TIMING"]="1"
import world_class_amd as w
from world_class_amd.synth import make_utterance
fs=48000
base=[make_utterance(fs,10.0,3000+u) for u in range(8)]
xs=[base[i%8] for i in range(64)]
p=w.Pipeline(fs)
xl=[len(x) for x in xs]
res=p.host_buffers(xl,pinned=True)
p.run_batch_host(xs,out=res)
for _ in range(2):
    sys.stderr.write("--- run\n")
    t0=time.perf_counter(); p.run_batch_host(xs,out=res); print("total %.1f ms"%((time.perf_counter()-t0)*1e3))

res=p.run_batch_host_coded(xs)
for _ in range(2):
    sys.stderr.write("--- coded run\n")
    t0=time.perf_counter(); p.run_batch_host_coded(xs,out=res); print("coded total %.1f ms"%((time.perf_counter()-t0)*1e3))
