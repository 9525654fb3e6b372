"""BASELINE config 5 alone (bench.py's stage_config5: 512 concurrent 24 kHz streams, 1 ms frames, 200 ms per push), for profiling:
    rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/stream_prof -o p -- python tools/stream_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import world_class_amd as w
L = w.lib(); L.wc_set_device(0)
r = bench.stage_config5(w, L, torch, torch.device("cuda", 0))
print(json.dumps({k: (v if isinstance(v, str) else {a: v[a] for a in ("push_ms", "frames_per_push", "real_time_factor")}) for k, v in r.items() if k != "workload"}))
