"""The drop-in caller's four compute() calls on one 48 kHz 10 s utterance, first and steady (bench.py's stage_dropin alone;
development aid): python tools/dropin_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import world_class_amd as w
from world_class_amd.synth import make_utterance
L = w.lib(); L.wc_set_device(0)
x = make_utterance(bench.FS, 10.0, 2000)
r = bench.stage_dropin(w, L, x)
for k in ("first", "steady"):
    print(k, " ".join("%s %.2f+%.2f" % (s, r[k][s]["ctor_ms"], r[k][s]["compute_ms"]) for s in ("harvest", "cheaptrick", "d4c", "synthesis")),
          "total %.2f ms" % r[k]["total_compute_ms"])
