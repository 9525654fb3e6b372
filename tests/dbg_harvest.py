"""Stage-by-stage comparison of the HIP Harvest path with the CPU oracle (development aid, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import world_class_amd as w
from oracle import port
from world_class_amd.synth import make_utterance

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
floor = float(sys.argv[4]) if len(sys.argv) > 4 else 71.0
P = port.Port()
x = make_utterance(fs, sec, seed)
t0 = time.time(); d = P.harvest_debug(x, fs, f0_floor=floor); print('oracle %.2fs' % (time.time() - t0))
h = w.Harvest(fs, f0_floor=floor)
t0 = time.time(); tpos, f0 = h.compute(x); print('gpu %.3fs' % (time.time() - t0))
L1 = len(d['f0_1ms']); nb = d['raw'].shape[0]; mc = d['cand'].shape[1]
def cmp(name, a, b):
    a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
    nzm = int(((a == 0) != (b == 0)).sum())
    both = (a != 0) & (b != 0)
    mx = np.abs(a - b)[both].max() if both.any() else 0.0
    print(f'{name:8s} n={a.size} zero-mismatch={nzm} maxabs(nonzero both)={mx:.3e} maxabs={np.abs(a-b).max():.3e}')
cmp('y', h.debug_fetch('y'), d['y'])
cmp('raw', h.debug_fetch('raw').reshape(nb, L1), d['raw'])
# candidates: the device keeps a fixed stride S per overlap block; compare as per-frame sorted multisets
S = mc // 7
gc = h.debug_fetch('cand').reshape(L1, 7 * S); gs = h.debug_fetch('score').reshape(L1, 7 * S)
oc, os_ = d['cand'], d['score']
bad = 0; mx = 0.0
for i in range(L1):
    a = np.sort(gc[i][gc[i] != 0]); b = np.sort(oc[i][oc[i] != 0])
    if len(a) != len(b): bad += 1
    elif len(a): mx = max(mx, np.abs(a - b).max())
print('cand: frames with different candidate count', bad, 'max diff', mx)
cmp('base', h.debug_fetch('base'), d['f0_base'])
cmp('fixed', h.debug_fetch('fixed'), d['f0_fixed'])
cmp('f0_1ms', h.debug_fetch('f0_1ms'), d['f0_1ms'])
tr, fr = P.harvest(x, fs, f0_floor=floor)
cmp('f0', f0, fr); print('tpos equal', np.array_equal(tpos, tr))
