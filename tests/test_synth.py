"""The seeded signal generators (world_class_amd/synth.py) are what the gated sweeps, the goldens' inputs and the bench stand on:
a numpy upgrade that changed them would silently move every one of those.  The utterances are pinned through their int16
quantisation bit for bit; the float64 kinds of the two sweep sets through three sums each (1e-10 relative: numpy's SIMD sin / cos may
differ in the last bit between hosts)."""
import hashlib

import numpy as np

from world_class_amd.synth import SIGNAL_KINDS, SIGNAL_KINDS2, make_signal, make_signal2, make_utterance

ZOO = [[3875.8933367053132, 0.04564742685059637, -357263.30634537915], [6111.2167168939795, 0.03188960751682624, -19960.55386563094],
       [1975.8801797079966, -0.08010069867737382, -61326.700957282505], [89.10000000000001, 0.0, 711641.7],
       [2856.260891673592, 0.10023875571220625, -52956.451959481725], [7996.0, -0.3, 25728067.999999996],
       [0.06498200890749156, -1.7198344333332987e-06, -0.8693619538765343], [20886.101246169954, -1.8842940220322204, 1084993.505993649],
       [1606.8964805798819, 0.0, 9436.37590405905], [2979.6838048759964, -0.07737772126437513, -78877.938393736]]
ZOO2 = [[1004.2770080566406, 0.0030517578125, 129460.74133300781], [6944.782145612559, -0.5, -2847631.6011236766],
        [928.9983458406338, 2.6077283961865995e-13, -42208.72660899862], [7.220733642578125, 0.00030517578125, 68.15994262695312],
        [1214.510544671618, -0.0006814065964533536, -73569.70139907248], [246.97857001520742, 0.0, 990813.0181129184],
        [4799.993653609412, 0.3002975343793953, 38397422.230082214], [3514.6725642441393, 0.3630504598757329, 16015.810668821025],
        [2904.4368021048676, -0.1913769296475906, -98708.65918643697], [101.22981226803634, 4.892265601606015e-07, -19.1362716425538]]


def test_utterances_are_pinned_bit_for_bit():
    def h(x):
        return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()[:16]
    assert h(make_utterance(16000, 1.0, 2000)) == "1ca5f7bbc4e8585f"
    assert h(make_utterance(48000, 1.0, 2001)) == "e2f2e79fdc54d5db"


def test_both_sets_of_sweep_signals_are_pinned():
    assert len(SIGNAL_KINDS) == len(SIGNAL_KINDS2) == 10
    for want, fn, seed0 in ((ZOO, make_signal, 230000), (ZOO2, make_signal2, 1800000)):
        for i, w in enumerate(want):
            x = fn(16000, 1.0, seed0 + i)
            got = [float(np.abs(x).sum()), float(x[7777]), float((x * np.arange(len(x))).sum())]
            scale = float(np.abs(x).sum())
            assert abs(got[0] - w[0]) <= 1e-10 * scale, (fn.__name__, i)
            assert abs(got[1] - w[1]) <= 1e-10 * max(abs(w[1]), float(np.abs(x).max())), (fn.__name__, i)
            assert abs(got[2] - w[2]) <= 1e-10 * scale * len(x), (fn.__name__, i)
