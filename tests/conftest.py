import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The tests that PIN the checkers -- the CPU restatement and the host helpers against the real reference's fixtures and, where
# oracle/_ref is there, the live reference -- need no device.  They run in the CPU suite (-m "not gpu") here, and on a box with a
# GPU they carry the gpu mark as well, so that the driver's `pytest -m gpu` run holds the pin inside the same process as the
# parity tests it pins (round-5 verdict, item 4).
PIN_MODULES = ("test_oracle_golden.py", "test_helpers.py")


def on_gpu_box():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if not on_gpu_box():
        return
    for it in items:
        if os.path.basename(str(it.fspath)) in PIN_MODULES:
            it.add_marker(pytest.mark.gpu)


class Golden:
    def __init__(self):
        d = os.path.join(ROOT, "tests", "golden")
        self.npz = np.load(os.path.join(d, "world_golden.npz"))
        with open(os.path.join(d, "world_golden.json")) as f:
            self.meta = json.load(f)

    def __getitem__(self, k):
        return self.npz[k]

    def case(self, name):
        m = dict(self.meta["cases"][name])
        m["x"] = self.npz[name + "/x_i16"].astype(np.float64) / 32768.0
        for k in ("tpos", "f0", "sp_rows", "ap_rows", "sp_rowsum", "ap_rowsum", "y"):
            m[k] = self.npz[name + "/" + k]
        return m


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def port():
    from oracle import port as _port
    return _port.Port()


class RefChecker:
    """The REAL reference (oracle/_ref/libworld_ref.so, compiled from /root/reference by oracle/Makefile) as the checker of the
    off-fixture parity tests: every signal in a process of its own (its noise stream is process-global and cannot be set back),
    the method names of oracle/port.Port.  Where the reference itself crashes on a signal (its Harvest corrupts its heap on DC
    offsets and 42-70 Hz voices, its Synthesis overflows its pulse arrays: DESIGN.md section 8) the CPU restatement answers
    instead and the log says so."""
    name = "the real reference (oracle/_ref)"

    def __init__(self, port_):
        from oracle import ref
        self.ref = ref
        self.port = port_
        self.fell_back = []

    def _fresh(self, what, method, *a, **kw):
        try:
            return self.ref.run_fresh(method, *a, **kw)
        except Exception:
            self.fell_back.append(what)
            print("checker: the real reference crashed on %s: the CPU restatement answers" % what)
            return None

    def pipeline(self, x, fs, harvest_floor=71.0, frame_period=5.0, what="a signal"):
        try:
            return self.ref.run_fresh("pipeline", x, fs, harvest_floor=harvest_floor, frame_period=frame_period)
        except Exception:
            pass
        # (the reference's Synthesis overflows its pulse arrays on many of the sweeps' signals: its Harvest alone then -- F0 and
        # voicing stay with the real reference -- and the CPU restatement for the stages behind it, on the reference's contour)
        try:
            hv = self.ref.run_fresh("harvest", x, fs, f0_floor=harvest_floor, frame_period=frame_period)
        except Exception:
            self.fell_back.append(what)
            print("checker: the real reference's Harvest crashed on %s: the CPU restatement answers" % what)
            return self.port.pipeline(x, fs, harvest_floor=harvest_floor, frame_period=frame_period)
        print("checker: the real reference's pipeline crashed on %s: its Harvest, then the CPU restatement on its contour" % what)
        return self.port.pipeline(x, fs, harvest_floor=harvest_floor, frame_period=frame_period, given_f0=hv)

    def harvest(self, x, fs, f0_floor=71.0, f0_ceil=800.0, frame_period=5.0, what="a signal"):
        o = self._fresh(what, "harvest", x, fs, f0_floor=f0_floor, f0_ceil=f0_ceil, frame_period=frame_period)
        return o if o is not None else self.port.harvest(x, fs, f0_floor=f0_floor, f0_ceil=f0_ceil, frame_period=frame_period)

    def synthesis_behind_analysis(self, x, fs, o, f0, sp, ap, frame_period=5.0):
        """Synthesis of GIVEN parameters from the place in the noise stream where the reference's own pipeline run on x reached its
        Synthesis: a fresh process repeats CheapTrick and D4C on the reference's contour (their draws depend on nothing else), then
        synthesises (f0, sp, ap)"""
        return self.ref.run_fresh("synthesis_behind_analysis", x, fs, o["tpos"], o["f0"], f0, sp, ap, frame_period)

    def stage_at(self, start, method, *a, **kw):
        """one stage call with the noise stream `start` draws from its seed"""
        return self.ref.run_fresh("at", int(start), method, *a, **kw)


@pytest.fixture(scope="session")
def checker(port):
    """the real reference where oracle/_ref is there (it travels to the GPU box with the tree), the CPU restatement otherwise"""
    from oracle import ref
    if ref.available():
        c = RefChecker(port)
        print("checker:", c.name)
        return c
    print("checker: the CPU restatement (oracle/port.py) -- oracle/_ref is not built")
    return None


HARVEST_LONG_CASES = ["tie_48k_10s_9033", "plain_48k_10s_9001", "plain_16k_10s_12003_floor40", "edge_rows_16k_3s_duet",
                      "equal_refined_16k_3s_loud"]


def harvest_long_case(name):
    """(x, fs, f0_floor, expected F0 of the real reference) from tests/golden/harvest_long.npz (oracle/gen_golden_harvest.py);
    the samples are regenerated from the seed and checked against the stored digest"""
    import hashlib
    from world_class_amd.synth import make_utterance
    z = np.load(os.path.join(ROOT, "tests", "golden", "harvest_long.npz"))
    fs, sec, seed, floor = z[name + "/meta"]
    if name + "/x_i16" in z:
        x = z[name + "/x_i16"].astype(np.float64) / 32768.0
    elif name + "/x_f32" in z:
        x = z[name + "/x_f32"].astype(np.float64)
    else:
        x = make_utterance(int(fs), float(sec), int(seed))
        assert hashlib.sha256(x.tobytes()).digest() == z[name + "/x_sha256"].tobytes(), "synthetic generator drifted"
    return x, int(fs), float(floor), z[name + "/f0"]


def harvest_edge_rows():
    """candidates of frames 1 and L-2 after removeUnreliableCandidates of the real reference for "edge_rows_16k_3s_duet":
    (frame indices, [2][max_candidates])"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "harvest_long.npz"))
    return z["edge_rows_16k_3s_duet/cand_rows"], z["edge_rows_16k_3s_duet/cand"]


def same_candidates(a, b, tol=1e-8):
    a, b = np.sort(a[a != 0]), np.sort(b[b != 0])
    return len(a) == len(b) and (len(a) == 0 or np.abs(a - b).max() < tol)


def harvest_option_cases():
    """[(name, x, fs, options, expected F0 of the real reference)] from tests/golden/harvest_options.npz
    (oracle/gen_golden_harvest_options.py): target_fs, channels_in_octave, use_cos_table"""
    from world_class_amd.synth import make_utterance
    d = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(d, "harvest_options.npz"))
    with open(os.path.join(d, "harvest_options.json")) as f:
        meta = json.load(f)
    return [(name, make_utterance(m["fs"], m["seconds"], m["seed"]), m["fs"], m["options"], z[name + "/f0"]) for name, m in sorted(meta.items())]


def stage_option_cases():
    """CheapTrickOption / D4COption away from their defaults on one 16 kHz utterance (tests/golden/stage_options.npz, made by
    oracle/gen_golden_stage_options.py from the real reference): x, fs, tpos, f0, row stride,
    [(name, CheapTrick kwargs, rows, rowsum)], [(name, threshold, rows, rowsum)]"""
    from world_class_amd.synth import make_utterance
    z = np.load(os.path.join(ROOT, "tests", "golden", "stage_options.npz"))
    fs = 16000
    x = make_utterance(fs, 1.0, 4321)
    f0 = z["f0"]
    tpos = np.arange(len(f0)) * 5.0 / 1000.0
    ct = [("q1_-0.09", dict(q1=-0.09)), ("floor40", dict(f0_floor=40.0)), ("fft2048", dict(fft_size=2048)),
          ("q1_-0.3_floor100", dict(q1=-0.3, f0_floor=100.0)), ("fft4096", dict(fft_size=4096))]
    d4 = [("thr0", 0.0), ("thr0.5", 0.5), ("thr0.95", 0.95)]
    return (x, fs, tpos, f0, 16, [(n, kw, z["ct/" + n + "/rows"], z["ct/" + n + "/rowsum"]) for n, kw in ct],
            [(n, t, z["d4c/" + n + "/rows"], z["d4c/" + n + "/rowsum"]) for n, t in d4])


def rate96k_case():
    """x, fs and the real reference's outputs for one 96 kHz utterance (tests/golden/rate96k.npz, oracle/gen_golden_96k.py)"""
    from world_class_amd.synth import make_utterance
    z = np.load(os.path.join(ROOT, "tests", "golden", "rate96k.npz"))
    return make_utterance(96000, 0.6, 9600), 96000, 16, z


PIPELINE_CASES = ["c1_16k_2s_floor71", "c1_16k_2s_floor40", "m48k_1s", "m24k_1s_1ms"]


def headline_case(u, fixture="headline_48k_10s.npz"):
    """utterance u (0 .. 7) of the benchmark's workload at its full size, 48 kHz x 10 s, with what the real reference's full
    pipeline returns for it (tests/golden/headline_48k_10s.npz, oracle/gen_golden_headline.py): x and a dict of f0, sp/ap row
    sums and every `stride`-th row, block sums and windows of the waveform.  fixture = "config2_16k_10s.npz": the same for
    BASELINE config 2's 16 kHz x 10 s utterances (oracle/gen_golden_config2.py)."""
    import hashlib
    from world_class_amd.synth import make_utterance
    z = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    k = "u%d/" % u
    fs, sec, seed, stride, block, win = z[k + "meta"]
    x = make_utterance(int(fs), float(sec), int(seed))
    assert hashlib.sha256(x.tobytes()).digest() == z[k + "x_sha256"].tobytes(), "synthetic generator drifted"
    g = {n: z[k + n] for n in ("f0", "sp_rowsum", "ap_rowsum", "sp_rows", "ap_rows", "y_blocksum", "y_win_start", "y_win")}
    g.update(fs=int(fs), stride=int(stride), block=int(block), win=int(win), y_len=int(z[k + "y_len"][0]))
    return x, g


def check_headline(r, g, f0_abs, sp_rel, ap_abs, y_abs):
    """outputs of a full-size pipeline run against headline_case's expectations"""
    assert np.array_equal(r["f0"] == 0, g["f0"] == 0)
    assert np.abs(r["f0"] - g["f0"]).max() < f0_abs
    s = g["stride"]
    assert (np.abs(r["sp"][::s] - g["sp_rows"]) / g["sp_rows"]).max() < sp_rel
    assert np.abs(r["ap"][::s] - g["ap_rows"]).max() < ap_abs
    bins = r["sp"].shape[1]
    assert (np.abs(r["sp"].sum(1) - g["sp_rowsum"]) / g["sp_rowsum"]).max() < sp_rel  # every row, through its sum
    assert np.abs(r["ap"].sum(1) - g["ap_rowsum"]).max() < ap_abs * bins
    y = r["y"]
    assert len(y) == g["y_len"]
    for st, w in zip(g["y_win_start"], g["y_win"]):
        assert np.abs(y[st:st + g["win"]] - w).max() < y_abs
    nb = len(y) // g["block"]
    assert np.abs(y[:nb * g["block"]].reshape(nb, g["block"]).sum(1) - g["y_blocksum"]).max() < y_abs * g["block"]
