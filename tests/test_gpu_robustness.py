"""-m gpu: capacity-retry paths, long and odd-sized inputs, option variations, noise-stream continuity across calls."""
import os

import numpy as np
import pytest

from world_class_amd.synth import make_utterance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wca():
    import world_class_amd as w
    w.lib()
    return w


def check_f0(f0, ref):
    assert np.array_equal(f0 == 0, ref == 0)
    assert np.abs(f0 - ref).max() < 1e-6


def test_harvest_long_utterance(wca, port):
    fs = 16000
    x = np.concatenate([make_utterance(fs, 6.0, 200 + i) for i in range(4)])  # 24 s, many voiced sections
    t, f = wca.Harvest(fs).compute(x)
    tr, fr = port.harvest(x, fs)
    assert np.array_equal(t, tr)
    check_f0(f, fr)


def test_harvest_zero_crossing_buffer_overflow_is_retried(wca, port):
    fs = 16000
    x = make_utterance(fs, 1.0, 321)
    os.environ["WC_DEBUG_SMALL_CAPS"] = "1"
    try:
        t, f = wca.Harvest(fs).compute(x)
        (r,) = wca.Pipeline(fs).run_batch([x])
    finally:
        del os.environ["WC_DEBUG_SMALL_CAPS"]
    tr, fr = port.harvest(x, fs)
    check_f0(f, fr)
    check_f0(r["f0"], fr)


def test_synthesis_pulse_buffer_overflow_is_retried(wca, port):
    # F0 above the rate bound of the pulse buffers (960 Hz) forces the hard-bound retry
    from oracle.gen_golden import synth_params
    fs, n = 48000, 2048
    f0, sp, ap = synth_params(fs, n, 41, 777)
    f0 = np.where(f0 > 0, 1500.0, 0.0)
    s = wca.Synthesis(fs, n, 5.0)
    wca.rng_set_position(0)
    port.rng_reset()
    y = s.compute(f0, sp, ap)
    assert np.abs(y - port.synthesis(f0, sp, ap, fs, 5.0)).max() < 1e-8
    port.rng_reset()


@pytest.mark.parametrize("floor,ceil", [(50.0, 600.0), (100.0, 400.0), (71.0, 1000.0), (20.0, 800.0), (30.0, 1200.0)])
def test_harvest_option_variations(wca, port, floor, ceil):
    fs = 16000
    x = make_utterance(fs, 1.0, 55)
    t, f = wca.Harvest(fs, f0_floor=floor, f0_ceil=ceil).compute(x)
    tr, fr = port.harvest(x, fs, f0_floor=floor, f0_ceil=ceil)
    check_f0(f, fr)


@pytest.mark.parametrize("n", [801, 1000, 12345, 16001])
def test_odd_lengths_through_the_pipeline(wca, port, n):
    fs = 16000
    x = make_utterance(fs, 1.1, 66)[:n]
    (r,) = wca.Pipeline(fs).run_batch([x])
    ref = port.pipeline(x, fs)
    check_f0(r["f0"], ref["f0"])
    assert (np.abs(r["sp"] - ref["sp"]) / ref["sp"]).max() < 1e-7
    assert np.abs(r["ap"] - ref["ap"]).max() < 1e-7
    assert np.abs(r["y"] - ref["y"]).max() < 1e-8


def test_noise_stream_continues_across_calls_like_one_reference_process(wca, port):
    """Two utterances analysed and synthesised back to back through the host API consume one global stream,
    exactly as two runs of the demo's stage sequence inside one reference process would."""
    fs = 16000
    xs = [make_utterance(fs, 0.5, 91), make_utterance(fs, 0.4, 92)]
    wca.rng_set_position(0)
    port.rng_reset()
    hv, ct, d4 = wca.Harvest(fs), wca.CheapTrick(fs), wca.D4C(fs)
    sy = wca.Synthesis(fs, ct.fft_size, 5.0)
    for x in xs:
        t, f = hv.compute(x)
        sp = ct.compute(x, t, f)
        ap = d4.compute(x, t, f, ct.fft_size)
        y = sy.compute(f, sp, ap)
        tr, fr = port.harvest(x, fs)
        spr = port.cheaptrick(x, fs, tr, fr)
        apr = port.d4c(x, fs, tr, fr, ct.fft_size)
        yr = port.synthesis(fr, spr, apr, fs)
        assert wca.rng_get_position() == port.rng_position()
        assert np.abs(y - yr).max() < 1e-8
    port.rng_reset()


def test_full_size_batch_properties(wca):
    """The benchmark's own workload shape (48 kHz, 10 s utterances) through the fused pipeline: every copy of
    the same utterance in a batch yields identical analysis results, and a second run reproduces the first."""
    fs = 48000
    xs = [make_utterance(fs, 10.0, 3000), make_utterance(fs, 10.0, 3001)]
    p = wca.Pipeline(fs)
    a = p.run_batch([xs[0], xs[1], xs[0], xs[1], xs[0]])
    b = p.run_batch([xs[1], xs[0]])
    for i, j in ((0, 2), (0, 4), (1, 3)):
        for k in ("f0", "sp", "ap"):
            assert np.array_equal(a[i][k], a[j][k]), (i, j, k)
        assert np.abs(a[i]["y"] - a[j]["y"]).max() < 1e-12
    assert np.array_equal(a[0]["f0"], b[1]["f0"]) and np.array_equal(a[1]["sp"], b[0]["sp"])
    assert all(len(r["f0"]) == 2001 and r["sp"].shape == (2001, 1025) and len(r["y"]) == 480001 for r in a)


def test_digital_silence(wca, port):
    """exact-zero runs: the reference's fixStep1 reads uninitialised memory there; the oracle
    and the device define it as upstream WORLD does (zeros), so both must agree, and nothing may turn non-finite"""
    fs = 16000
    x = make_utterance(fs, 1.2, 555)
    x[6000:11000] = 0.0
    (r,) = wca.Pipeline(fs).run_batch([x])
    o = port.pipeline(x, fs)
    check_f0(r["f0"], o["f0"])
    assert np.isfinite(r["sp"]).all() and np.isfinite(r["ap"]).all() and np.isfinite(r["y"]).all()
    assert (np.abs(r["sp"] - o["sp"]) / o["sp"]).max() < 1e-7
    assert np.abs(r["ap"] - o["ap"]).max() < 1e-7
    assert np.abs(r["y"] - o["y"]).max() < 1e-8
    z = np.zeros(8000)
    (rz,) = wca.Pipeline(fs).run_batch([z])
    assert not rz["f0"].any() and np.isfinite(rz["sp"]).all() and np.isfinite(rz["y"]).all()
    oz = port.pipeline(z, fs)
    assert np.abs(rz["y"] - oz["y"]).max() < 1e-8


def test_no_voiced_section_and_noise_free_bands(wca, port):
    """(1) White noise leaves no voiced section after fixStep2: the reference then copies a channel that does not exist
    (reference src/harvest.cpp:515, a crash); here the contour is all unvoiced, as in the oracle.  (2) A clean synthetic voice
    with digital-silence gaps has bands without any noise: LinearSmoothing's cumulative sum has to stay non-decreasing there
    or the logarithm of a negative difference turns whole envelope rows into NaN."""
    from world_class_amd.synth import make_signal
    fs = 16000
    noise = make_signal(fs, 1.0, 50020)
    (r,) = wca.Pipeline(fs).run_batch([noise])
    o = port.pipeline(noise, fs)
    assert not o["f0"][:-1].any()  # (the very last frame is outside every section and keeps its base value)
    check_f0(r["f0"], o["f0"])
    assert np.isfinite(r["sp"]).all() and np.isfinite(r["ap"]).all() and np.isfinite(r["y"]).all()
    assert (np.abs(r["sp"] - o["sp"]) / o["sp"]).max() < 1e-7 and np.abs(r["y"] - o["y"]).max() < 1e-8
    for seed in (40018, 40022, 40008):  # "gaps" and "jumps"
        x = make_signal(fs, 3.0, seed)
        (r,) = wca.Pipeline(fs).run_batch([x])
        check_f0(r["f0"], port.harvest(x, fs)[1])
        assert np.isfinite(r["sp"]).all() and np.isfinite(r["ap"]).all() and np.isfinite(r["y"]).all()


@pytest.mark.timeout(120)
def test_non_finite_input_stays_local(wca):
    """NaN / inf samples (a bad decode upstream) and absurd F0 values: every call returns, nothing non-finite leaks into the
    frames that do not see the bad samples, and the other utterance of the batch is bit for bit what it is on its own"""
    fs = 16000
    good = make_utterance(fs, 0.8, 11)
    (alone,) = wca.Pipeline(fs).run_batch([good])
    for bad in (np.nan, np.inf, -np.inf, 1e300):
        x = make_utterance(fs, 1.0, 12)
        x[5000:5010] = bad
        r_bad, r_good = wca.Pipeline(fs).run_batch([x, good])
        assert np.array_equal(r_good["f0"], alone["f0"]) and np.array_equal(r_good["sp"], alone["sp"]) and np.array_equal(r_good["ap"], alone["ap"])
        assert np.abs(r_good["y"] - alone["y"]).max() < 1e-12
        assert np.isfinite(r_bad["f0"]).all()
        assert np.isfinite(r_bad["sp"]).all(axis=1).sum() >= len(r_bad["f0"]) - 4  # the frames whose window holds the bad samples
        assert np.isfinite(r_bad["y"]).mean() > 0.85
    n = (alone["sp"].shape[1] - 1) * 2
    k = np.arange(len(alone["f0"]))
    for f0 in (np.where(k % 17 == 3, np.nan, alone["f0"]), -alone["f0"], alone["f0"] * 1e6, np.where(alone["f0"] > 0, np.inf, 0.0)):
        y = wca.Synthesis(fs, n, 5.0).compute(f0, alone["sp"], alone["ap"])
        assert np.isfinite(y).all()
    for f0 in (np.where(k % 13 == 2, np.nan, alone["f0"]), np.where(k % 13 == 2, 1e9, alone["f0"]), np.where(k % 13 == 2, 1e-9, alone["f0"])):
        sp = wca.CheapTrick(fs).compute(good, alone["tpos"], f0)
        ap = wca.D4C(fs).compute(good, alone["tpos"], f0, n)
        ok = np.isfinite(f0)
        assert np.isfinite(sp[ok]).all() and np.isfinite(ap).all()
