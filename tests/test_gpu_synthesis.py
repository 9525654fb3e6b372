"""-m gpu parity tests of the HIP Synthesis path (through the C-ABI): golden waveforms from the real
reference (exact noise stream position included) and the CPU oracle on seeded inputs."""
import numpy as np
import pytest

from world_class_amd.synth import make_utterance

pytestmark = pytest.mark.gpu

# waveform parity, absolute (signals are O(0.5)); SURVEY.md section 8(c) proposes 1e-8 with exact RNG
Y_ABS = 1e-8


@pytest.fixture(scope="module")
def wca():
    import world_class_amd as w
    w.lib()
    return w


def test_synthesis_only_golden(golden, wca):
    from oracle.gen_golden import synth_params
    m = golden.meta["synth_only"]
    f0, sp, ap = synth_params(m["fs"], m["fft_size"], m["n_frames"], m["seed"])
    s = wca.Synthesis(m["fs"], m["fft_size"], m["frame_period"])
    wca.rng_set_position(0)
    y = s.compute(f0, sp, ap)
    assert np.abs(y - golden["synth_only/y"]).max() < Y_ABS


def test_synthesis_only_golden_48k_sixteen_utterances(wca):
    """SURVEY section 8(c) golden (3) at BASELINE config 4's size: Synthesis ALONE from given {f0, spectrogram, aperiodicity},
    sixteen utterances of 48 kHz x 10 s against the real reference's waveforms (tests/golden/synth_only_48k_10s.npz,
    oracle/gen_golden_synth48k.py: reference src/synthesis.cpp:77-177, every utterance in a process of its own).  Sixteen in one
    stage call: the two-halves path of the Synthesis stage (n_utt >= 16), through the batch entry point, noise positions given."""
    import os
    from oracle.gen_golden import synth_params
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synth_only_48k_10s.npz"))
    fs, fft, frames, fp, n_utt, first_seed, block, win = [int(v) for v in z["meta"]]
    params = [synth_params(fs, fft, frames, first_seed + u) for u in range(n_utt)]
    for u, (f0, sp, ap) in enumerate(params):  # (the regenerated parameters are the ones the fixture was made from)
        assert np.allclose([f0.sum(), sp.sum(), ap.sum()], z["u%d/param_sums" % u], rtol=1e-12, atol=0)
    s = wca.Synthesis(fs, fft, float(fp))
    ys = s.compute_batch([p[0] for p in params], [p[1] for p in params], [p[2] for p in params], rng_pos=[0] * n_utt)
    ys = ys[0] if isinstance(ys, tuple) else ys
    worst = 0.0
    for u, y in enumerate(ys):
        k = "u%d/" % u
        assert len(y) == int(z[k + "y_len"][0])
        for st, w in zip(z[k + "y_win_start"], z[k + "y_win"]):
            worst = max(worst, float(np.abs(y[st:st + win] - w).max()))
        nb = len(y) // block
        assert np.abs(y[:nb * block].reshape(nb, block).sum(1) - z[k + "y_blocksum"]).max() < Y_ABS * block
    assert worst < Y_ABS, worst


@pytest.mark.parametrize("fs,sec,seed,fp", [(16000, 1.0, 41, 5.0), (48000, 0.6, 42, 5.0), (24000, 0.5, 5006, 1.0),
                                           (8000, 0.5, 43, 5.0)])
def test_synthesis_vs_oracle(wca, port, fs, sec, seed, fp):
    x = make_utterance(fs, sec, seed)
    r = port.pipeline(x, fs, frame_period=fp)
    n = (r["sp"].shape[1] - 1) * 2
    s = wca.Synthesis(fs, n, fp)
    start = 31337
    port.rng_seek(start)
    ref = port.synthesis(r["f0"], r["sp"], r["ap"], fs, fp)
    end = port.rng_position()
    wca.rng_set_position(start)
    y = s.compute(r["f0"], r["sp"], r["ap"])
    assert wca.rng_get_position() == end
    assert np.abs(y - ref).max() < Y_ABS
    port.rng_reset()


def test_synthesis_batch_ragged_and_unvoiced(wca, port):
    fs, n = 16000, 1024
    from oracle.gen_golden import synth_params
    cases = []
    for i, nfr in enumerate((40, 101, 7)):
        f0, sp, ap = synth_params(fs, n, nfr, 600 + i)
        cases.append((f0, sp, ap))
    # an all-unvoiced utterance (the reference divides by max_f0 == 0 here; we define it as noise only)
    f0, sp, ap = synth_params(fs, n, 30, 610)
    cases.append((np.zeros_like(f0), sp, np.full_like(ap, 1.0 - 1e-12)))
    s = wca.Synthesis(fs, n, 5.0)
    start = [0, 10, 0, 5]
    ys, pos = s.compute_batch([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], rng_pos=start)
    for (f0, sp, ap), y, p0, p1 in zip(cases, ys, start, pos):
        port.rng_seek(p0)
        ref = port.synthesis(f0, sp, ap, fs, 5.0)
        assert port.rng_position() == p1
        assert np.abs(y - ref).max() < Y_ABS
    port.rng_reset()


def test_synthesis_short_output_and_errors(wca, port):
    fs, n = 16000, 1024
    from oracle.gen_golden import synth_params
    f0, sp, ap = synth_params(fs, n, 20, 620)
    s = wca.Synthesis(fs, n, 5.0)
    for out_len in (1, 100, 3000):  # shorter than / longer than the contour
        wca.rng_set_position(0)
        port.rng_reset()
        y = s.compute(f0, sp, ap, out_length=out_len)
        assert np.abs(y - port.synthesis(f0, sp, ap, fs, 5.0, out_length=out_len)).max() < Y_ABS
    with pytest.raises(wca.WorldClassError):
        s.compute(f0[:1], sp[:1], ap[:1])
    with pytest.raises(wca.WorldClassError):
        wca.Synthesis(fs, 1000, 5.0)
    port.rng_reset()


@pytest.mark.parametrize("fs", [8000, 16000, 48000])
def test_exact_parallel_phase_accumulation_matches_the_serial_chain(wca, port, fs):
    """The pulse positions hang on the rounding of the reference's sequential phase sum.  The default kernel reproduces
    that sum exactly with integer prefix sums per binade; WC_SYN_TIMEBASE=serial runs the one-wavefront sequential chain.
    Contours chosen to hit the hard cases: all unvoiced (500 Hz: the phase lands on multiples of 2 pi exactly), constant
    F0 values that divide the sampling rate (a constant increment can tie in every step of a binade), a random contour."""
    import os
    from oracle.gen_golden import synth_params
    n_frames, fft = 1601, wca.cheaptrick_fft_size(fs)          # 8 s
    _, sp, ap = synth_params(fs, fft, n_frames, 91)
    rng = np.random.default_rng(5)
    contours = [np.zeros(n_frames), np.full(n_frames, 100.0), np.full(n_frames, 125.0), np.full(n_frames, 250.0),
                np.where(rng.random(n_frames) > 0.4, rng.uniform(60, 700, n_frames), 0.0)]
    ys = {}
    for mode in ("parallel", "serial", "utterance", "single"):
        if mode == "serial":
            os.environ["WC_SYN_TIMEBASE"] = "serial"
        if mode == "single":  # the exact parallel sum by one workgroup per utterance instead of the segment kernels
            os.environ["WC_SYN_PHASE"] = "single"
        if mode == "utterance":  # the pulses picked out of the finished phase by one workgroup per utterance instead of one per tile
            os.environ["WC_SYN_PULSES"] = "utterance"
        try:
            s = wca.Synthesis(fs, fft, 5.0)
        finally:
            os.environ.pop("WC_SYN_TIMEBASE", None)
            os.environ.pop("WC_SYN_PULSES", None)
            os.environ.pop("WC_SYN_PHASE", None)
        out = []
        for f0 in contours:
            wca.rng_set_position(0)
            out.append(s.compute(f0, sp, ap))
        ys[mode] = out
    wca.rng_set_position(0)
    for a, b, c, d in zip(ys["parallel"], ys["serial"], ys["utterance"], ys["single"]):
        assert np.abs(a - b).max() < 1e-9    # a pulse moved by one sample shows up as ~1e-2
        assert np.abs(a - c).max() < 1e-9
        # (at 48 kHz the overlap-add runs in pulse order -- equal phases give equal bits, also against the serial chain; the
        # workgroup-per-pulse kernel of the other rates adds with atomics, in any order)
        assert (np.array_equal(a, d) and np.array_equal(a, b)) if fs == 48000 else np.abs(a - d).max() < 1e-12
    port.rng_reset()
    assert np.abs(ys["parallel"][4] - port.synthesis(contours[4], sp, ap, fs, 5.0)).max() < 1e-8
    port.rng_reset()


def test_aperiodicity_next_to_its_clamp(wca, port):
    """Aperiodicity within 1e-12 of one in the top bands of voiced frames (what D4C's band edge at -1e-12 dB produces): the
    periodic weight 1 - s^2 is 2e-12 there, one ulp of the interpolated s moves it by 1e-4 of itself and the minimum-phase
    transform spreads that over the whole response -- s has to be rounded exactly as the reference rounds it
    (reference src/synthesis.cpp:388-390).  A fused multiply-add in that interpolation cost 3e-8 on the waveform."""
    fs, n, nfr = 16000, 1024, 120
    rng = np.random.default_rng(99)
    bins = n // 2 + 1
    k = np.arange(bins) / (bins - 1.0)
    f0 = np.full(nfr, 354.7)
    f0[40:50] = 0.0
    sp = np.stack([1e-4 + 1e-2 * np.exp(-((k - rng.uniform(0.1, 0.3)) / 0.05) ** 2) for _ in range(nfr)])
    edge = 10.0 ** (-1e-12 / 20.0)
    ap = np.stack([np.interp(k, [0.0, 0.3, 0.6, rng.uniform(0.7, 0.9), 1.0], [0.001, rng.uniform(0.05, 0.4), rng.uniform(0.5, 0.99), 1.0, edge])
                   for _ in range(nfr)])
    ap[40:50] = 1.0 - 1e-12
    port.rng_seek(555)
    ref = port.synthesis(f0, sp, ap, fs, 5.0)
    wca.rng_set_position(555)
    y = wca.Synthesis(fs, n, 5.0).compute(f0, sp, ap)
    assert wca.rng_get_position() == port.rng_position()
    assert np.abs(ref).max() > 0.01
    assert np.abs(y - ref).max() < 1e-12
    port.rng_reset()


@pytest.mark.parametrize("fs,fp,sec", [(16000, 5.0, 1.0), (24000, 1.0, 0.5), (48000, 5.0, 0.5)])
def test_synthesis_wavefront_kernels_against_the_block_kernel_and_beyond_the_row_budget(wca, port, monkeypatch, fs, fp, sec):
    """one wavefront per pulse (N = 1024: syn_pulse_wave8_kernel, eight points per lane; N = 2048: syn_pulse_wave_kernel) with
    ordered overlap-add of response rows; against the workgroup-per-pulse kernel (atomics) on the same input, and with the rows'
    memory budget set to nothing, where the atomics kernel has to take over on its own (an overflow retry of a large batch)."""
    x = make_utterance(fs, sec, 77)
    r = port.pipeline(x, fs, frame_period=fp)
    n = (r["sp"].shape[1] - 1) * 2
    port.rng_seek(4242)
    ref = port.synthesis(r["f0"], r["sp"], r["ap"], fs, fp)
    port.rng_reset()
    ys = []
    for env in ({}, {"WC_SYN_IMPL": "block"}, {"WC_SYN_ROWS_BUDGET_MB": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        wca.rng_set_position(4242)
        ys.append(wca.Synthesis(fs, n, fp).compute(r["f0"], r["sp"], r["ap"]))
        for k in env:
            monkeypatch.delenv(k)
    for y in ys:
        assert np.abs(y - ref).max() < Y_ABS
    assert np.abs(ys[0] - ys[1]).max() < 1e-12 and np.abs(ys[1] - ys[2]).max() < 1e-12
    # rows: the same bits on every run (ordered sums, no atomics)
    wca.rng_set_position(4242)
    again = wca.Synthesis(fs, n, fp).compute(r["f0"], r["sp"], r["ap"])
    assert np.array_equal(again, ys[0])


def test_stage_call_in_two_halves_equals_one_piece_and_retries_only_the_half_that_overflowed(wca, port, monkeypatch):
    """a batch of 16 utterances and more runs as two halves with a twin handle (wc_synthesis.hip: syn_run_device): the same bits as
    one piece (WC_SYN_HALVES=0), the noise-stream positions in and out per utterance, and -- F0 beyond the 960 Hz rate bound in the
    second half only -- a hard-bound retry of that half alone that must leave the first half's waveforms and end positions as
    they were (the retry read the positions the first attempt had already advanced)."""
    fs, n = 16000, 1024
    from oracle.gen_golden import synth_params
    cases = [synth_params(fs, n, 24 + 3 * (i % 5), 700 + i) for i in range(18)]
    start = [1000 * i + 7 for i in range(18)]
    s = wca.Synthesis(fs, n, 5.0)
    arg = lambda cs: ([c[0] for c in cs], [c[1] for c in cs], [c[2] for c in cs])
    ys, pos = s.compute_batch(*arg(cases), rng_pos=start)
    monkeypatch.setenv("WC_SYN_HALVES", "0")
    ys1, pos1 = wca.Synthesis(fs, n, 5.0).compute_batch(*arg(cases), rng_pos=start)
    monkeypatch.delenv("WC_SYN_HALVES")
    assert pos == pos1
    for a, b in zip(ys, ys1):
        assert np.array_equal(a, b)
    for u in (0, 8, 9, 17):
        f0, sp, ap = cases[u]
        port.rng_seek(start[u])
        ref = port.synthesis(f0, sp, ap, fs, 5.0)
        assert port.rng_position() == pos[u]
        assert np.abs(ys[u] - ref).max() < Y_ABS
    # the second half overflows its rate-bounded pulse buffers, the first half does not
    hot = [(np.where(f0 > 0, 1500.0, 0.0), sp, ap) if u in (12, 15) else (f0, sp, ap) for u, (f0, sp, ap) in enumerate(cases)]
    ys2, pos2 = s.compute_batch(*arg(hot), rng_pos=start)
    for u in range(18):
        if u not in (12, 15):
            assert np.array_equal(ys2[u], ys[u]) and pos2[u] == pos[u]
        f0, sp, ap = hot[u]
        one, p1 = wca.Synthesis(fs, n, 5.0).compute_batch([f0], [sp], [ap], rng_pos=[start[u]])
        assert p1[0] == pos2[u]
        assert np.array_equal(one[0], ys2[u])
    port.rng_seek(start[12])
    assert np.abs(ys2[12] - port.synthesis(*hot[12], fs, 5.0)).max() < Y_ABS
    assert port.rng_position() == pos2[12]
    port.rng_reset()
    # a row length other than the handle's is refused, in the mirror and behind it
    with pytest.raises(ValueError):
        s.compute_batch([cases[0][0]], [cases[0][1][:, :-1]], [cases[0][2][:, :-1]])
    import ctypes as C
    assert wca.lib().wc_synthesis_get_fft_size(s._h) == n
    z = (C.c_void_p * 1)()
    one = (C.c_int * 1)(2)
    assert wca.lib().wc_synthesis_compute_batch(s._h, 1, z, one, 2 * n, z, z, one, z, None) == -1
    assert "fft_size" in wca.last_error()
    assert wca.lib().wc_release_scratch() == 0
    ys3 = s.compute_batch(*arg(cases[:3]), rng_pos=start[:3])[0]  # (the staging comes back on the next call)
    assert all(np.array_equal(a, b) for a, b in zip(ys3, ys[:3]))
