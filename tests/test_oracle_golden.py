"""CPU tests: our restatement (oracle/wc_oracle.cpp) against the golden vectors generated from the real
reference (oracle/gen_golden.py), and against the real reference itself when oracle/_ref is present.
These pin the oracle that the -m gpu parity tests then use as the checker."""
import numpy as np
import pytest

from conftest import HARVEST_LONG_CASES, PIPELINE_CASES, check_headline, headline_case, harvest_edge_rows, harvest_long_case, harvest_option_cases, rate96k_case, same_candidates, stage_option_cases
from world_class_amd.synth import make_utterance

# tolerances of the restatement vs the reference (FP64; only the FFT rounding differs)
F0_ABS, SP_REL, AP_ABS, Y_ABS = 1e-9, 1e-9, 1e-10, 1e-9


def test_synth_generator_is_pinned(golden):
    for name in PIPELINE_CASES:
        c = golden.case(name)
        x = make_utterance(c["fs"], c["seconds"], c["seed"])
        assert np.array_equal(x, c["x"])


def test_randn_stream(golden, port):
    port.rng_reset()
    assert np.array_equal(port.randn(4096), golden["randn/first4096"])
    port.rng_seek(1000)
    assert np.array_equal(port.randn(96), golden["randn/first4096"][1000:1096])
    assert port.rng_position() == 1096
    port.rng_reset()


@pytest.mark.parametrize("n", [128, 1024, 2048, 4096])
def test_fft_conventions(golden, port, n):
    x = golden[f"fft/r2c_in_{n}"]
    X = golden[f"fft/r2c_out_{n}"]
    Xc = X[:, 0] + 1j * X[:, 1]
    got = port.fft_r2c(x)
    assert np.abs(got - Xc).max() < 1e-11 * n
    # reference "forward" is e^{+i}: conj of numpy's rfft
    assert np.abs(Xc - np.conj(np.fft.rfft(x))).max() < 1e-10
    assert np.abs(port.fft_c2r(Xc * (1 + 0.5j), n) - golden[f"fft/c2r_out_{n}"]).max() < 1e-10 * n


def test_fft_c2c_and_minimum_phase(golden, port):
    z = golden["fft/c2c_in_1024"]
    z = z[:, 0] + 1j * z[:, 1]
    for s in (1, 2):
        Z = golden[f"fft/c2c_out_1024_sign{s}"]
        assert np.abs(port.fft_c2c(z, s) - (Z[:, 0] + 1j * Z[:, 1])).max() < 1e-10
    M = golden["minphase/out_1024"]
    got = port.minimum_phase(golden["minphase/in_1024"], 1024)
    assert np.abs(got - (M[:, 0] + 1j * M[:, 1])).max() < 1e-12


def test_matlab_helpers(golden, port):
    xs, ys, xi = golden["interp1/x"], golden["interp1/y"], golden["interp1/xi"]
    assert np.array_equal(port.histc(xs, xi), golden["interp1/histc"])
    assert np.array_equal(port.interp1(xs, ys, xi), golden["interp1/yi"])
    assert np.array_equal(port.interp1Q(0.5, 0.25, golden["interp1Q/y"], golden["interp1Q/xi"]), golden["interp1Q/yi"])
    for r in (2, 3, 6, 12):
        assert np.array_equal(port.decimate(golden["decimate/x"], r), golden[f"decimate/y_r{r}"])
    spec = golden["spec/in_1025"]
    assert np.array_equal(port.dc_correction(spec, 200.0, 48000, 2048), golden["spec/dc_f200_48k_2048"])
    assert np.array_equal(port.linear_smoothing(spec, 400.0 / 3.0, 48000, 2048), golden["spec/ls_w133_48k_2048"])
    assert np.array_equal(port.nuttall(769), golden["nuttall/769"])
    for k, v in golden.meta["matlab_round"].items():
        assert port.matlab_round(float(k)) == v
    for k, v in golden.meta["suitable_fft_size"].items():
        assert port.suitable_fft_size(int(k)) == v
    for k, v in golden.meta["cheaptrick_fft_size"].items():
        assert port.cheaptrick_fft_size(int(k)) == v
    for k, v in golden.meta["get_samples"].items():
        fs, n, fp = k.split(":")
        assert port.get_samples(int(fs), int(n), float(fp)) == v


@pytest.mark.parametrize("name", PIPELINE_CASES)
def test_pipeline_against_golden(golden, port, name):
    c = golden.case(name)
    r = port.pipeline(c["x"], c["fs"], harvest_floor=c["harvest_floor"], frame_period=c["frame_period"])
    assert np.array_equal(r["tpos"], c["tpos"])
    assert np.array_equal(r["f0"] == 0, c["f0"] == 0)
    assert np.abs(r["f0"] - c["f0"]).max() < F0_ABS
    s = c["stride"]
    assert (np.abs(r["sp"][::s] - c["sp_rows"]) / c["sp_rows"]).max() < SP_REL
    assert np.abs(r["ap"][::s] - c["ap_rows"]).max() < AP_ABS
    assert (np.abs(r["sp"].sum(1) - c["sp_rowsum"]) / c["sp_rowsum"]).max() < SP_REL
    assert np.abs(r["ap"].sum(1) - c["ap_rowsum"]).max() < AP_ABS * r["ap"].shape[1]
    assert np.abs(r["y"] - c["y"]).max() < Y_ABS


def test_synthesis_only_against_golden(golden, port):
    from oracle.gen_golden import synth_params
    m = golden.meta["synth_only"]
    f0, sp, ap = synth_params(m["fs"], m["fft_size"], m["n_frames"], m["seed"])
    port.rng_reset()
    y = port.synthesis(f0, sp, ap, m["fs"], m["frame_period"])
    assert np.abs(y - golden["synth_only/y"]).max() < Y_ABS


def test_threaded_oracle_is_bit_identical_to_serial(golden, port):
    c = golden.case("m48k_1s")
    a = port.pipeline(c["x"], c["fs"])
    port.set_threads(4)
    try:
        b = port.pipeline(c["x"], c["fs"])
    finally:
        port.set_threads(0)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_draw_count_contract(golden, port):
    c = golden.case("c1_16k_2s_floor71")
    port.rng_reset()
    port.cheaptrick(c["x"], c["fs"], c["tpos"], c["f0"])
    assert port.rng_position() == port.cheaptrick_draws(c["fs"], c["f0"])


def test_port_against_live_reference(golden, port):
    """When the real reference is built here (oracle/_ref), check a case that is NOT in the goldens."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built in this environment")
    x = make_utterance(16000, 1.0, 1234)
    r = ref.run_fresh("pipeline", x, 16000, harvest_floor=40.0)
    p = port.pipeline(x, 16000, harvest_floor=40.0)
    assert np.abs(r["f0"] - p["f0"]).max() < F0_ABS
    assert (np.abs(r["sp"] - p["sp"]) / r["sp"]).max() < SP_REL
    assert np.abs(r["ap"] - p["ap"]).max() < AP_ABS
    assert np.abs(r["y"] - p["y"]).max() < Y_ABS


@pytest.mark.parametrize("name", HARVEST_LONG_CASES)
def test_harvest_long_utterances_against_golden(port, name):
    """whole 10 s utterances; "tie_*" is the one whose contour hangs on std::sort's order of voiced sections that
    start on the same frame (reference src/harvest.cpp:508-517)"""
    x, fs, floor, f0 = harvest_long_case(name)
    port.set_threads(4)
    try:
        _, got = port.harvest(x, fs, f0_floor=floor)
    finally:
        port.set_threads(0)
    assert np.array_equal(got == 0, f0 == 0)
    assert np.abs(got - f0).max() < F0_ABS


def test_unreliable_candidates_edge_rows(port):
    """removeUnreliableCandidates compares frames 1 and L-2 with rows the reference never wrote (reference
    src/harvest.cpp:714-715; zero in the oracle's build of it): the lower voice at frame L-2 is matched by frame L-1 only
    and has to go"""
    x, fs, floor, _ = harvest_long_case("edge_rows_16k_3s_duet")
    rows, cand = harvest_edge_rows()
    d = port.harvest_debug(x, fs, f0_floor=floor)
    for r, c in zip(rows, cand):
        assert same_candidates(d["cand"][r], c)
    assert (d["cand"][rows[1]] != 0).sum() == 7


def test_harvest_stages_against_live_reference(port):
    """stage by stage against the real reference's own member functions (oracle/ref_harvest_taps.cpp), when it is built"""
    from oracle import ref
    if not ref.taps_available():
        pytest.skip("oracle/_ref/libworld_ref_taps.so not built (no reference sources here)")
    fs = 16000
    x = make_utterance(fs, 1.5, 4242)
    for floor in (71.0, 40.0):
        a, b = port.harvest_debug(x, fs, f0_floor=floor), ref.harvest_taps(x, fs, f0_floor=floor)
        assert a["n_cand"] == b["n_cand"] and a["cand"].shape == b["cand"].shape
        assert np.array_equal(a["y"], b["y"])
        assert np.array_equal(a["raw"] == 0, b["raw"] == 0) and np.abs(a["raw"] - b["raw"]).max() < 1e-7
        assert np.array_equal(a["cand"] == 0, b["cand"] == 0) and np.abs(a["cand"] - b["cand"]).max() < 1e-9
        assert (np.abs(a["score"] - b["score"]) / np.maximum(b["score"], 1.0)).max() < 1e-6
        for k in ("f0_base", "f0_fixed", "f0_1ms"):
            assert np.array_equal(a[k] == 0, b[k] == 0) and np.abs(a[k] - b[k]).max() < 1e-9


def test_harvest_options_against_golden(port):
    """target_fs, channels_in_octave and use_cos_table (reference include/harvest.hpp:16-24): the restatement against contours
    of the real reference; the cosine table moves the contour by 0.01 Hz, which the fixture also shows"""
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "harvest_options.npz"))
    assert 1e-3 < np.abs(z["table_16k/f0"] - z["table_16k/f0_exact_cosines"]).max() < 0.1
    for name, x, fs, opts, f0 in harvest_option_cases():
        o = dict(opts)
        kw = {k: o.pop(k) for k in ("f0_floor", "frame_period") if k in o}
        port.set_harvest_options(**o)
        try:
            _, got = port.harvest(x, fs, **kw)
        finally:
            port.set_harvest_options()
        assert np.array_equal(got == 0, f0 == 0), name
        assert np.abs(got - f0).max() < F0_ABS, name


def test_cheaptrick_and_d4c_options_against_golden(port):
    """q1, f0_floor, fft_size (reference include/cheaptrick.hpp) and the D4C threshold (include/d4c.hpp) away from their defaults"""
    x, fs, tpos, f0, stride, ct, d4 = stage_option_cases()
    assert np.array_equal(port.harvest(x, fs)[0], tpos)
    for name, kw, rows, rowsum in ct:
        port.rng_reset()
        sp = port.cheaptrick(x, fs, tpos, f0, **kw)
        assert sp.shape[1] == rows.shape[1], name
        assert (np.abs(sp[::stride] - rows) / rows).max() < SP_REL and (np.abs(sp.sum(axis=1) - rowsum) / rowsum).max() < SP_REL, name
    for name, thr, rows, rowsum in d4:
        port.rng_reset()
        ap = port.d4c(x, fs, tpos, f0, 1024, threshold=thr)
        assert np.abs(ap[::stride] - rows).max() < AP_ABS and np.abs(ap.sum(axis=1) - rowsum).max() < AP_ABS * 1024, name
    port.rng_reset()


def test_pipeline_at_96_khz_against_golden(port):
    """decimation ratio 12, 4096-point CheapTrick / Synthesis, 8192-point D4C and LoveTrain"""
    x, fs, stride, z = rate96k_case()
    port.set_threads(4)
    try:
        r = port.pipeline(x, fs)
    finally:
        port.set_threads(0)
    assert np.array_equal(r["tpos"], z["tpos"]) and np.array_equal(r["f0"] == 0, z["f0"] == 0)
    assert np.abs(r["f0"] - z["f0"]).max() < F0_ABS
    assert (np.abs(r["sp"][::stride] - z["sp_rows"]) / z["sp_rows"]).max() < SP_REL
    assert (np.abs(r["sp"].sum(axis=1) - z["sp_rowsum"]) / z["sp_rowsum"]).max() < SP_REL
    assert np.abs(r["ap"][::stride] - z["ap_rows"]).max() < AP_ABS
    assert np.abs(r["y"] - z["y"]).max() < Y_ABS


def test_headline_utterance_against_golden(port):
    """the benchmark's utterance size (48 kHz, 10 s): the restatement against the real reference on every frame"""
    x, g = headline_case(0)
    port.set_threads(8)
    try:
        r = port.pipeline(x, g["fs"])
    finally:
        port.set_threads(0)
    check_headline(r, g, F0_ABS, SP_REL, AP_ABS, Y_ABS)


def test_config2_utterance_against_golden(port):
    """BASELINE config 2's utterance size (16 kHz, 10 s): the restatement against the real reference on every frame"""
    x, g = headline_case(3, "config2_16k_10s.npz")
    port.set_threads(8)
    try:
        r = port.pipeline(x, g["fs"])
    finally:
        port.set_threads(0)
    check_headline(r, g, F0_ABS, SP_REL, AP_ABS, Y_ABS)


def test_device_argsort_reproduces_std_sort(tmp_path):
    """world_class_amd/csrc/wc_argsort.hpp (what hv_contour_kernel runs) against std::sort of the library the reference is
    built with, on tie-heavy keys and on a sequence that drives introsort into its heap sort"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "argsort_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "argsort_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_restatement_at_rates_without_decimation_and_where_it_stops_being_a_checker(port):
    """8 kHz and 11.025 kHz input is not decimated (reference src/harvest.cpp:217-219).  On a signal with a noise floor the
    restatement follows the live reference there as everywhere else; on an undithered impulse train -- exact zeros between the
    pulses in the upper bands, where the reference's band-passed signal is the rounding noise of ITS FFT convolution -- it does not
    (DESIGN.md section 7 (iv)): the GPU sweeps at these rates are checked against oracle/_ref itself."""
    from oracle import ref
    from world_class_amd.synth import make_signal, make_utterance
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs the reference sources at build time)")
    for fs in (8000, 11025):
        x = make_utterance(fs, 1.5, 8800 + fs)
        f_ref = ref.run_fresh("harvest", x, fs)[1]
        f_port = port.harvest(x, fs)[1]
        assert np.array_equal(f_ref == 0, f_port == 0) and np.abs(f_ref - f_port).max() < 1e-9, fs
    x = make_signal(8000, 3.0, 1540043)  # impulses, period 90 samples
    f_ref = ref.run_fresh("harvest", x, 8000)[1]
    f_port = port.harvest(x, 8000)[1]
    assert int((f_ref > 0).sum()) > 500
    # (documented, not required: should the restatement ever agree here, the sentence above wants rewriting)
    if np.array_equal(f_ref == 0, f_port == 0):
        pytest.xfail("the restatement agrees with the reference on the 8 kHz impulse train: update DESIGN.md section 7 (iv)")
