"""the reference's WAV and parameter-file formats (host code of libworldclass_hip.so, no GPU
needed).  Fixtures in tests/golden/io/ were written by the reference's own tools (oracle/gen_golden_io.py): our readers
must return what the reference's readers returned, our writers must produce the same bytes."""
import filecmp
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IO = os.path.join(ROOT, "tests", "golden", "io")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(IO, "io_golden.npz"))


@pytest.fixture(scope="module")
def wio():
    from world_class_amd import build
    build.build()
    from world_class_amd import io
    return io


def test_wavread_matches_the_reference_reader(g, wio):
    x, fs, nbit = wio.wavread(os.path.join(IO, "ref_written_16k.wav"))
    assert (len(x), fs, nbit) == (int(g["wav_read_len"]), int(g["wav_read_fs"]), int(g["wav_read_nbit"]))
    assert np.array_equal(x, g["wav_read_x"])
    for name in ("list_chunk_16bit", "pcm_24bit", "pcm_8bit", "pcm_32bit"):
        x, fs, nbit = wio.wavread(os.path.join(IO, name + ".wav"))
        assert (len(x), fs, nbit) == (int(g[name + "_len"]), int(g[name + "_fs"]), int(g[name + "_nbit"])), name
        assert np.array_equal(x, g[name + "_x"]), name


def test_rejected_and_missing_files_behave_like_the_reference(g, wio):
    for name in ("bad_stereo", "bad_fmt_size", "bad_riff"):
        assert wio.audio_length(os.path.join(IO, name + ".wav")) == int(g[name + "_len"]) == -1
        # wavread leaves its outputs untouched (the generator pre-set fs = nbit = -1)
        assert int(g[name + "_fs"]) == -1 and int(g[name + "_nbit"]) == -1
        with pytest.raises(wio.WorldClassError):
            wio.wavread(os.path.join(IO, name + ".wav"))
    assert wio.audio_length(os.path.join(IO, "does_not_exist.wav")) == int(g["missing_len"]) == 0


def test_sample_sizes_other_than_1_to_4_bytes_are_rejected(wio):
    # the header's bits-per-sample is a free byte; 40 would index past a 4-byte sample scratch, 12 is not whole bytes
    for name in ("bad_nbit", "bad_nbit12"):
        assert wio.audio_length(os.path.join(IO, name + ".wav")) == -1
        with pytest.raises(wio.WorldClassError):
            wio.wavread(os.path.join(IO, name + ".wav"))


def test_wavwrite_is_byte_identical(g, wio, tmp_path):
    out = tmp_path / "ours.wav"
    wio.wavwrite(g["wav_x"], 16000, out)
    assert filecmp.cmp(out, os.path.join(IO, "ref_written_16k.wav"), shallow=False)
    from oracle import port_io
    pcm, fs = wio.wavread_pcm16(out)
    assert fs == 16000 and np.array_equal(pcm, port_io.pcm16_of(g["wav_x"]))
    assert np.array_equal(pcm / 32768.0, g["wav_read_x"])  # the device path's scaling equals wavread's


def test_parameter_files_are_byte_identical_and_read_back(g, wio, tmp_path):
    wio.write_f0(tmp_path / "o.f0", g["par_tpos"], g["par_f0"], 5.0)
    wio.write_f0(tmp_path / "o.txt", g["par_tpos"], g["par_f0"], 5.0, text=True)
    wio.write_spectral_envelope(tmp_path / "o.sp", g["par_sp"], 16000, 5.0, 64)
    wio.write_aperiodicity(tmp_path / "o.ap", g["par_ap"], 16000, 5.0, 64)
    wio.write_spectral_envelope(tmp_path / "o20.sp", g["par_sp"], 16000, 5.0, 64, number_of_dimensions=20)
    for ours, ref in (("o.f0", "ref.f0"), ("o.txt", "ref_f0.txt"), ("o.sp", "ref.sp"), ("o.ap", "ref.ap"), ("o20.sp", "ref_nod20.sp")):
        assert filecmp.cmp(tmp_path / ours, os.path.join(IO, ref), shallow=False), ref
    t, f = wio.read_f0(os.path.join(IO, "ref.f0"))
    assert np.array_equal(t, g["read_f0_tpos"]) and np.array_equal(f, g["read_f0"])
    assert np.array_equal(wio.read_spectral_envelope(os.path.join(IO, "ref.sp")), g["read_sp"])
    assert np.array_equal(wio.read_aperiodicity(os.path.join(IO, "ref.ap")), g["read_ap"])
    assert np.array_equal(wio.read_spectral_envelope(os.path.join(IO, "ref_nod20.sp")), g["read_sp20"])
    for key, val in zip(g["hdr_keys"], g["hdr_vals"]):
        fname, tag = str(key).split("|")
        assert wio.header_information(os.path.join(IO, fname), tag) == val, key
    with pytest.raises(wio.WorldClassError):
        wio.read_spectral_envelope(os.path.join(IO, "ref.ap"))  # wrong magic: "Header error."


def test_numpy_restatement_of_the_modification_is_pinned_to_the_reference(g):
    from oracle import port_io
    fs, fft = int(g["mod_fs"]), int(g["mod_fft"])
    for tag in ("scale_only", "up", "down", "down_small"):
        n_args, shift, ratio = g[f"mod_{tag}_args"]
        f, s = port_io.parameter_modification(fs, fft, g["mod_f0"], g["mod_sp"], shift, ratio if n_args >= 2 else None)
        assert np.array_equal(f, g[f"mod_{tag}_f0"]), tag
        assert np.abs(s / g[f"mod_{tag}_sp"] - 1).max() < 1e-13, tag
