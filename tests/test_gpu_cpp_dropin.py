"""-m gpu: the reference demo's call sequence compiled against the drop-in C++ headers (tests/cpp/demo.cpp)
reproduces the golden outputs of the real reference for the demo's own option set (Harvest f0_floor 40)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_demo_matches_reference_golden(golden, tmp_path):
    from world_class_amd import build
    lib = build.build()
    exe = tmp_path / "demo"
    subprocess.run(["g++", "-std=c++11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "demo.cpp"),
                    "-o", str(exe), "-L" + os.path.dirname(lib), "-lworldclass_hip", "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    c = golden.case("c1_16k_2s_floor40")
    inp = tmp_path / "x.f64"
    c["x"].tofile(inp)
    subprocess.run([str(exe), str(inp), str(c["fs"]), str(tmp_path / "out")], check=True)
    f0 = np.fromfile(tmp_path / "out.f0.f64")
    bins = c["fft_size"] // 2 + 1
    sp = np.fromfile(tmp_path / "out.sp.f64").reshape(-1, bins)
    ap = np.fromfile(tmp_path / "out.ap.f64").reshape(-1, bins)
    y = np.fromfile(tmp_path / "out.y.f64")
    s = c["stride"]
    assert np.array_equal(f0 == 0, c["f0"] == 0) and np.abs(f0 - c["f0"]).max() < 1e-6
    assert (np.abs(sp[::s] - c["sp_rows"]) / c["sp_rows"]).max() < 1e-7
    assert np.abs(ap[::s] - c["ap_rows"]).max() < 1e-7
    assert np.abs(y - c["y"]).max() < 1e-8


def test_cpp_file_level_demo(golden, port, tmp_path):
    """WAV in -> analysis -> parameter files, coded envelope, resynthesised WAV, through the drop-in tool / codec headers;
    checked against the oracle fed with the same quantised samples."""
    from oracle import port_codec, port_io
    from world_class_amd import build, io as wio
    lib = build.build()
    exe = tmp_path / "wavdemo"
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "wavdemo.cpp"), "-o", str(exe), "-L" + os.path.dirname(lib),
                    "-lworldclass_hip", "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    c = golden.case("c1_16k_2s_floor71")
    fs = int(c["fs"])
    wio.wavwrite(c["x"][:fs], fs, tmp_path / "in.wav")      # 1 s; the WAV holds clamp(int(x * 32767)) / 32768
    subprocess.run([str(exe), str(tmp_path / "in.wav"), str(tmp_path / "out")], check=True)
    x = port_io.pcm16_of(c["x"][:fs]) / 32768.0
    port.rng_reset()
    r = port.pipeline(x, fs)
    port.rng_reset()
    t, f0 = wio.read_f0(tmp_path / "out.f0")
    # ReadF0 rebuilds the time axis as i / 1000.0 * frame_period (reference tools/parameterio.cpp:105-106), an ulp off Harvest's
    assert np.array_equal(t, np.arange(len(f0)) / 1000.0 * 5.0) and np.abs(t - r["tpos"]).max() < 1e-15
    assert np.array_equal(f0 == 0, r["f0"] == 0) and np.abs(f0 - r["f0"]).max() < 1e-6
    sp, ap = wio.read_spectral_envelope(tmp_path / "out.sp"), wio.read_aperiodicity(tmp_path / "out.ap")
    assert np.abs(sp / r["sp"] - 1).max() < 1e-7 and np.abs(ap - r["ap"]).max() < 1e-7
    mcep = wio.read_spectral_envelope(tmp_path / "out.mcep")
    assert mcep.shape == (len(f0), 40) and np.abs(mcep - port_codec.code_spectral_envelope(r["sp"], fs, 1024, 40)).max() < 1e-6
    y, fs_y, nbit = wio.wavread(tmp_path / "out.wav")
    assert (fs_y, nbit) == (fs, 16)
    assert np.abs(y * 32768 - port_io.pcm16_of(r["y"])).max() <= 1  # a sample within 1e-8 of a quantisation step may flip


def test_reference_demo_unchanged_on_the_product(golden, port, tmp_path):
    """oracle/_ref/ref_demo_on_product is the reference's own test/test.cpp, compiled unchanged against this repository's
    include/ and linked to libworldclass_hip.so (oracle/Makefile; built where the reference sources exist).  Run with the
    demo's F0 and spectral factors and compared with the oracle: CPU analysis (Harvest floor 40 like the demo), numpy
    modification, CPU synthesis, wavwrite quantisation."""
    from oracle import port_io
    from world_class_amd import io as wio
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_demo_on_product")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_demo_on_product not built (needs the reference sources at build time)")
    c = golden.case("c1_16k_2s_floor40")
    fs = int(c["fs"])
    wio.wavwrite(c["x"][:fs], fs, tmp_path / "in.wav")
    subprocess.run([exe, str(tmp_path / "in.wav"), str(tmp_path / "out"), "1.2", "0.9"], check=True, cwd=tmp_path,
                   stdout=subprocess.DEVNULL)
    x = port_io.pcm16_of(c["x"][:fs]) / 32768.0
    port.rng_reset()
    tpos, f0 = port.harvest(x, fs, f0_floor=40.0)
    sp = port.cheaptrick(x, fs, tpos, f0)
    ap = port.d4c(x, fs, tpos, f0, 1024)
    f0m, spm = port_io.parameter_modification(fs, 1024, f0, sp, 1.2, 0.9)
    y_ref = port.synthesis(f0m, spm, ap, fs, 5.0)
    port.rng_reset()
    y, fs_y, nbit = wio.wavread(tmp_path / "out_1.wav")
    assert (fs_y, nbit, len(y)) == (fs, 16, len(y_ref))
    assert np.abs(y * 32768 - port_io.pcm16_of(y_ref)).max() <= 1  # a sample within 1e-8 of a quantisation step may flip
