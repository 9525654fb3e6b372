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
