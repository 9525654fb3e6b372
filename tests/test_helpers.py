"""the reference's free helper functions (include/world_matlabfunctions.hpp, world_common.hpp) as
host functions of the product library, against the golden vectors produced by the real reference (no GPU needed)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def h():
    from world_class_amd import build
    build.build()
    from world_class_amd import helpers
    return helpers


def test_matlab_helpers_match_the_reference(golden, h):
    xs, ys, xi = golden["interp1/x"], golden["interp1/y"], golden["interp1/xi"]
    assert np.array_equal(h.histc(xs, xi), golden["interp1/histc"])
    assert np.array_equal(h.interp1(xs, ys, xi), golden["interp1/yi"])
    assert np.array_equal(h.interp1Q(0.5, 0.25, golden["interp1Q/y"], golden["interp1Q/xi"]), golden["interp1Q/yi"])
    for r in (2, 3, 6, 12):
        assert np.array_equal(h.decimate(golden["decimate/x"], r), golden[f"decimate/y_r{r}"])
    spec = golden["spec/in_1025"]
    assert np.array_equal(h.dc_correction(spec, 200.0, 48000, 2048), golden["spec/dc_f200_48k_2048"])
    assert np.array_equal(h.linear_smoothing(spec, 400.0 / 3.0, 48000, 2048), golden["spec/ls_w133_48k_2048"])
    assert np.array_equal(h.nuttall_window(769), golden["nuttall/769"])
    for k, v in golden.meta["matlab_round"].items():
        assert h.matlab_round(float(k)) == v
    for k, v in golden.meta["suitable_fft_size"].items():
        assert h.suitable_fft_size(int(k)) == v
    x = np.arange(10.0)
    assert np.array_equal(h.fftshift(x), np.concatenate([x[5:], x[:5]]))
    assert np.array_equal(h.diff(x ** 2), np.diff(x ** 2))
    assert abs(h.matlab_std(x) - np.std(x, ddof=1)) < 1e-15


def test_randn_is_the_process_wide_stream(golden, h):
    import world_class_amd as w
    w.rng_set_position(0)
    assert np.array_equal(h.randn(64), golden["randn/first4096"][:64])
    assert w.rng_get_position() == 64
    w.rng_set_position(1000)                       # a jump, as after a stage consumed draws
    assert np.array_equal(h.randn(96), golden["randn/first4096"][1000:1096])
    w.rng_set_position(0)


def _fft_api():
    import ctypes as C
    from world_class_amd import lib

    class Plan(C.Structure):
        _fields_ = [("n", C.c_int), ("sign", C.c_int), ("flags", C.c_uint), ("c_in", C.c_void_p), ("in_", C.c_void_p),
                    ("c_out", C.c_void_p), ("out", C.c_void_p), ("input", C.c_void_p), ("ip", C.c_void_p), ("w", C.c_void_p)]
    L = lib()
    L.fft_plan_dft_1d.restype = Plan
    L.fft_plan_dft_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    L.fft_plan_dft_r2c_1d.restype = Plan
    L.fft_plan_dft_r2c_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
    L.fft_plan_dft_c2r_1d.restype = Plan
    L.fft_plan_dft_c2r_1d.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
    L.fft_execute.argtypes = [Plan]
    L.fft_destroy_plan.argtypes = [Plan]
    return L


def test_fft_plan_api_conventions_match_the_reference(golden, h):
    """include/world_fft.hpp: r2c = e^{+i}, c2r unnormalised inverse, c2c FORWARD e^{+i} / BACKWARD e^{-i} (goldens from the
    reference's own FFT; rounding differs at the 1e-13 level between the two factorisations)"""
    L = _fft_api()
    for n in (128, 1024, 2048, 4096):
        x = np.ascontiguousarray(golden[f"fft/r2c_in_{n}"])
        X = np.zeros((n // 2 + 1, 2))
        p = L.fft_plan_dft_r2c_1d(n, x.ctypes.data, X.ctypes.data, 3)
        L.fft_execute(p)
        L.fft_destroy_plan(p)
        assert np.abs(X - golden[f"fft/r2c_out_{n}"]).max() < 1e-10 and X[0, 1] == 0.0 and X[n // 2, 1] == 0.0
        y = np.zeros(n)
        Xc = (golden[f"fft/r2c_out_{n}"][:, 0] + 1j * golden[f"fft/r2c_out_{n}"][:, 1]) * (1 + 0.5j)  # the golden's input
        Xin = np.ascontiguousarray(np.stack([Xc.real, Xc.imag], 1))
        p = L.fft_plan_dft_c2r_1d(n, Xin.ctypes.data, y.ctypes.data, 3)
        L.fft_execute(p)
        L.fft_destroy_plan(p)
        assert np.abs(y - golden[f"fft/c2r_out_{n}"]).max() < 1e-10 * n
    z = np.ascontiguousarray(golden["fft/c2c_in_1024"])
    for sign in (1, 2):
        Z = np.zeros_like(z)
        p = L.fft_plan_dft_1d(1024, z.ctypes.data, Z.ctypes.data, sign, 3)
        L.fft_execute(p)
        L.fft_destroy_plan(p)
        assert np.abs(Z - golden[f"fft/c2c_out_1024_sign{sign}"]).max() < 1e-10


def test_world_common_structs_compile_and_run(golden, tmp_path):
    """include/world_common.hpp: MinimumPhaseAnalysis / ForwardRealFFT / InverseRealFFT used the way the reference's callers use
    them, compiled with plain g++ against the product library (host-only code: runs without a GPU); the minimum-phase spectrum
    against the real reference's (minphase/* goldens), a forward / inverse real transform pair against numpy"""
    import os
    import subprocess
    from world_class_amd import build
    lib = build.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include "world_common.hpp"
int main(int argc, char **argv) {
	const int n = 1024;
	MinimumPhaseAnalysis m;
	m.initialize(n);
	FILE *f = fopen(argv[1], "rb");
	if (fread(m.log_spectrum, 8, n / 2 + 1, f) != (size_t)(n / 2 + 1)) return 2;
	fclose(f);
	m.compute();
	f = fopen(argv[2], "wb");
	fwrite(m.minimum_phase_spectrum, 16, n / 2 + 1, f);
	fclose(f);
	m.destroy();
	// the real transform pair: a circular convolution of two short sequences through the spectra
	ForwardRealFFT fw; InverseRealFFT iv;
	fw.initialize(16); iv.initialize(16);
	const double x[4] = {1, 2, 3, 4}, hh[3] = {1, -1, 0.5};
	double xs[9][2];
	for (int i = 0; i < 16; ++i) fw.waveform[i] = i < 4 ? x[i] : 0.0;
	fft_execute(fw.forward_fft);
	for (int i = 0; i <= 8; ++i) { xs[i][0] = fw.spectrum[i][0]; xs[i][1] = fw.spectrum[i][1]; }
	for (int i = 0; i < 16; ++i) fw.waveform[i] = i < 3 ? hh[i] : 0.0;
	fft_execute(fw.forward_fft);
	for (int i = 0; i <= 8; ++i) {
		iv.spectrum[i][0] = xs[i][0] * fw.spectrum[i][0] - xs[i][1] * fw.spectrum[i][1];
		iv.spectrum[i][1] = xs[i][0] * fw.spectrum[i][1] + xs[i][1] * fw.spectrum[i][0];
	}
	fft_execute(iv.inverse_fft);
	for (int i = 0; i < 6; ++i) printf("%.12f\n", iv.waveform[i] / 16);  // (the inverse is unnormalised)
	fw.destroy(); iv.destroy();
	return 0;
}
''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L" + os.path.dirname(lib), "-lworldclass_hip", "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    inp = tmp_path / "in.f64"
    np.ascontiguousarray(golden["minphase/in_1024"][:513]).tofile(inp)
    out = subprocess.run([str(exe), str(inp), str(tmp_path / "out.f64")], check=True, stdout=subprocess.PIPE, text=True).stdout
    got = np.fromfile(tmp_path / "out.f64").reshape(-1, 2)
    assert np.abs(got - golden["minphase/out_1024"][:513]).max() < 1e-12
    conv = np.array([float(v) for v in out.split()])
    assert np.abs(conv - np.convolve([1, 2, 3, 4], [1, -1, 0.5])).max() < 1e-9


def test_host_copy_threads_hand_over_fork_and_serve_two_callers(tmp_path):
    """The host-pointer entry points move rows between the caller's arrays and page-locked staging with a set of threads that is
    started once per process and woken per copy (world_class_amd/csrc/wc_hostcopy.hpp, round 6).  Host code only: a hundred
    scatters in a row, rows that do not lie one behind the other, a forked child (it has none of the parent's threads) and two
    calling threads at once -- tests/cpp/hostcopy_pool.cpp, built against the library as a host program."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "world_class_amd")
    if not os.path.exists(os.path.join(lib_dir, "libworldclass_hip.so")) or not shutil.which("hipcc"):
        pytest.skip("needs the built library and hipcc")
    exe = str(tmp_path / "hostcopy_pool")
    subprocess.run(["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-pthread", "-w", "-I" + os.path.join(lib_dir, "csrc"),
                    "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "hostcopy_pool.cpp"), "-L" + lib_dir, "-lworldclass_hip",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True, timeout=300).stdout
    assert out.strip().endswith("ok"), out

