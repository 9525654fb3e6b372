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
