"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares,
the Python mirror binds exactly those symbols, the product never touches the oracle, and everything fails
loudly without a HIP device (no CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "world_class_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wc_[a-z0-9_]+)\s*\(", src)))


def io_header_symbols(name="world_class_io.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from world_class_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], check=True, stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (wc_[a-z0-9_]+)", out))
    declared = header_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    # the data-format header: the reference's own function names plus the wc_ extensions
    exported_all = set(re.findall(r" T ([A-Za-z_][A-Za-z0-9_]*)", out))
    io_declared = io_header_symbols()
    assert {"wavread", "wavwrite", "GetAudioLength", "ReadF0", "WriteSpectralEnvelope", "wc_modify_parameters_device"} <= set(io_declared)
    assert not [s for s in io_declared if s not in exported_all]
    helper_declared = [n for n in io_header_symbols("world_matlabfunctions.hpp") if not n.startswith(("My", "GetSafe"))]
    assert {"interp1", "decimate", "randn", "DCCorrection"} <= set(helper_declared)
    assert not [s for s in helper_declared if s not in exported_all]
    fft_declared = io_header_symbols("world_fft.hpp")
    assert {"fft_plan_dft_r2c_1d", "fft_execute", "fft_destroy_plan"} <= set(fft_declared)
    assert not [s for s in fft_declared if s not in exported_all]
    stream_declared = [n for n in io_header_symbols("world_class_stream.h") if n.startswith("wc_")]
    assert {"wc_stream_create", "wc_stream_push_device", "wc_stream_reset"} <= set(stream_declared)
    assert not [s for s in stream_declared if s not in exported_all]
    shard_declared = [n for n in io_header_symbols("world_class_shard.h") if n.startswith("wc_")]
    assert set(shard_declared) == {"wc_shard_partition", "wc_gather_device", "wc_gather_to_root_device"}
    assert not [s for s in shard_declared if s not in exported_all]
    codec_declared = io_header_symbols("world_class_codec.h")
    assert {"CodeSpectralEnvelope", "DecodeAperiodicity", "GetNumberOfAperiodicities"} <= set(codec_declared)
    assert not [s for s in codec_declared if s not in exported_all]


def test_python_mirror_binds_the_header(built_lib):
    import world_class_amd as w
    assert sorted(w.EXPORTED_SYMBOLS) == header_symbols()
    from world_class_amd import io as wio
    assert sorted(wio.IO_SIGNATURES) == io_header_symbols()
    from world_class_amd import codec
    assert sorted(codec.CODEC_SIGNATURES) == io_header_symbols("world_class_codec.h")
    from world_class_amd import stream
    assert sorted(stream.STREAM_SIGNATURES) == [n for n in io_header_symbols("world_class_stream.h") if n.startswith("wc_")]
    lib = w.lib()  # loads and sets every prototype
    assert lib.wc_version().startswith(b"world_class_amd")
    # pure host helpers work without a device and match the reference's formulas (goldens in test_oracle_golden)
    assert w.get_samples(48000, 480000, 5.0) == 2001
    assert w.cheaptrick_fft_size(48000) == 2048 and w.cheaptrick_fft_size(16000) == 1024
    assert w.synthesis_out_length(2001, 5.0, 48000) == 480001


def test_no_cpu_fallback_without_device(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    import world_class_amd as w
    for make in (lambda: w.Harvest(16000), lambda: w.CheapTrick(16000), lambda: w.D4C(16000),
                 lambda: w.Synthesis(16000, 1024, 5.0)):
        with pytest.raises(w.WorldClassError, match="no usable HIP device"):
            make()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "world_class_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the CPU oracle", ""), f
                assert "/root/reference" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read(), f


def test_cpp_dropin_headers_compile_with_plain_gxx(built_lib, tmp_path):
    exe = tmp_path / "demo"
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "demo.cpp"), "-o", str(exe),
           "-L" + os.path.dirname(built_lib), "-lworldclass_hip", "-Wl,-rpath," + os.path.dirname(built_lib)]
    subprocess.run(cmd, check=True)
    assert exe.exists()


def test_constant_and_macro_headers_of_the_reference_are_shipped(tmp_path):
    """include/world_constantnumbers.hpp and include/macrodefinitions.hpp (reference include/world_constantnumbers.hpp:1-44,
    include/macrodefinitions.hpp:1-144): a caller that includes either compiles against include/, as C++11 and -- the macro
    header -- as C; the constants carry the reference's values."""
    src = tmp_path / "consts.cpp"
    src.write_text('''#include "macrodefinitions.hpp"
#include "world_constantnumbers.hpp"
#include "harvest.hpp"
#include <cstdio>
WORLD_BEGIN_C_DECLS
WORLD_API int exported_by_a_caller(void) { return world::kHanning + world::kBlackman; }
WORLD_END_C_DECLS
static_assert(world::kFloorF0 == 71.0 && world::kCeilF0 == 800.0 && world::kDefaultF0 == 500.0, "F0 range");
static_assert(world::kFrequencyInterval == 3000.0 && world::kUpperLimit == 15000.0 && world::kThreshold == 0.85 && world::kFloorF0D4C == 47.0, "D4C");
static_assert(world::kM0 == 1127.01048 && world::kF0 == 700.0 && world::kFloorFrequency == 40.0 && world::kCeilFrequency == 20000.0, "codec");
static_assert(world::kMySafeGuardMinimum == 1e-12 && world::kMaximumValue == 100000.0, "guards");
int main() {
    std::printf("%.20g %.20g %.20g %d\\n", world::kPi, world::kEps, world::kLog2, exported_by_a_caller());
    return 0;
}
''')
    exe = tmp_path / "consts"
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert float(out[0]) == 3.1415926535897932384 and float(out[1]) == 2.0 ** -52 and float(out[2]) == 0.69314718055994529 and out[3] == "3"
    csrc = tmp_path / "m.c"
    csrc.write_text('#include "macrodefinitions.hpp"\nWORLD_BEGIN_C_DECLS\nWORLD_API int f(void);\nWORLD_END_C_DECLS\nint f(void) { return 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-DWORLD_LIBRARIES_EXPORTS", "-DWORLD_SRC", "-I" + os.path.join(ROOT, "include"), "-c", str(csrc),
                    "-o", str(tmp_path / "m.o")], check=True)
