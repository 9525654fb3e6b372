#!/usr/bin/env python
"""Code bytes of every kernel in libworldclass_hip.so (symbol sizes of the embedded gfx950 code objects); the instruction
cache is 64 KB per two CUs: python tools/kernel_code_size.py [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("WC_LIB_PATH") or os.path.join(ROOT, "world_class_amd", "libworldclass_hip.so")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
with tempfile.TemporaryDirectory() as d:
    fb = os.path.join(d, "fb")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True)
    blob = open(fb, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    for n, st in enumerate(starts):
        part = os.path.join(d, "b%d" % n)
        open(part, "wb").write(blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)])
        co = os.path.join(d, "co%d" % n)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], stdout=subprocess.PIPE, text=True).stdout
        for line in out.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC" and not f[7].endswith(".kd"):
                rows.append((int(f[2]), f[7]))
for sz, nm in sorted(rows):
    dem = subprocess.run(["c++filt", nm], stdout=subprocess.PIPE, text=True).stdout.strip()
    if flt and flt not in dem: continue
    print("%8d  %s" % (sz, dem[:100]))
