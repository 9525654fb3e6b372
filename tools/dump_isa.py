#!/usr/bin/env python
"""Disassembly of one kernel of libworldclass_hip.so (development aid): python tools/dump_isa.py ct_wave_kernel > /tmp/ct.s"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("WC_LIB_PATH") or os.path.join(ROOT, "world_class_amd", "libworldclass_hip.so")
want = sys.argv[1]
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
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True).stdout
        on = False
        for line in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                on = want in m.group(1) and not m.group(1).endswith(".kd")
                if on: print(line)
                continue
            if on: print(line)
