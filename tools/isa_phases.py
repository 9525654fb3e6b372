#!/usr/bin/env python
"""Static instruction counts of one kernel BETWEEN its shader-clock stamps (a WC_SYN_TRACE / WC_RQ_PROF style build): the
disassembly of the kernel cut at every s_memtime / s_memrealtime, per stretch the vector instructions by class, LDS and
memory instructions and waits -- to set beside the cycles the stamps measure (tools/syn_trace.py):
    WC_LIB_PATH=world_class_amd/_variants/syntrace.so python tools/isa_phases.py 'syn_pulse_wave_kernel<false>' [--dump file]
Static: a loop body counts once, both sides of a branch count; stretches come in ADDRESS order, with the label that starts them."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mix import LLVM, ROOT, classify  # noqa: E402


def disassembly(so):
    out = []
    with tempfile.TemporaryDirectory() as d:
        fb = os.path.join(d, "fb")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True)
        blob = open(fb, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for n, st in enumerate(starts):
            part = os.path.join(d, "b%d" % n)
            with open(part, "wb") as f:
                f.write(blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "co%d" % n)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
            out.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", co], stdout=subprocess.PIPE, text=True).stdout)
    return "\n".join(out)


def main():
    so = os.environ.get("WC_LIB_PATH") or os.path.join(ROOT, "world_class_amd", "libworldclass_hip.so")
    want = sys.argv[1]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    cur, lines = None, []
    for line in disassembly(so).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            continue
        if cur and want in cur and ".kd" not in cur:
            lines.append(line)
    if dump:
        open(dump, "w").write("\n".join(lines))
    seg, segs, label = Counter(), [], "entry"
    for line in lines:
        t = line.split()
        if not t:
            continue
        if t[0].startswith("s_memtime") or t[0].startswith("s_memrealtime"):
            segs.append((label, seg))
            seg, label = Counter(), "stamp"
            continue
        if re.match(r"^[vs]_|^ds_|^global_|^scratch_|^flat_|^buffer_", t[0]):
            seg[t[0]] += 1
    segs.append((label, seg))
    print("%-4s %8s %8s %8s %6s %6s %6s %6s" % ("#", "valu", "fp64", "trans", "lds", "vmem", "salu", "waits"))
    for n, (label, ops) in enumerate(segs):
        cls = Counter()
        for op, c in ops.items():
            cls[classify(op)] += c
        valu = sum(c for k, c in cls.items() if k.startswith("valu_"))
        trans = sum(c for op, c in ops.items() if re.match(r"^v_(rcp|rsq|sqrt|log|exp|sin|cos)_", op))
        print("%-4d %8d %8d %8d %6d %6d %6d %6d" % (n, valu, cls["valu_fp64"], trans, cls["lds"], cls["vmem"], cls["salu_branch"], cls["wait_barrier_nop"]))


if __name__ == "__main__":
    main()
