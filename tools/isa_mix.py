#!/usr/bin/env python
"""Static opcode mix of the hot kernels in libworldclass_hip.so (disassembly of the embedded gfx950 code objects):
    python tools/isa_mix.py [filter ...] > profiles/<tag>_isa_mix.txt
Per kernel: instructions by class, the FP64-arithmetic share of the vector instructions, and the most frequent opcodes of
the vector instructions that are NOT FP64 arithmetic (what an instruction diet would go after).  Static counts: a loop body
counts once, both sides of a branch count."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT = ["ct_wave_kernel", "d4c2_", "syn_pulse_wave_kernel", "hv_refine_group", "hv_bandpass_sdft", "hv_raw_kernel", "hv_raw_wave_kernel", "hv_decimate_scan", "ct_frames_kernel<2048",
           "d4c_frames_kernel<4096, 512, true", "d4c_band_kernel<4096", "syn_pulse_kernel<2048"]
F64 = re.compile(r"^v_(add|mul|fma|fmac|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|rndne|floor|fract|trig_preop|ceil|trunc)_f64")


def classify(op):
    if op.startswith("v_"):
        if F64.match(op):
            return "valu_fp64"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_compare"
        if op.startswith("v_cvt") or op.startswith("v_frexp_exp"):
            return "valu_convert"
        if op.startswith("v_cndmask"):
            return "valu_select"
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_swap"):
            return "valu_move"
        if "dpp" in op or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or op.startswith("v_permlane"):
            return "valu_crosslane"
        return "valu_int_address"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait_barrier_nop"
    if op.startswith("s_"):
        return "salu_branch"
    return "other"


def main():
    so = os.environ.get("WC_LIB_PATH") or os.path.join(ROOT, "world_class_amd", "libworldclass_hip.so")
    filters = sys.argv[1:] or DEFAULT
    kernels = {}
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
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", co], stdout=subprocess.PIPE, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels.setdefault(cur, Counter())
                    continue
                t = line.split()
                if cur and t and re.match(r"^[vs]_|^ds_|^global_|^scratch_|^flat_|^buffer_", t[0]):
                    kernels[cur][t[0]] += 1
    print("static opcode mix, %s" % os.path.basename(so))
    for name in sorted(kernels):
        short = name.replace("wc::", "").replace("void ", "")
        short = short[:short.index("(")] if "(" in short and not short.startswith("(") else short
        if not any(f in short for f in filters) or ".kd" in name:
            continue
        ops = kernels[name]
        cls = Counter()
        for op, c in ops.items():
            cls[classify(op)] += c
        valu = sum(c for k, c in cls.items() if k.startswith("valu_"))
        if valu == 0:
            continue
        print("\n%s" % short)
        print("  vector instructions %d, FP64 arithmetic %d (%.0f %% of vector)" % (valu, cls["valu_fp64"], 100.0 * cls["valu_fp64"] / valu))
        print("  " + ", ".join("%s %d" % (k, cls[k]) for k in ("valu_fp64", "valu_int_address", "valu_select", "valu_move", "valu_convert", "valu_compare",
                                                                "valu_crosslane", "lds", "vmem", "scratch", "salu_branch", "wait_barrier_nop") if cls[k]))
        rest = Counter({op: c for op, c in ops.items() if classify(op).startswith("valu_") and classify(op) != "valu_fp64"})
        print("  top non-FP64 vector opcodes: " + ", ".join("%s %d" % (op, c) for op, c in rest.most_common(8)))


if __name__ == "__main__":
    main()
