#!/usr/bin/env python
"""VGPRs / SGPRs / scratch / LDS of every kernel in libworldclass_hip.so, read from the code-object metadata
(llvm-readelf --notes of the embedded gfx950 code object): python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = os.environ.get("WC_LIB_PATH") or os.path.join(ROOT, "world_class_amd", "libworldclass_hip.so")
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    notes = ""
    with tempfile.TemporaryDirectory() as d:
        # the library carries one offload bundle per translation unit, back to back in .hip_fatbin
        fb = os.path.join(d, "fb")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True)
        blob = open(fb, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(magic, blob)]
        for n, st in enumerate(starts):
            part = os.path.join(d, "b%d" % n)
            with open(part, "wb") as f:
                f.write(blob[st:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "co%d" % n)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True).stdout
    rows = []
    for blk in notes.split("- .agpr_count")[1:]:
        def g(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = g("name")
        if flt and flt not in name:
            continue
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        # "void (anonymous namespace)::k<...>(args)" / "wc::k(args)": drop the return type and the argument list, keep template arguments
        dem = re.sub(r"^void ", "", dem).replace("(anonymous namespace)::", "").replace("wc::", "")
        depth, cut = 0, len(dem)
        for i, ch in enumerate(dem):
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                cut = i
                break
        dem = dem[:cut]
        rows.append((dem, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print("%-60s %5s %5s %8s %8s" % ("kernel", "vgpr", "sgpr", "scratch", "lds"))
    for r in sorted(rows):
        print("%-60s %5s %5s %8s %8s" % r)


if __name__ == "__main__":
    main()
