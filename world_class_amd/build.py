"""Builds libworldclass_hip.so (the C-ABI of include/world_class_c.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
The built .so is git-ignored but travels with the tree to the GPU box.

A shipped library is reused only when it provably belongs to the sources and flags in the tree: the SHA-256 of
every source, header and the compiler flags is compiled into the library (wc_core.hip, `wc_build_hash()`, found
here by scanning the file for its marker -- no dlopen) and must equal the hash of what is on disk now; file
times play no part in that decision.  Anything else (edited source, other flags, WC_EXTRA_FLAGS experiments,
a library from an older tree) rebuilds.
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libworldclass_hip.so")
OBJ = os.path.join(HERE, "_obj")
EXTRA = os.environ.get("WC_EXTRA_FLAGS", "").split()
# -simplifycfg-sink-common=false: where a wavefront's lane 0 takes a branch of its own on the register-resident transforms
# (wc_wavefft.hpp), sinking the branches' common tail turns static register-array indices into dynamic ones and the arrays
# into scratch memory
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result", "-mllvm", "-simplifycfg-sink-common=false"]
HASH_MARKER = b"WC_SOURCE_HASH="


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    inc = os.path.join(os.path.dirname(HERE), "include")
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))
    hs += [os.path.join(inc, f) for f in ("world_class_c.h", "world_class_io.h", "world_class_codec.h", "world_class_stream.h", "world_class_shard.h",
                                          "world_matlabfunctions.hpp", "world_fft.hpp") if os.path.exists(os.path.join(inc, f))]
    return hs


def source_hash():
    """SHA-256 over the flags and the bytes of every translation unit and header the library is built from"""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, f) for f in sources()] + headers():
        h.update(b"\0" + os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def embedded_hash(path=None):
    """the source hash compiled into a built library, or None"""
    path = path or OUT
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(HASH_MARKER + rb"([0-9a-f]{64})", f.read())
    return m.group(1).decode() if m else None


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    want = source_hash()
    if not force and embedded_hash() == want:
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    # objects depend on the flags too: a flags stamp forces a full rebuild when they change
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    jobs = []
    objs = []
    hs = headers()
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        cmd = [hipcc] + FLAGS
        if src == "wc_core.hip":  # carries the hash: recompiled whenever anything changed
            cmd += ['-DWC_SOURCE_HASH="%s"' % want]
            jobs.append(cmd + ["-c", s, "-o", o])
        elif force or _stale(o, [s] + hs):
            jobs.append(cmd + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return r.returncode, r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        for rc, out in ex.map(run, jobs):
            if out.strip():
                print(out)
            if rc != 0:
                raise RuntimeError("hipcc failed")
    with open(stamp, "w") as f:
        f.write(flags_now)
    rc, out = run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])
    if out.strip():
        print(out)
    if rc != 0:
        raise RuntimeError("link failed")
    if embedded_hash() != want:
        raise RuntimeError("built library does not carry the expected source hash")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
