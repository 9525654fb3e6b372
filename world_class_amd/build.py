"""Builds libworldclass_hip.so (the C-ABI of include/world_class_c.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
The built .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libworldclass_hip.so")
OBJ = os.path.join(HERE, "_obj")
EXTRA = os.environ.get("WC_EXTRA_FLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "world_class_c.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "world_class_io.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "world_class_codec.h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "world_matlabfunctions.hpp"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "world_fft.hpp"))
    # a shipped library that is newer than every source is used as is (the object directory does not travel to
    # the GPU box); experimental flags always rebuild
    if not force and not EXTRA and not _stale(OUT, [os.path.join(CSRC, f) for f in sources()] + headers):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    # objects depend on the flags too: a flags stamp forces a full rebuild when they change
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    jobs = []
    objs = []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])

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
    if jobs or force or _stale(OUT, objs):
        rc, out = run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
        if out.strip():
            print(out)
        if rc != 0:
            raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
