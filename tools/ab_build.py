#!/usr/bin/env python
"""Builds a variant of libworldclass_hip.so with extra compiler flags for A/B runs on the GPU box:

    python tools/ab_build.py NAME -DWC_SOMETHING=1 ...      ->  world_class_amd/_variants/NAME.so

and is selected with WC_LIB_PATH=world_class_amd/_variants/NAME.so (the directory travels with gpurun, is git-ignored).
The default library is not touched."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    from world_class_amd import build
    vdir = os.path.join(build.HERE, "_variants")
    os.makedirs(vdir, exist_ok=True)
    build.OUT = os.path.join(vdir, name + ".so")
    build.OBJ = os.path.join(vdir, "_obj_" + name)
    build.FLAGS = flags + build.FLAGS
    print(build.build(force=True))


if __name__ == "__main__":
    main()
