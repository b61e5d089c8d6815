#!/usr/bin/env python3
"""Diagnostic variant of libgpc_hip.so in seconds: ONE translation unit recompiled with extra flags, linked with the shipped objects of the
others (gp_compressor_amd/build.py rebuilds every unit for a variant).
    python tools/build_variant.py <name> <unit.hip> [-DFLAG ...]   -> gp_compressor_amd/libgpc_hip_<name>.so   (select it with GPC_LIB_PATH)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gp_compressor_amd", "csrc")


def main():
    name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    # (the other units come from the objects of the last regular build as they are: a variant never touches the shipped library)
    src = os.path.join(CSRC, unit)
    obj = os.path.join(CSRC, os.path.splitext(unit)[0] + f".{name}.o")
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-c", src, "-o", obj, *flags])
    objs = [o for o in sorted(glob.glob(os.path.join(CSRC, "*.o"))) if o.count(".") == 1 and os.path.basename(o) != os.path.splitext(unit)[0] + ".o"]
    lib = os.path.join(ROOT, "gp_compressor_amd", f"libgpc_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *objs, "-ldl"])
    os.remove(obj)
    print(lib)


if __name__ == "__main__":
    main()
