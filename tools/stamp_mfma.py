#!/usr/bin/env python3
"""Diagnostic: per-phase cycle breakdown of dense_mfma_kernel on the bench workload (C2).
Build (in the container, travels with gpurun):   python tools/stamp_mfma.py --build          (coarse phase stamps)
                                                 python tools/stamp_mfma.py --build --trace  (per-step timeline of every wave)
Run (on the GPU box):                            python tools/stamp_mfma.py [--trace]
The diagnostic libraries are never used by tests or bench; the shipped kernel executes no stamp."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE = "--trace" in sys.argv
STAMP_LIB = os.path.join(ROOT, "gp_compressor_amd", "libgpc_hip_trace.so" if TRACE else "libgpc_hip_stamps.so")

if "--build" in sys.argv:
    from gp_compressor_amd import build
    flags = ("-DMF_TRACE=1",) if TRACE else ("-DMF_STAMPS=" + os.environ.get("MF_STAMPS", "1"),)
    print(build.build(lib=STAMP_LIB, extra_flags=flags, verbose=True))
    sys.exit(0)

os.environ.setdefault("GPC_LIB_PATH", STAMP_LIB)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from gp_compressor_amd import capi, synth  # noqa: E402

P = int(os.environ.get("P", "2048"))
n = int(os.environ.get("N", "256"))
off, x0, x1, y = synth.make_patches(P, n, seed=2)
ctx = capi.Context(0)
for rep in range(2):
    f, st = ctx.dense_fit_predict_grid(capi.default_params_dense(), off, x0, x1, y, 0.15, 20)
print("kernel:", ctx.last_dense_kernel(), "status ok:", bool(np.all(st == 0)))
