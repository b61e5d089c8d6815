#!/usr/bin/env python3
"""Diagnostic: per-phase cycle breakdown of dense_big_kernel (tiled kernel, 256 < n <= 1024), incl. the sub-phases of a
factorisation step.   Build (container):  python tools/stamp_big.py --build      Run (GPU box):  N=512 P=2048 python tools/stamp_big.py
The diagnostic library (-DBG_SUBSTAMPS) is never used by tests or bench; the shipped kernel executes no sub-stamp."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMP_LIB = os.path.join(ROOT, "gp_compressor_amd", "libgpc_hip_bigstamps.so")

if "--build" in sys.argv:
    from gp_compressor_amd import build
    print(build.build(lib=STAMP_LIB, extra_flags=("-DBG_SUBSTAMPS=1",), verbose=False))
    sys.exit(0)

os.environ.setdefault("GPC_LIB_PATH", STAMP_LIB)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from gp_compressor_amd import capi, synth  # noqa: E402

P = int(os.environ.get("P", "2048"))
n = int(os.environ.get("N", "512"))
off, x0, x1, y = synth.make_patches(P, n, seed=2)
ctx = capi.Context(0)
f, st = ctx.dense_fit_predict_grid(capi.default_params_dense(), off, x0, x1, y, 0.15, 20)
t0 = time.perf_counter()
f, st = ctx.dense_fit_predict_grid(capi.default_params_dense(), off, x0, x1, y, 0.15, 20)
print("kernel:", ctx.last_dense_kernel(), "status ok:", bool(np.all(st == 0)), "host-call s:", time.perf_counter() - t0)
os.environ["GPC_BIG_STAMPS"] = "1"
f, st = ctx.dense_fit_predict_grid(capi.default_params_dense(), off, x0, x1, y, 0.15, 20)
