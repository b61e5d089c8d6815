#!/usr/bin/env python3
"""The host-pointer entry of the dense path on the C2 batch (H2D + kernel + D2H, page-locked caller buffers): ms per call, and that the
result equals the device entry's.  GPC_HOST_ONE_STREAM=1 / GPC_HOST_NO_PIPELINE=1 select the older forms (same-box A/B)."""
import ctypes as C_
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gp_compressor_amd import capi, synth  # noqa: E402

P, n, RES, SZ = int(os.environ.get("P", "8192")), int(os.environ.get("N", "256")), 0.15, 20
M = SZ * SZ
off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=2)
ctx = capi.Context(0)
if os.environ.get("NULL_STREAM"):                       # as bench.py runs it: the context on torch's current (legacy default) stream
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
prm = capi.default_params_dense()
if os.environ.get("LIKE_BENCH"):                        # the sequence of bench.py before its host-pointer measurement
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_off, d_x0, d_x1, d_y = tt(off), tt(x0), tt(x1), tt(y)
    d_f = torch.empty((P, 1, M), dtype=torch.float64, device="cuda")
    d_st = torch.empty((P,), dtype=torch.int32, device="cuda")
    for _ in range(25 if "dev" in os.environ["LIKE_BENCH"] or os.environ["LIKE_BENCH"] == "1" else 0):
        ctx.dense_fit_predict_grid_dev(prm, P, d_off, n, P * n, d_x0, d_x1, d_y, 1, RES, SZ, d_f, status=d_st)
    torch.cuda.synchronize()
    for _ in range(3 if "page" in os.environ["LIKE_BENCH"] or os.environ["LIKE_BENCH"] == "1" else 0):
        ctx.dense_fit_predict_grid(prm, off, x0, x1, y, RES, SZ)
f_dev, st_dev = ctx.dense_fit_predict_grid(prm, off, x0, x1, y, RES, SZ)          # pageable path once (also the reference result)
pin = {k: ctx.host_array(a.shape, a.dtype) for k, a in (("off", off), ("x0", x0), ("x1", x1), ("y", y))}
for k, a in (("off", off), ("x0", x0), ("x1", x1), ("y", y)):
    pin[k][...] = a
pf = ctx.host_array((P, 1, M))
pst = ctx.host_array((P,), np.int32)


def call():
    rc = ctx.lib.gpc_dense_fit_predict_grid(ctx.h, C_.byref(prm), P, pin["off"].ctypes.data, pin["x0"].ctypes.data, pin["x1"].ctypes.data,
                                            pin["y"].ctypes.data, 1, RES, SZ, pf.ctypes.data, None, pst.ctypes.data)
    assert rc == 0, rc


for _ in range(1 if os.environ.get("LIKE_BENCH") else 3):
    call()
ts = []
for _ in range(5 if os.environ.get("LIKE_BENCH") else 15):
    t0 = time.perf_counter()
    call()
    ts.append(1e3 * (time.perf_counter() - t0))
ts = np.array(ts)
print(json.dumps({"P": P, "n": n, "ms_median": float(np.median(ts)), "ms_min": float(ts.min()), "ms_max": float(ts.max()),
                  "patches_per_s": P / (np.median(ts) * 1e-3), "equal_to_first_call": bool(np.array_equal(pf, f_dev)),
                  "status_ok": bool(np.all(pst == 0)), "kernel": ctx.last_dense_kernel(),
                  "null_stream": bool(os.environ.get("NULL_STREAM")), "mode": "one stream" if os.environ.get("GPC_HOST_ONE_STREAM") else "no pipeline" if os.environ.get("GPC_HOST_NO_PIPELINE") else "default"}))
ctx.close()
