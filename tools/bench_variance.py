#!/usr/bin/env python3
"""Side measurement: dense fit + mean + predictive variance (gaussian_process::predict_measurements computes V* always,
src/gaussian_process.cpp:35-43) at the C2 / C3 / C5 patch sizes through gpc_dense_fit_predict_dev.  One JSON line per size."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gp_compressor_amd import capi, synth  # noqa: E402

res, sz = 0.15, 20
m = sz * sz
dev = torch.device("cuda:0")
ctx = capi.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
xs0, xs1 = synth.grid(res, sz)
d_xs0, d_xs1 = t(xs0), t(xs1)
for P, n in ((8192, 256), (4096, 512), (1024, 1024)):
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=2)
    prm = capi.default_params_dense(want_variance=1)
    d = [t(a) for a in (off, x0, x1, y)]
    f = torch.empty((P, 1, m), dtype=torch.float64, device=dev)
    v = torch.empty((P, m), dtype=torch.float64, device=dev)
    st = torch.empty((P,), dtype=torch.int32, device=dev)
    call = lambda: ctx.dense_fit_predict_dev(prm, P, d[0], n, P * n, d[1], d[2], d[3], 1, m, d_xs0, d_xs1, f, v_star=v, status=st)
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    fl = (3.5 * n * n + n ** 3 / 3 + 2 * n * n + 9 * n * m + float(n) * n * m + 2 * n * m) * P
    ok = bool((st == 0).all().item()) and bool(torch.isfinite(v).all().item()) and bool((v > -1e-12).all().item())
    print(json.dumps({"workload": f"{P} patches x {n} pts, fit + mean + variance on {m} points", "ms": 1e3 * dt, "patches_per_s": P / dt,
                      "tflops": fl / dt / 1e12, "frac_of_78.6": fl / dt / 1e12 / 78.6, "kernel": ctx.last_dense_kernel(), "ok": ok}))
