#!/usr/bin/env python3
"""Side measurement (not the headline bench): BASELINE config 4 shape -- online sparse GP, capacity 200, patches of 256
points streamed in 4 chunks of 64 -- through gpc_sparse_add_dev / gpc_sparse_predict_dev.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gp_compressor_amd import capi, synth  # noqa: E402

P = int(os.environ.get("P", "4096"))
cap = int(os.environ.get("CAP", "200"))
n, chunks, res, sz = 256, 4, 0.15, 20
dev = torch.device("cuda:0")
off, x0, x1, y = synth.make_patches(P, n, res=res, seed=4)
ctx = capi.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
kw = dict(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4, capacity=cap) if os.environ.get("KERNEL", "fill") == "fill" else dict(capacity=cap)
g = capi.Sparse(ctx, capi.default_params_sparse(1, **kw), P, 1)
t = lambda a: torch.from_numpy(a).to(dev)
xs0, xs1 = synth.grid(res, sz)
d_xs0, d_xs1 = t(xs0), t(xs1)
f = torch.empty((P, 1, sz * sz), dtype=torch.float64, device=dev)
bufs = []
cn = n // chunks
coff = t((np.arange(P + 1) * cn).astype(np.int32))
for c in range(chunks):
    idx = (off[:-1, None] + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
    bufs.append((t(x0[idx]), t(x1[idx]), t(np.ascontiguousarray(y[:, idx]))))
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in range(chunks):
    g.add_dev(coff, cn, P * cn, *bufs[c])
torch.cuda.synchronize()
t_add = time.perf_counter() - t0
t0 = time.perf_counter()
g.predict_dev(sz * sz, d_xs0, d_xs1, f)
torch.cuda.synchronize()
t_pred = time.perf_counter() - t0
b = g.sizes()
# row f1: registration inner loop on the trained state -- likelihoods + derivatives at 256 points per patch
doff = t(off.astype(np.int32))
d_q0, d_q1, d_yq = t(x0), t(x1), t(np.ascontiguousarray(y))
dX = torch.empty((P * n, 3), dtype=torch.float64, device=dev)
lik = torch.empty((P * n,), dtype=torch.float64, device=dev)
g.likelihood_dev(doff, P * n, d_q0, d_q1, d_yq, dX, lik)
torch.cuda.synchronize()
t0 = time.perf_counter()
g.likelihood_dev(doff, P * n, d_q0, d_q1, d_yq, dX, lik)
torch.cuda.synchronize()
t_lik = time.perf_counter() - t0
bm = float(b.mean())
lik_flops = P * n * (2.0 * bm * bm + 30.0 * bm)
lik_bytes = P * (n / 32.0) * 8.0 * bm * bm          # C streamed once per 32-point chunk
print(json.dumps({"workload": f"C4-shape: {P} patches x {n} pts in {chunks} chunks, capacity {cap}", "add_s": t_add, "predict_s": t_pred,
                  "patches_per_s": P / (t_add + t_pred), "bv_mean": float(b.mean()), "bv_max": int(b.max()),
                  "finite": bool(torch.isfinite(f).all().item()),
                  "likelihood": {"s": t_lik, "points_per_s": P * n / t_lik, "gflops": lik_flops / t_lik / 1e9,
                                 "c_stream_GBps": lik_bytes / t_lik / 1e9, "nonfinite_frac": float((~torch.isfinite(dX).all(dim=1)).double().mean().item()),
                                 "note": "non-finite rows = sigma <= 0 after cancellation; the reference's likelihood_dx does not clamp either"}}))
