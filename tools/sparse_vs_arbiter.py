#!/usr/bin/env python3
"""Diagnostic: distance of the GPU sparse add and of the fp64 CPU oracle from the binary128 arbiter, per patch, on a larger sample
than the parity tests use.   P=64 CAP=50 python tools/sparse_vs_arbiter.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from gp_compressor_amd import capi, synth
import oracle_lib as oracle
P, cap, n, res = int(os.environ.get("P", "64")), int(os.environ.get("CAP", "50")), 256, 0.15
ctx = capi.Context(0)
for seed in (44, 45):
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=seed)
    xs0, xs1 = synth.grid(res, 8)
    kw = dict(p0=1.0, p1=(res / 8) ** 2, s20=1e-4, capacity=cap)
    p = capi.default_params_sparse(1, sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4, capacity=cap)
    g = capi.Sparse(ctx, p, P, 1)
    g.add(off, x0, x1, y)
    f_gpu = g.predict(xs0, xs1)[0][:, 0, :]
    g.close()
    e_g, e_o = [], []
    for i in range(P):
        sl = slice(off[i], off[i + 1])
        h = oracle.SparseHP(oracle.sparse_params(1, **kw), cap + 2)
        h.add_measurements(x0[sl], x1[sl], y[:, sl], None)
        fh = h.predict(xs0, xs1)[0][0]
        o = oracle.Sparse(oracle.sparse_params(1, **kw), cap + 2)
        o.add_measurements(x0[sl], x1[sl], y[:, sl], None)
        fo = o.predict(xs0, xs1)[0][0]
        sc = np.max(np.abs(fh))
        e_g.append(np.max(np.abs(f_gpu[i] - fh)) / sc); e_o.append(np.max(np.abs(fo - fh)) / sc)
    e_g, e_o = np.array(e_g), np.array(e_o)
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    print(f"seed {seed} P {P} cap {cap}: GPU rms {rms(e_g):.2e} median {np.median(e_g):.2e} max {e_g.max():.2e} | "
          f"CPU oracle rms {rms(e_o):.2e} median {np.median(e_o):.2e} max {e_o.max():.2e}", flush=True)
