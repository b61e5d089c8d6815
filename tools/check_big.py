"""Diagnostic: the tiled kernel (GPC_FORCE_BIG=1) against the generic kernel on single patches of NS=n1,n2,.. points, four repetitions
each (a hand-over race shows as an error that changes from run to run).   NS=512,816,1024 python tools/check_big.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gp_compressor_amd import capi, synth
ctx = capi.Context(0)
xs0, xs1 = synth.grid(0.15, 12)
for (P, n, rag) in [(1, int(a), False) for a in os.environ.get('NS', '1024').split(',')]:
    off, x0, x1, y = synth.make_patches(P, n, seed=11, ragged=rag, ny=1)
    p = capi.default_params_dense()
    os.environ["GPC_FORCE_GENERIC"] = "1"
    fg, _, sg, ag = ctx.dense_fit_predict(p, off, x0, x1, y, xs0, xs1, want_alpha=True)
    kg = ctx.last_dense_kernel()
    del os.environ["GPC_FORCE_GENERIC"]
    os.environ["GPC_FORCE_BIG"] = "1"
    outs = []
    for rep in range(4):
        f, _, s, a = ctx.dense_fit_predict(p, off, x0, x1, y, xs0, xs1, want_alpha=True)
        outs.append((f.copy(), a.copy(), s.copy()))
    del os.environ["GPC_FORCE_BIG"]
    kb = ctx.last_dense_kernel()
    sc = np.max(np.abs(ag))
    errs = [float(np.max(np.abs(o[1] - ag)) / sc) for o in outs]
    print(P, n, kg, kb, "alpha rel err per rep:", ["%.2e" % e for e in errs], "status", outs[0][2][:4], flush=True)
    if max(errs) > 1e-7:
        a0 = outs[int(np.argmax(errs))][1].reshape(-1); d = np.abs(a0 - ag.reshape(-1)) / sc
        bad = np.nonzero(d > 1e-8)[0]
        print("   first/last bad alpha index:", bad[:5], bad[-5:], "count", bad.size, "patch offs", off[:5])
