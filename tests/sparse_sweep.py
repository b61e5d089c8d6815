"""Randomised sweep of the sparse add's phases (round 4: rows -> second rows phase -> mid -> regular kernel over chained work lists).

For every drawn configuration (channels, capacity, kernel regime, patch count, points per patch, ragged, insertion order, three add calls)
the states, basis sizes, status words, point counts and per-point decision bytes of the DEFAULT path must equal, bit for bit, those of the
path without any early phase (GPC_SPARSE_NO_SMALL: the regular kernel alone) in the full mode -- the property
test_sparse_rows_phase_is_bit_identical asserts on its fixed cases.  Test infrastructure (used by tests/test_sparse_gpu.py and
tools/r4_stress_sparse.py); the first 300 configurations of seed 1 include the two (3 channels, capacity 17, eps_tol 1e-14) on which
divisions sharing one refined reciprocal diverged from the other kernel shapes (round 4, not kept)."""
import os

import numpy as np


def draw(rng, synth, res=0.15):
    ny = int(rng.choice([1, 1, 3]))
    cap = int(rng.choice([5, 12, 16, 17, 23, 24, 25, 33, 47, 48, 49, 64, 100, 200, -1]))
    kernel = str(rng.choice(["default", "mixed", "geo", "fill", "mixed2"]))
    P = int(rng.integers(1, 90))
    n = int(rng.integers(2, 200))
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), ny=ny, n_min=1)
    perm = synth.sattolo_perms(off, seed=int(rng.integers(1 << 30))) if rng.integers(2) else None
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4 if ny == 1 else 1.0)
    if kernel == "geo":
        kw.update(sigmaf_sq=1.0, l_sq=(res * float(rng.choice([0.6, 1.0, 2.0]))) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 3) ** 2, noise=1e-3 if ny == 1 else 1.0, eps_tol=1e-3)
    if kernel == "mixed2":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 5) ** 2, noise=1e-2 if ny == 1 else 1.0, eps_tol=1e-4)
    return dict(ny=ny, cap=cap, kernel=kernel, P=P, n=n, off=off, x0=x0, x1=x1, y=y, perm=perm, kw=kw)


def run(capi, ctx, c, env=None, predict=None):
    """three add calls (the first in a drawn insertion order) under the environment switches `env` ("A+B" for two); returns
    (status / trace per call, basis sizes, state)"""
    names = env.split("+") if env else []
    for e in names:
        os.environ[e] = "1"
    try:
        g = capi.Sparse(ctx, capi.default_params_sparse(c["ny"], **c["kw"]), c["P"], c["ny"])
        r = []
        for call in range(3):
            st, tr = g.add(c["off"], c["x0"], c["x1"], c["y"], c["perm"] if call == 0 else None, trace=True)
            r += [st, tr]
        out = (r, g.sizes(), g.state())
        if predict is not None:
            # the mean on a grid through the small-basis predict kernels and through the regular one: the same bits (sigma differs in the
            # order of a sum and is held by test_sparse_small_basis_predict_kernel)
            xs0, xs1 = predict
            f_small = g.predict(xs0, xs1)[0]
            os.environ["GPC_SPARSE_NO_SMALL_PREDICT"] = "1"
            try:
                f_reg = g.predict(xs0, xs1)[0]
            finally:
                os.environ.pop("GPC_SPARSE_NO_SMALL_PREDICT", None)
            out = out + (bool(np.array_equal(f_small, f_reg, equal_nan=True)),)
        g.close()
    finally:
        for e in names:
            os.environ.pop(e, None)
    return out


def same(a, b, P):
    (ra, ba, sa), (rb, bb, sb) = a[:3], b[:3]
    if not all(np.array_equal(u, v) for u, v in zip(ra, rb)) or not np.array_equal(ba, bb):
        return False
    for i in range(P):
        nb = int(ba[i])
        if not (np.array_equal(sa[0][i][:, :nb], sb[0][i][:, :nb], equal_nan=True) and np.array_equal(sa[1][i][:nb, :nb], sb[1][i][:nb, :nb], equal_nan=True)
                and np.array_equal(sa[2][i][:nb, :nb], sb[2][i][:nb, :nb], equal_nan=True) and np.array_equal(sa[3][i][:nb], sb[3][i][:nb], equal_nan=True)):
            return False
    return True


def sweep(capi, synth, ctx, ncfg, seed, progress=None):
    """-> (mismatching configurations, histogram of the final basis sizes seen).  The caller sets GPC_SPARSE_FULL=1 (kernel SHAPES are compared
    in the full mode: the triangular passes of the four-wave shape sum a row in another order)."""
    rng = np.random.default_rng(seed)
    grid = synth.grid(0.15, 12)
    bad, hist = [], {"max_b": 0, "le16": 0, "17_24": 0, "25_48": 0, "gt48": 0}
    for k in range(ncfg):
        c = draw(rng, synth)
        a, b = run(capi, ctx, c, predict=grid), run(capi, ctx, c, "GPC_SPARSE_NO_SMALL")
        pred_ok = a[3]
        a = a[:3]
        ba = a[1]
        hist["max_b"] = max(hist["max_b"], int(ba.max()))
        hist["le16"] += int((ba <= 16).sum())
        hist["17_24"] += int(((ba > 16) & (ba <= 24)).sum())
        hist["25_48"] += int(((ba > 24) & (ba <= 48)).sum())
        hist["gt48"] += int((ba > 48).sum())
        if not same(a, b, c["P"]) or not pred_ok:
            bad.append({"config": k, "ny": c["ny"], "cap": c["cap"], "kernel": c["kernel"], "P": c["P"], "n": c["n"], "predict_ok": pred_ok})
        if progress:
            progress(k, bad)
    return bad, hist
