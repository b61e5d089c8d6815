"""GPU parity tests of the online sparse GP (sparse_gp / sparse_gp_field) through the C-ABI.

Oracle: oracle/gpc_oracle.c run patch by patch with the same explicit insertion order.
Stated tolerance (fp64, tree-reduced sums on the GPU vs sequential sums in the oracle):
  * well-conditioned regimes (exact / capacity-bounded with a kernel that fills the basis):
        identical basis-vector bookkeeping (counts, BV order); f* and sigma within 2e-5 of max|f*| on random batches
        (Q = K_BV^-1 reaches 1e6 with eps_tol = 1e-6f, and the two CPU restatements -- C oracle vs NumPy -- already
        differ by up to 4e-6 on these very inputs), within 1e-7 on the committed golden sequences
  * the reference's default hyper-parameters (sigma_f^2 = 100, l^2 = 1 on a 0.15 m patch): the sparse-vs-full
    decision `gamma < 1e-6f` is taken on rounding noise (|Q| ~ 1e6), so two correct fp64 implementations disagree
    in the BV count; only f* is compared, to 1e-2 of max|f*| (see tests/test_oracle.py for the CPU-vs-CPU evidence).

What these tolerances mean is measured, not assumed: test_sparse_gpu_vs_arbiter (bottom of this file) runs the same recursion
in IEEE binary128 (oracle/gpc_oracle_hp.c) and shows the GPU and the fp64 CPU oracle to be equally far from the exact
recursion in every regime -- ~1e-6 of max|f*| when well conditioned (hence 2e-5 between the two), ~1e-2 with the C4
basis-filling kernel at capacity 200 and at the reference's default hyper-parameters (hence 1e-2 .. 2e-2 there), where each
fp64 implementation also takes 1-3 % of the branch decisions differently from the exact recursion.
"""
import os

import numpy as np
import pytest

from gp_compressor_amd import synth
from np_restatement import train_sigmaf_np

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gp():
    from gp_compressor_amd import capi
    capi.load()
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def _case(name):
    z = np.load(os.path.join(GOLD, "sparse_cases.npz"))
    return {k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(name + ".")}


def _params(capi, d):
    p0, p1, s20, eps, cap, ny, bug, probit = d["params"]
    return capi.default_params_sparse(int(ny), sigmaf_sq=p0, l_sq=p1, noise=s20, eps_tol=eps, capacity=int(cap),
                                      ref_field_delete_bug=int(bug), noise_model=int(probit)), int(ny)


@pytest.mark.parametrize("name,ftol,strict", [("exact", 1e-8, True), ("cap12", 1e-7, True), ("field_bug", 1e-7, True),
                                              ("field_fixed", 1e-7, True), ("defaults", 1e-2, False)])
def test_sparse_golden(gp, name, ftol, strict):
    capi, ctx = gp
    d = _case(name)
    p, ny = _params(capi, d)
    n = d["x0"].shape[0]
    off = np.array([0, n], dtype=np.int32)
    g = capi.Sparse(ctx, p, 1, ny)
    st = g.add(off, d["x0"], d["x1"], d["y"], d["perm"])
    f, s, st2 = g.predict(d["xs0"], d["xs1"])
    b = int(g.sizes()[0])
    fscale = max(np.max(np.abs(d["f_star"])), 1e-6)
    assert np.max(np.abs(f[0] - d["f_star"])) <= ftol * fscale
    assert np.max(np.abs(s[0] - d["sigma"])) <= ftol * max(np.max(d["sigma"]), 1.0)
    if strict:
        assert st[0] == 0 and b == int(d["b"])
        alpha, C, Q, BV = g.state()
        assert np.array_equal(BV[0, :b], d["BV"])
        ascale = np.max(np.abs(d["alpha"]))
        assert np.max(np.abs(alpha[0, :, :b] - d["alpha"])) <= 1e-5 * ascale
    g.close()


def _oracle_batch(oracle, op, off, x0, x1, y, perm, xs0, xs1, max_bv):
    P = len(off) - 1
    ny = op.ny
    f = np.zeros((P, ny, len(xs0)))
    s = np.zeros((P, len(xs0)))
    b = np.zeros(P, dtype=np.int32)
    for i in range(P):
        sl = slice(off[i], off[i + 1])
        g = oracle.Sparse(op, max_bv)
        g.add_measurements(x0[sl], x1[sl], y[:, sl], None if perm is None else perm[sl])
        f[i], s[i] = g.predict(xs0, xs1)
        b[i] = g.size()
    return f, s, b


@pytest.mark.parametrize("ny,cap", [(1, 16), (3, 10), (1, 40)])
def test_sparse_batch_vs_oracle(gp, oracle, ny, cap):
    capi, ctx = gp
    res = 0.15
    P, n = 23, 120
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=30 + cap, ragged=True, ny=ny)
    perm = synth.sattolo_perms(off, seed=5)
    xs0, xs1 = synth.grid(res, 10)
    kw = dict(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4 if ny == 1 else 1.0, capacity=cap)
    p = capi.default_params_sparse(ny, **kw)
    op = oracle.sparse_params(ny, p0=kw["sigmaf_sq"], p1=kw["l_sq"], s20=kw["noise"], capacity=cap)
    g = capi.Sparse(ctx, p, P, ny)
    st = g.add(off, x0, x1, y, perm)
    f, s, st2 = g.predict(xs0, xs1)
    fo, so, bo = _oracle_batch(oracle, op, off, x0, x1, y, perm, xs0, xs1, cap + 2)
    assert np.all(st == 0)
    assert np.array_equal(g.sizes(), bo)
    scale = np.max(np.abs(fo))
    assert np.max(np.abs(f - fo)) <= 2e-5 * scale
    assert np.max(np.abs(s - so)) <= 2e-5 * np.max(so)
    # confidence form (src/sparse_gp.hpp:340-345) and the mean-only call the compressor makes
    f2, c2, _ = g.predict(xs0, xs1, conf=True)
    kss = kw["sigmaf_sq"] + kw["noise"]
    assert np.allclose(c2, 100.0 * (1.0 - s ** 2 / kss), rtol=0, atol=1e-9)       # the confidence form of the same sigma
    assert np.allclose(c2, 100.0 * (1.0 - so ** 2 / kss), rtol=0, atol=2e-3)     # vs the oracle: 100 x the 2e-5 above
    f3, none, _ = g.predict(xs0, xs1, want_sigma=False)
    assert none is None and np.array_equal(f3, f)
    g.close()


def test_sparse_online_growth_equals_one_shot(gp, oracle):
    """gp_mapping::train_processes keeps calling add_measurements on trained GPs (src/gp_mapping.cpp:338-339):
    four chunks of 64 give the same state as one call with the concatenated order (BASELINE config 4 shape)."""
    capi, ctx = gp
    res, P, n, cap = 0.15, 64, 256, 50          # 64 patches: the distance to the exact recursion below is heavy-tailed
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=44)
    xs0, xs1 = synth.grid(res, 8)
    p = capi.default_params_sparse(1, sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4, capacity=cap)
    g1 = capi.Sparse(ctx, p, P, 1)
    g1.add(off, x0, x1, y)
    f1, s1, _ = g1.predict(xs0, xs1)
    g2 = capi.Sparse(ctx, p, P, 1)
    for c in range(4):
        idx = np.concatenate([np.arange(off[i] + 64 * c, off[i] + 64 * (c + 1)) for i in range(P)])
        coff = (np.arange(P + 1) * 64).astype(np.int32)
        g2.add(coff, x0[idx], x1[idx], y[:, idx])
    f2, s2, _ = g2.predict(xs0, xs1)
    assert np.array_equal(g1.sizes(), g2.sizes()) and np.all(g1.sizes() == cap)
    assert np.array_equal(f1, f2) and np.array_equal(s1, s2)
    op = oracle.sparse_params(1, p0=1.0, p1=(res / 8) ** 2, s20=1e-4, capacity=cap)
    fo, so, bo = _oracle_batch(oracle, op, off, x0, x1, y, None, xs0, xs1, cap + 2)
    assert np.array_equal(g1.sizes(), bo)
    # the tolerance in the arbiter's terms (test_sparse_gpu_vs_arbiter): in this basis-filling regime |Q| grows large and the two
    # fp64 implementations sit a few 1e-5 of max|f*| from the exact (binary128) recursion, each on its own side -- the GPU must be
    # as close to it as the CPU oracle is, not close to the CPU oracle
    ident = np.concatenate([np.arange(off[i + 1] - off[i]) for i in range(P)]).astype(np.int32)
    f_hp, _, b_hp = _arbiter_batch(oracle, dict(p0=1.0, p1=(res / 8) ** 2, s20=1e-4, capacity=cap), off, x0, x1, y, ident, xs0, xs1, cap)
    sc = np.max(np.abs(f_hp), axis=1, keepdims=True)
    e_gpu, e_orc = np.max(np.abs(f1[:, 0, :] - f_hp) / sc, axis=1), np.max(np.abs(fo[:, 0, :] - f_hp) / sc, axis=1)     # per patch
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    print(f"online growth: |f - f_exact| / max|f_exact| per patch: GPU rms {rms(e_gpu):.2e} max {e_gpu.max():.2e}; "
          f"CPU oracle rms {rms(e_orc):.2e} max {e_orc.max():.2e}")
    assert np.array_equal(g1.sizes(), b_hp)
    # the same statement as test_sparse_gpu_vs_arbiter makes (RMS and median over the patches: medians ~3e-7, the worst patch of 64
    # ~1e-5 .. 1e-4 for the GPU and for the CPU oracle alike, tools/sparse_vs_arbiter.py -- the error is the conditioning of Q
    # times the rounding of ONE summation order, and it is not the same patch for the two)
    # The worst patch is bounded against the oracle's worst too (tests/sparse_parity.py MAX_FACTOR), not by a constant: a kernel
    # change that makes the single worst patch 10x worse than the CPU's worst fails here.
    assert rms(e_gpu) <= 3.0 * rms(e_orc) + 2e-6 and np.median(e_gpu) <= 3.0 * np.median(e_orc) + 2e-6
    # ... with an absolute ceiling beside it: the relative bound follows whatever the CPU oracle's outlier happens to
    # be; measured worst patches of this batch: GPU 1e-5 .. 1e-4 over the builds of rounds 2-3, the oracle the same
    assert e_gpu.max() <= min(10.0 * e_orc.max() + 2e-6, 1e-3), (e_gpu.max(), e_orc.max())
    # reset() (src/sparse_gp.hpp:573-582)
    g2.reset()
    assert np.all(g2.sizes() == 0)
    f0, s0, _ = g2.predict(xs0, xs1)
    assert np.all(f0 == 0) and np.allclose(s0, np.sqrt(1.0 + 1e-4))
    g1.close(); g2.close()


def test_sparse_known_answers_and_empty(gp):
    """First point closed form (src/sparse_gp.hpp:100-114), b == 0 prediction (:321-327), ragged batch with empty patches."""
    capi, ctx = gp
    p = capi.default_params_sparse(1)
    off = np.array([0, 0, 1, 1], dtype=np.int32)
    g = capi.Sparse(ctx, p, 3, 1)
    st = g.add(off, np.array([0.01]), np.array([-0.02]), np.array([[0.7]]))
    alpha, C, Q, BV = g.state()
    s20 = float(np.float32(1e-1))
    assert g.sizes().tolist() == [0, 1, 0] and st.tolist() == [0, 0, 0]
    assert alpha[1, 0, 0] == 0.7 / (100.0 + s20) and C[1, 0, 0] == -1.0 / (100.0 + s20) and Q[1, 0, 0] == 1.0 / 100.0
    assert BV[1, 0].tolist() == [0.01, -0.02]
    f, s, _ = g.predict(np.array([0.0, 0.1]), np.array([0.0, -0.1]))
    assert np.all(f[0] == 0) and np.all(s[0] == np.sqrt(100.0 + s20)) and np.all(f[2] == 0)
    g.close()
    with pytest.raises(capi.GpcError):
        capi.Sparse(ctx, capi.default_params_sparse(1, capacity=0), 1, 1)
    with pytest.raises(capi.GpcError):
        capi.Sparse(ctx, capi.default_params_sparse(1), 1, 2)


def test_sparse_defaults_regime_batch(gp, oracle):
    """Reference defaults over a batch: BV sets stay tiny, predictions agree with the oracle to the loose tolerance
    the regime allows, and reconstruct the training surface about as well as the oracle does."""
    capi, ctx = gp
    P, n = 64, 128
    off, x0, x1, y = synth.make_patches(P, n, seed=50)
    perm = synth.sattolo_perms(off, seed=6)
    xs0, xs1 = synth.grid(0.15, 10)
    g = capi.Sparse(ctx, capi.default_params_sparse(1), P, 1)
    g.add(off, x0, x1, y, perm)
    f, s, _ = g.predict(xs0, xs1)
    fo, so, bo = _oracle_batch(oracle, oracle.sparse_params(1), off, x0, x1, y, perm, xs0, xs1, 130)
    assert g.sizes().max() <= 40 and bo.max() <= 40
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    assert rms(f - fo) <= 2e-2 * max(rms(fo), 1e-6)
    g.close()


def test_sparse_predict_points_vs_oracle_and_grid_entry(gp, oracle):
    """gpc_sparse_predict_points = predict_measurements with every patch on its own (ragged) point set, what the reference's
    training-set RMS block does (src/gp_compressor.cpp:303-315).  Against the oracle, and bit for bit against the shared-grid entry
    called with one patch's points as the grid; empty patches; sigma and the 3-channel variant."""
    capi, ctx = gp
    for ny, cap in ((1, 20), (3, 12)):
        P, n, res = 9, 70, 0.15
        off, x0, x1, y = synth.make_patches(P, n, res=res, seed=90 + ny, ragged=True, ny=ny)
        off = off.copy()
        off[4:] -= off[4] - off[3]                                   # patch 3 is empty
        N = int(off[-1])
        x0, x1, y = x0[:N].copy(), x1[:N].copy(), np.ascontiguousarray(y[:, :N])
        kw = dict(sigmaf_sq=1.0, l_sq=(res / 4) ** 2, noise=1e-3, capacity=cap)
        g = capi.Sparse(ctx, capi.default_params_sparse(ny, **kw), P, ny)
        g.add(off, x0, x1, y)
        ft, st_, stat = g.predict_points(off, x0, x1, want_sigma=True)
        assert ft.shape == (ny, N) and np.all(stat == 0) and np.all(np.isfinite(ft)) and np.all(np.isfinite(st_))
        op = oracle.sparse_params(ny, p0=1.0, p1=kw["l_sq"], s20=1e-3, capacity=cap)
        xs0, xs1 = synth.grid(res, 4)
        _, _, bo, fto = oracle.sparse_fit_predict_batch(op, off, x0, x1, y, xs0, xs1, train=True)
        assert np.array_equal(g.sizes(), bo)
        assert np.max(np.abs(ft - fto)) <= 2e-5 * np.max(np.abs(fto))
        for i in (0, 5, P - 1):
            sl = slice(off[i], off[i + 1])
            fg, sg, _ = g.predict(x0[sl], x1[sl])                    # shared-grid entry, X* = patch i's own points
            assert np.array_equal(fg[i], ft[:, sl]) and np.array_equal(sg[i], st_[sl])
        g.close()


@pytest.mark.parametrize("ny,regime", [(1, "defaults"), (3, "defaults"), (1, "mid"), (1, "conf")])
def test_sparse_small_basis_predict_kernel(gp, oracle, ny, regime, monkeypatch):
    """Round 4: patches with at most 32 basis vectors are predicted by sparse_predict_small_kernel (one wave per patch, a lane per
    grid point, k in registers, C by broadcast from LDS) -- the shape of the reference's default regime, where predict_measurements
    computes sigma for a basis of ~13 (src/sparse_gp.hpp:299-351).  Against sparse_predict_kernel on the same states
    (GPC_SPARSE_NO_SMALL_PREDICT=1): the mean is the same BIT FOR BIT (same operations, same order), sigma^2 agrees to rounding of its
    largest term (its own summation order of k^T C k), status words equal -- grid entry and per-patch-points entry, empty patches,
    patches on either side of the 16- and 32-vector boundaries, three channels, the confidence form; and sigma against the oracle."""
    capi, ctx = gp
    res, P, n = 0.15, 300, 96
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=120 + ny, ragged=True, ny=ny, n_min=(6 if regime == "mid" else None))
    off = off.copy()
    off[8:] -= off[8] - off[7]                                       # patch 7 is empty (b = 0)
    N = int(off[-1])
    x0, x1, y = x0[:N].copy(), x1[:N].copy(), np.ascontiguousarray(y[:, :N])
    if regime == "mid":                                               # bases of 10 .. 45: both instances and the regular kernel
        kw = dict(sigmaf_sq=1.0, l_sq=(res / 3.5) ** 2, noise=1e-3, capacity=45, eps_tol=1e-5)
    else:
        kw = dict(capacity=200)
    prm = capi.default_params_sparse(ny, **kw)
    g = capi.Sparse(ctx, prm, P, ny)
    g.add(off, x0, x1, y, synth.sattolo_perms(off, seed=9))
    bv = g.sizes()
    # (the colour GP at its defaults -- s20 = 100 -- keeps 6 .. 7 vectors; the depth GP 8 .. 41; "mid": 10 .. 45)
    assert bv[7] == 0 and (ny == 3 or bv.max() > 16) and (regime != "mid" or (bv.max() > 32 and bv[bv > 0].min() <= 16))
    xs0, xs1 = synth.grid(res, 20)
    conf = regime == "conf"
    out = []
    for small in (True, False):
        if not small:
            monkeypatch.setenv("GPC_SPARSE_NO_SMALL_PREDICT", "1")
        f, sg, st = g.predict(xs0, xs1, conf=conf)
        ft, sgt, stt = g.predict_points(off, x0, x1, want_sigma=True, conf=conf)
        fm, _, _ = g.predict(xs0, xs1, want_sigma=False)
        out.append((f, sg, st, ft, sgt, stt, fm))
    a, b_ = out
    for q in (0, 2, 3, 5, 6):
        assert np.array_equal(a[q], b_[q], equal_nan=True), q      # means and status words: bit for bit
    assert np.array_equal(a[0], a[6])                               # the mean does not depend on whether sigma is asked for
    kk = prm.sigmaf_sq + prm.noise
    # sigma: two summation orders of k^T C k.  At the reference's defaults |C| reaches 1e4 .. 1e6 and the sum cancels from ~1e10 down to
    # ~1e2, so the two orders legitimately differ by up to ~1e-4 in sigma^2: each is held against an extended-precision evaluation of
    # s20 + k* + k^T C k on the GPU's own state, within a rounding bound on the sum of the magnitudes of its terms.
    al_, Cs, Qs, BVs = g.state()
    sample = [i for i in range(0, P, 13) if bv[i] > 0] + [int(np.argmax(bv))]
    for i in sample:
        nb = int(bv[i])
        Bv = BVs[i][:nb].astype(np.longdouble)
        Cm = Cs[i][:nb, :nb].astype(np.longdouble)
        d0 = xs0.astype(np.longdouble)[None, :] - Bv[:, 0][:, None]
        d1 = xs1.astype(np.longdouble)[None, :] - Bv[:, 1][:, None]
        K = np.longdouble(prm.sigmaf_sq) * np.exp(np.longdouble(-0.5) / np.longdouble(prm.l_sq) * (d0 * d0 + d1 * d1))
        exact = np.longdouble(prm.noise) + np.longdouble(prm.sigmaf_sq) + np.sum(K * (Cm @ K), axis=0)
        mag = np.sum(np.abs(K) * (np.abs(Cm) @ np.abs(K)), axis=0).astype(np.float64)
        bound = 256 * np.finfo(np.float64).eps * (mag + kk) * max(nb, 4)      # (incl. the <= 2 ulp of the device's exp in every k)
        ex = np.maximum(exact.astype(np.float64), 0.0)
        for arr in (a[1][i], b_[1][i]):
            if conf:
                s2 = (1.0 - arr / 100.0) * kk                                  # back from 100 (1 - sigma^2 / (k* + s20))
            else:
                s2 = arr ** 2
            assert np.all(np.abs(s2 - ex) <= bound + 1e-12 * kk), (i, float(np.max(np.abs(s2 - ex) / bound)))
    # against the oracle (sigma is sqrt(s20 + k* + k^T C k): compare squares, the cancellation is the oracle's too)
    op = oracle.sparse_params(ny, p0=prm.sigmaf_sq, p1=prm.l_sq, s20=prm.noise, eps_tol=prm.eps_tol, capacity=kw["capacity"])
    if regime == "mid":
        worst, same = 0.0, 0
        for i in range(0, P, 17):
            sl = slice(off[i], off[i + 1])
            h = oracle.Sparse(op, kw["capacity"] + 2)
            h.add_measurements(x0[sl], x1[sl], y[:, sl], synth.sattolo_perms(off, seed=9)[sl])
            fo, so = h.predict(xs0, xs1)
            if h.size() != bv[i]:
                continue                                              # (a gamma within rounding of eps_tol fell the other way)
            same += 1
            worst = max(worst, float(np.max(np.abs(a[1][i] ** 2 - so ** 2))) / kk)
        print(f"small-basis predict vs oracle [mid]: {same} patches with equal basis size, worst |sigma^2 - oracle| / (k* + s20) = {worst:.2e}")
        # (a sanity bound -- the states themselves differ at 1e-9 .. 1e-8 relative and |C| reaches 1 / s20 = 1e3; the rigorous statement
        # about the kernel is the extended-precision one above)
        assert same >= 8 and worst <= 1e-4, (same, worst)
    g.close()


def test_sparse_c4_defaults_full_size_parity(gp, oracle):
    """BASELINE config 4 at the reference's DEFAULT hyper-parameters (src/sparse_gp.h:48, src/rbf_kernel.h:24) -- the production
    regime, where `gamma < eps_tol` (src/sparse_gp.hpp:144-163) is decided by rounding noise -- at the full 32768 x 256 size, 4 add
    calls.  The parity statement is tests/sparse_parity.py's: (a) reconstruction RMSE against the training targets for the GPU, the
    fp64 oracle and the binary128 arbiter ("matched RMSE"), (b) per-patch error against the arbiter as percentiles, (c) predictions
    that leave the data range -- and the GPU may not be worse than the fp64 oracle by more than the frozen factors.
    Measured (round 3, gpurun_out/r3/c4_parity_defaults.json; 1024-patch arbiter sample, oracle on all 32768):
      rmse_train  GPU 4.113e-3  oracle 4.098e-3  arbiter 4.098e-3   (y rms 6.14e-3)
      |f* - f*_exact| per patch  p50 / p90 / p99 / max:  GPU 1.3e-5 / 1.5e-4 / 1.5e-3 / 1.9e-2   oracle 1.6e-5 / 1.9e-4 / 1.7e-3 / 1.4e-2
      max|f*| > 5 max|y|:  GPU 4 of 32768 (worst: patch 21588, 21.6x -- the 0.32 outlier of round 2's bench line),  oracle 1 of 32768
      (patch 16515, 11.9x), arbiter 0: on every such patch the OTHER fp64 implementation and the exact recursion stay below 0.6x,
      i.e. the blow-ups are rounding artefacts of one summation order each, and they happen to the CPU restatement too."""
    import sparse_parity as SP
    capi, ctx = gp
    res, sz, P, n, cap, chunks = 0.15, 20, 32768, 256, 200, 4
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=4)
    prm = capi.default_params_sparse(1, capacity=cap)
    g = capi.Sparse(ctx, prm, P, 1)
    cn = n // chunks
    coff = (np.arange(P + 1) * cn).astype(np.int32)
    for c in range(chunks):
        idx = (off[:-1, None] + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
        assert np.all(g.add(coff, x0[idx], x1[idx], y[:, idx]) == 0)
    xs0, xs1 = synth.grid(res, sz)
    f, _, st2 = g.predict(xs0, xs1, want_sigma=False)
    ft, _, st3 = g.predict_points(off, x0, x1)
    assert np.all(st2 == 0) and np.all(st3 == 0) and np.all(np.isfinite(f)) and np.all(np.isfinite(ft))
    bv = g.sizes()
    assert 8 <= bv.mean() <= 25 and bv.max() <= 64                 # the basis stays tiny whatever the capacity
    op = oracle.sparse_params(1, p0=prm.sigmaf_sq, p1=prm.l_sq, s20=prm.noise, eps_tol=prm.eps_tol, capacity=cap)
    st = SP.stats(op, off, x0, x1, y, xs0, xs1, f, ft, np.arange(1024), full_oracle=True)
    print("C4 defaults, full size:", {k: st[k] for k in ("rmse_train", "err_vs_arbiter_abs", "blowups", "gate")})
    assert st["gate"]["ok"], st["gate"]["why"]
    g.close()


def test_sparse_c4_defaults_blowup_rate_pooled(gp, oracle):
    """(c) of tests/sparse_parity.py as a RATE: patches whose prediction leaves the data range, counted over
    four seeded batches of 32768 x 256 at the reference's default hyper-parameters for the GPU and the fp64 oracle on the same patches.
    One batch shows 0 .. 9 of them per implementation (Poisson noise); the pooled counts are gated -- gpu <= 3 x oracle + 2 and no
    evidence at the 0.1 % level that the GPU's rate exceeds the oracle's -- and every named patch is checked against the binary128
    arbiter, which has none.  Measured over 8 batches: GPU 3.6, oracle 5.4 per 32768 (profiles/r04_blowup_rate.json)."""
    import sparse_parity as SP
    capi, ctx = gp
    res, sz, P, n, chunks, cap = 0.15, 20, 32768, 256, 4, 200
    prm = capi.default_params_sparse(1, capacity=cap)
    op = oracle.sparse_params(1, p0=prm.sigmaf_sq, p1=prm.l_sq, s20=prm.noise, eps_tol=prm.eps_tol, capacity=cap)
    xs0, xs1 = synth.grid(res, sz)

    def run_gpu(off, x0, x1, y):
        g = capi.Sparse(ctx, prm, P, 1)
        cn = n // chunks
        coff = (np.arange(P + 1) * cn).astype(np.int32)
        for c in range(chunks):
            idx = (off[:-1, None] + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
            assert np.all(g.add(coff, x0[idx], x1[idx], y[:, idx]) == 0)
        f, _, st = g.predict(xs0, xs1, want_sigma=False)
        g.close()
        assert np.all(st == 0) and np.all(np.isfinite(f))
        return f
    out = SP.blowup_counts(run_gpu, op, P, n, [21, 22, 23, 24], res, sz, synth)
    print("blow-up rate:", {k: out[k] for k in ("gpu", "oracle", "two_sample", "pooled_gate", "arbiter_blowups_on_named_patches")})
    assert out["pooled_gate"]["ok"], out
    assert out["arbiter_blowups_on_named_patches"] == 0            # rounding artefacts of ONE summation order each, not a property of the data
    assert out["gpu"]["count"] <= 40                               # absolute ceiling: 4 batches at 2.5 x the measured rate


def _closed_form_likelihood(p0, p1, s20, alpha, Cm, BV, q0, q1, yq):
    """likelihood / likelihood_dx (src/sparse_gp.hpp:387-427, 463-508; field .hpp:322-392) written with NumPy on a given state"""
    ny = alpha.shape[0]
    Xq = np.stack([q0, q1], 1)
    D = Xq[None, :, :] - BV[:, None, :]
    K = p0 * np.exp(-0.5 / p1 * np.sum(D * D, axis=2))
    mu = alpha @ K
    CK = Cm @ K
    sigma = s20 + p0 + np.sum(K * CK, axis=0)
    offs = yq - mu
    sq = np.sum(offs * offs, axis=0)
    lik = 1.0 / np.sqrt((2 * np.pi) ** ny * sigma) * np.exp(-0.5 / sigma * sq)
    Kdx = -(1.0 / p1) * D * K[:, :, None]
    sdx = 2.0 * np.einsum("imd,im->md", Kdx, CK)
    exppart = 0.5 / sigma ** 1.5 * np.exp(-0.5 / sigma * sq)
    second = 2.0 * np.einsum("imd,ci,cm->md", Kdx, alpha, offs)
    d12 = exppart[:, None] * (-sdx + second + sdx / sigma[:, None] * sq[:, None])
    d0 = -1.0 / sigma ** 1.5 * offs[0] * exppart if ny == 1 else np.zeros_like(sigma)
    return np.concatenate([d0[:, None], d12], axis=1), lik


@pytest.mark.parametrize("ny,cap", [(1, 24), (3, 12)])
def test_sparse_likelihood_and_derivatives(gp, oracle, ny, cap):
    """gpc_sparse_likelihood = compute_derivatives + compute_likelihoods, batched and ragged.
    (i) against the closed form evaluated on the GPU's OWN state: 1e-8 (isolates this kernel from the 1e-6-level
    differences two fp64 implementations of the training recursion show, see the module docstring);
    (ii) against the oracle end to end: the loose tolerance of the state;  (iii) empty patches, empty point sets."""
    capi, ctx = gp
    res = 0.15
    P, n = 9, 90
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=60 + cap, ragged=True, ny=ny)
    perm = synth.sattolo_perms(off, seed=6)
    kw = dict(sigmaf_sq=1.0, l_sq=(res / 5) ** 2, noise=1e-3 if ny == 1 else 1.0, capacity=cap)
    p = capi.default_params_sparse(ny, **kw)
    g = capi.Sparse(ctx, p, P, ny)
    # patch 3 stays untrained (b = 0): feed it an empty point set
    off_t = off.copy()
    n3 = off[4] - off[3]
    off_t[4:] -= n3
    keep = np.r_[0:off[3], off[4]:off[-1]]
    st = g.add(off_t, x0[keep], x1[keep], y[:, keep], perm[keep])
    assert np.all(st == 0) and g.sizes()[3] == 0
    # query sets: ragged, patch 5 empty
    rng = np.random.default_rng(9)
    cnt = rng.integers(1, 70, P)
    cnt[5] = 0
    qoff = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    N = int(qoff[-1])
    q0, q1 = rng.uniform(-res / 2, res / 2, N), rng.uniform(-res / 2, res / 2, N)
    yq = rng.normal(0, 0.02 if ny == 1 else 30.0, (ny, N))
    dX, l = g.likelihood(qoff, q0, q1, yq)
    assert np.all(np.isfinite(dX)) and np.all(np.isfinite(l))
    alpha, Cm, Q, BV = g.state()
    sizes = g.sizes()
    for i in range(P):
        sl = slice(qoff[i], qoff[i + 1])
        if sl.stop == sl.start:
            continue
        b = int(sizes[i])
        dref, lref = _closed_form_likelihood(kw["sigmaf_sq"], kw["l_sq"], kw["noise"], alpha[i][:, :b], Cm[i][:b, :b], BV[i][:b],
                                             q0[sl], q1[sl], yq[:, sl])
        # sigma = s20 + k** + k^T C k cancels to ~s20 (1e-3 of its terms) and dX goes like sigma^-3
        assert np.max(np.abs(l[sl] - lref)) <= 1e-8 * np.max(np.abs(lref)), i
        assert np.max(np.abs(dX[sl] - dref)) <= 1e-8 * max(np.max(np.abs(dref)), 1e-300), i
        if b == 0:
            assert np.all(dX[sl, 1:] == 0.0)
    # end to end against the oracle (same insertion order)
    op = oracle.sparse_params(ny, p0=kw["sigmaf_sq"], p1=kw["l_sq"], s20=kw["noise"], capacity=cap)
    for i in (0, 3, 7):
        go = oracle.Sparse(op, cap + 2)
        a, b_ = off_t[i], off_t[i + 1]
        if b_ > a:
            go.add_measurements(x0[keep][a:b_], x1[keep][a:b_], y[:, keep][:, a:b_], perm[keep][a:b_])
        sl = slice(qoff[i], qoff[i + 1])
        do, lo = go.likelihood(q0[sl], q1[sl], yq[:, sl])
        assert np.max(np.abs(l[sl] - lo)) <= 1e-4 * np.max(np.abs(lo))
        assert np.max(np.abs(dX[sl] - do)) <= 1e-4 * np.max(np.abs(do))
    # only one of the outputs
    d2, none = g.likelihood(qoff, q0, q1, yq, want_l=False)
    assert none is None and np.array_equal(d2, dX)
    g.close()


def test_sparse_full_size_c4(gp, oracle):
    """BASELINE config 4 at full size: 32768 patches, capacity 200 (21 GB of per-patch state), 256 points per patch
    streamed in 4 online chunks.  Size-independent properties (basis grows monotonically up to capacity, predictions finite) plus the oracle on a sample of patches (same explicit insertion order)."""
    capi, ctx = gp
    res, sz, P, n, cap, chunks = 0.15, 20, 32768, 256, 200, 4
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=4)
    kw = dict(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4, capacity=cap)
    g = capi.Sparse(ctx, capi.default_params_sparse(1, **kw), P, 1)
    cn = n // chunks
    coff = (np.arange(P + 1) * cn).astype(np.int32)
    prev = np.zeros(P, np.int32)
    for c in range(chunks):
        idx = (off[:-1, None] + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
        st = g.add(coff, x0[idx], x1[idx], y[:, idx])
        assert np.all(st == 0)
        b = g.sizes()
        want = min((c + 1) * cn, cap)                       # most points are full updates; the rest are projected
        assert np.all(b <= want) and np.all(b >= prev) and b.mean() >= 0.85 * want
        prev = b.copy()
    xs0, xs1 = synth.grid(res, sz)
    f, _, st2 = g.predict(xs0, xs1, want_sigma=False)
    assert np.all(st2 == 0) and np.all(np.isfinite(f))
    op = oracle.sparse_params(1, p0=kw["sigmaf_sq"], p1=kw["l_sq"], s20=kw["noise"], capacity=cap)
    # 200 basis vectors on a patch 8 length-scales wide make C and Q nearly singular: the reference arithmetic itself
    # moves by 1e-3..1e-2 of the output scale when the inputs move by one ulp, so the bound on each sampled patch is
    # the oracle's own sensitivity to a +-1e-15 relative input perturbation (floor 2e-5, the well-conditioned bound).
    for i in (0, 12345, P - 1):
        sl = slice(off[i], off[i + 1])
        outs = []
        for eps in (0.0, 1e-15, -1e-15):
            go = oracle.Sparse(op, cap + 2, fast=True)
            go.add_measurements(x0[sl] * (1 + eps), x1[sl] * (1 - eps), y[:, sl])   # identity order = the chunked order
            outs.append(go.predict(xs0, xs1)[0])
            if eps == 0.0:
                assert go.size() == int(g.sizes()[i])
        scale = np.max(np.abs(outs[0]))
        sens = max(np.max(np.abs(o - outs[0])) for o in outs[1:])
        assert np.max(np.abs(f[i] - outs[0])) <= max(2e-5 * scale, 5.0 * sens), (i, sens / scale)
    g.close()


@pytest.mark.parametrize("regime", ["unit_amplitude", "small_amplitude", "early_stop"])
def test_sparse_train_sigmaf(gp, oracle, regime):
    """gpc_sparse_train_sigmaf = the live part of sparse_gp::train_parameters, per patch on the
    device.  (i) against the NumPy restatement on the GPU's OWN state: 1e-9 on the trained parameter, 1e-8 on the last gradient,
    1e-7 on the likelihood trace (cancellation in sigma), iteration counts equal; (ii) against the oracle end to end at the tolerance of the state;
    (iii) fewer than 20 basis vectors -> untouched (src/sparse_gp.hpp:609-611), empty point set, argument checks."""
    capi, ctx = gp
    res, P, n, cap = 0.15, 7, 200, 60
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=71, ragged=True)
    y = y[0]
    if regime == "unit_amplitude":          # 102 iterations: the gradient norm never falls below 1e-2 (p(1) is not moved)
        kw, step, maxc = dict(sigmaf_sq=1.0, l_sq=(res / 4) ** 2, noise=1e-3, capacity=cap), float(np.float32(1e-4)), 100
    elif regime == "small_amplitude":       # sigma_f^2 of the order of the signal: the parameter moves by tens of percent
        kw, step, maxc = dict(sigmaf_sq=0.002, l_sq=(res / 4) ** 2, noise=1e-5, capacity=cap), 6e-6, 100
    else:                                   # max_counter cuts the loop: counter > 3 -> 5 iterations
        kw, step, maxc = dict(sigmaf_sq=1.0, l_sq=(res / 4) ** 2, noise=1e-3, capacity=cap), float(np.float32(1e-4)), 3
    # patch 2: 12 points only -> fewer than 20 basis vectors; patch 4: trained, but evaluated on an empty point set
    cnt = np.diff(off)
    keep = np.ones(off[-1], bool)
    keep[off[2] + 12:off[3]] = False
    cnt_t = cnt.copy()
    cnt_t[2] = 12
    off_t = np.concatenate([[0], np.cumsum(cnt_t)]).astype(np.int32)
    g = capi.Sparse(ctx, capi.default_params_sparse(1, **kw), P, 1)
    st = g.add(off_t, x0[keep], x1[keep], y[None, keep])
    assert np.all(st == 0)
    sizes = g.sizes()
    assert sizes[2] < 20 and np.all(np.delete(sizes, 2) >= 20)
    qkeep = keep.copy()
    qkeep[off[4]:off[5]] = False
    cnt_q = cnt_t.copy()
    cnt_q[4] = 0
    qoff = np.concatenate([[0], np.cumsum(cnt_q)]).astype(np.int32)
    q0, q1, yq = x0[qkeep], x1[qkeep], y[qkeep]
    p0, iters, ls, delta = g.train_sigmaf(qoff, q0, q1, yq, step=step, max_counter=maxc)
    alpha, Cm, Q, BV = g.state()
    moved = 0.0
    for i in range(P):
        sl = slice(qoff[i], qoff[i + 1])
        b = int(sizes[i])
        pr, ir, lr, dr = train_sigmaf_np(kw["sigmaf_sq"], kw["l_sq"], kw["noise"], alpha[i][0, :b], Cm[i][:b, :b], BV[i][:b],
                                          q0[sl], q1[sl], yq[sl], step, maxc)
        assert iters[i] == ir, (i, iters[i], ir)
        assert abs(p0[i] - pr) <= 1e-9 * abs(pr), i
        if ir:
            # sigma = s20 + p + p^2 e^T C e cancels to ~s20 (1e-3 of its terms, see the likelihood test): the MFMA and the
            # NumPy summation orders show at 1e-8 in log(sigma)
            assert np.max(np.abs(ls[i] - lr)) <= 1e-7 * np.max(np.abs(lr)) + 1e-9, i
            assert np.max(np.abs(delta[i] - dr)) <= 1e-8 * np.max(np.abs(dr)) + 1e-12, i
            assert np.all(ls[i, ir:] == 0)
        moved = max(moved, abs(p0[i] / kw["sigmaf_sq"] - 1))
    assert iters[2] == 0 and p0[2] == kw["sigmaf_sq"]
    assert iters[4] == 1 and p0[4] == kw["sigmaf_sq"]             # no points: zero gradient, one pass through the loop
    if regime == "unit_amplitude":
        assert np.all(np.delete(iters, [2, 4]) == 102)
    elif regime == "small_amplitude":
        assert moved > 0.05
    else:
        assert np.all(np.delete(iters, [2, 4]) == 5)
    # end to end against the oracle (same insertion order = identity)
    op = oracle.sparse_params(1, p0=kw["sigmaf_sq"], p1=kw["l_sq"], s20=kw["noise"], capacity=cap)
    for i in (0, 5):
        go = oracle.Sparse(op, cap + 2)
        a, b_ = off_t[i], off_t[i + 1]
        go.add_measurements(x0[keep][a:b_], x1[keep][a:b_], y[None, keep][:, a:b_])
        sl = slice(qoff[i], qoff[i + 1])
        po, io, lo, do = go.train_sigmaf(q0[sl], q1[sl], yq[sl], step=step, max_counter=maxc)
        assert io == iters[i]
        assert abs(p0[i] - po) <= 1e-4 * abs(po)
        assert np.max(np.abs(ls[i] - lo)) <= 1e-4 * np.max(np.abs(lo))
    # argument checks
    with pytest.raises(capi.GpcError) as e:
        g.train_sigmaf(qoff, q0, q1, yq, max_counter=-1)
    assert e.value.code == capi.GPC_EINVAL
    g3 = capi.Sparse(ctx, capi.default_params_sparse(3, capacity=10), 1, 3)
    with pytest.raises(capi.GpcError) as e:
        g3.train_sigmaf(np.array([0, 1], np.int32), np.zeros(1), np.zeros(1), np.zeros(1))
    assert e.value.code == capi.GPC_EINVAL
    g3.close()
    g.close()


@pytest.mark.parametrize("ny,cap,kernel", [(1, 100, "fill"), (3, 60, "fill"), (1, 200, "fill"), (1, 100, "default"), (1, 33, "geo"),
                                            (1, 80, "mixed"), (3, 80, "mixed"), (1, 255, "fill"), (1, -1, "fill")])
def test_sparse_fused_next_matvec_is_bit_identical(gp, ny, cap, kernel, monkeypatch):
    """The full-update passes over C and Q also form the NEXT point's mat-vecs C k', Q k' from the values they store (a third
    less traffic per point).  Same numbers as the stand-alone mat-vec would give: with and without the fusion
    (GPC_SPARSE_NO_FUSE) the states are identical bit for bit -- while the basis grows, when it is full (fused update +
    capacity deletion), with explicit insertion orders, across chunked calls, and when geometric deletions invalidate the
    prefetched products."""
    capi, ctx = gp
    res, P, n = 0.15, 24, (400 if cap == 255 else 256)
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=7 + cap, ragged=True, ny=ny, n_min=(300 if cap == 255 else None))
    perm = synth.sattolo_perms(off, seed=2)
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4 if ny == 1 else 1.0)
    if kernel == "geo":            # long length scale: near-duplicate basis vectors -> geometric deletions right after full updates
        kw.update(sigmaf_sq=1.0, l_sq=(res * 2) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":          # a moderate length scale: sparse (projected) updates and full updates alternate on a mid-sized basis
        kw.update(sigmaf_sq=1.0, l_sq=(res / 3) ** 2, noise=1e-3 if ny == 1 else 1.0, eps_tol=1e-3)
    p = capi.default_params_sparse(ny, **kw)
    results = []
    for no_fuse in (False, True):
        if no_fuse:
            monkeypatch.setenv("GPC_SPARSE_NO_FUSE", "1")
        g = capi.Sparse(ctx, p, P, ny)
        st1 = g.add(off, x0, x1, y, perm)
        st2 = g.add(off, x0, x1, y)
        results.append((st1, st2, g.sizes(), *g.state()))
        g.close()
    a, b_ = results
    assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2])
    if kernel == "geo":
        assert a[2].max() < cap                    # the geometric rule, not the capacity, bounds the basis here
    if kernel == "mixed":
        assert 15 <= np.median(a[2]) < cap         # neither tiny nor saturated: both update forms ran
    if cap == 255:
        assert a[2].max() == 255                   # the largest capacity the kernels take (GPC_MAX_BV - 1)
    if cap == -1:                                  # exact GP: the basis runs into GPC_MAX_BV and further points are refused
        assert a[2].max() == capi.MAX_BV and np.any(a[1] == 4) and np.array_equal(a[1] == 4, a[2] == capi.MAX_BV)
    for i in range(P):
        nb = int(a[2][i])
        (al0, C0, Q0, BV0), (al1, C1, Q1, BV1) = a[3:], b_[3:]
        assert np.array_equal(al0[i][:, :nb], al1[i][:, :nb], equal_nan=True)
        assert np.array_equal(C0[i][:nb, :nb], C1[i][:nb, :nb], equal_nan=True)
        assert np.array_equal(Q0[i][:nb, :nb], Q1[i][:nb, :nb], equal_nan=True)
        assert np.array_equal(BV0[i][:nb], BV1[i][:nb], equal_nan=True)


@pytest.mark.parametrize("ny,cap,kernel", [(1, 64, "fill"), (3, 50, "fill"), (1, 20, "default"), (3, 40, "mixed"), (1, 33, "geo"), (1, 1, "fill"),
                                            (1, 100, "fill"), (3, 100, "default"), (1, 90, "mixed"), (3, 65, "fill")])
def test_sparse_one_wave_per_patch_is_bit_identical(gp, ny, cap, kernel, monkeypatch):
    """capacity <= 64: the add kernel runs one wave per patch (every basis row has its lane; no cross-wave barriers, four times
    as many patches in flight); capacity <= 100: two waves per patch.  The reductions and the quarter-wise mat-vec sums keep
    the layout of the four-wave shape (GPC_SPARSE_WIDE selects it): identical states, bit for bit, in every update regime."""
    monkeypatch.setenv("GPC_SPARSE_FULL", "1")     # the kernel SHAPES are compared in the full mode (the triangular passes of the four-wave shape sum a row in another order)
    capi, ctx = gp
    res, P, n = 0.15, 37, 200
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=3 + cap, ragged=True, ny=ny)
    perm = synth.sattolo_perms(off, seed=9)
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4 if ny == 1 else 1.0)
    if kernel == "geo":
        kw.update(sigmaf_sq=1.0, l_sq=(res * 2) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 3) ** 2, noise=1e-3 if ny == 1 else 1.0, eps_tol=1e-3)
    p = capi.default_params_sparse(ny, **kw)
    results = []
    for wide in (False, True):
        if wide:
            monkeypatch.setenv("GPC_SPARSE_WIDE", "1")
        g = capi.Sparse(ctx, p, P, ny)
        st1 = g.add(off, x0, x1, y, perm)
        st2 = g.add(off, x0, x1, y)
        results.append((st1, st2, g.sizes(), *g.state()))
        g.close()
    a, b_ = results
    assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2])
    for i in range(P):
        nb = int(a[2][i])
        (al0, C0, Q0, BV0), (al1, C1, Q1, BV1) = a[3:], b_[3:]
        assert np.array_equal(al0[i][:, :nb], al1[i][:, :nb], equal_nan=True)
        assert np.array_equal(C0[i][:nb, :nb], C1[i][:nb, :nb], equal_nan=True)
        assert np.array_equal(Q0[i][:nb, :nb], Q1[i][:nb, :nb], equal_nan=True)
        assert np.array_equal(BV0[i][:nb], BV1[i][:nb], equal_nan=True)


@pytest.mark.parametrize("ny,cap,kernel", [(1, 100, "default"), (3, 100, "default"), (1, 100, "fill"), (3, 60, "fill"), (1, 20, "fill"), (1, 32, "fill"),
                                            (1, 200, "fill"), (1, 80, "mixed"), (3, 40, "mixed"), (1, 33, "geo"), (1, -1, "fill"), (1, 255, "default")])
def test_sparse_small_basis_phase_is_bit_identical(gp, ny, cap, kernel, monkeypatch):
    """The add runs in two phases: a small-basis kernel (one wave per patch, C and Q resident in LDS) takes every patch as far as
    24 basis vectors, the regular kernel continues from the point where a patch outgrew it.  Same operations in the same order:
    with and without the first phase (GPC_SPARSE_NO_SMALL) the states, the basis sizes, the per-patch status and the point
    counts are identical -- patches that stay small, patches that cross over in the middle of a call, online growth over
    several calls, empty patches, capacities below and above the block size."""
    monkeypatch.setenv("GPC_SPARSE_FULL", "1")     # the kernel SHAPES are compared in the full mode (the triangular passes of the four-wave shape sum a row in another order)
    capi, ctx = gp
    res, P, n = 0.15, 29, 180
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=11 + cap, ragged=True, ny=ny)
    # an empty patch in the middle
    cnt = np.diff(off)
    keep = np.ones(off[-1], bool)
    keep[off[7]:off[8]] = False
    cnt[7] = 0
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    x0, x1, y = x0[keep], x1[keep], y[:, keep]
    perm = synth.sattolo_perms(off, seed=4)
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4 if ny == 1 else 1.0)
    if kernel == "geo":
        kw.update(sigmaf_sq=1.0, l_sq=(res * 2) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 3) ** 2, noise=1e-3 if ny == 1 else 1.0, eps_tol=1e-3)
    p = capi.default_params_sparse(ny, **kw)
    results = []
    for one_phase in (False, True):
        if one_phase:
            monkeypatch.setenv("GPC_SPARSE_NO_SMALL", "1")
        g = capi.Sparse(ctx, p, P, ny)
        st1 = g.add(off, x0, x1, y, perm)
        st2 = g.add(off, x0, x1, y)                  # online growth: the second call starts from trained states
        results.append((st1, st2, g.sizes(), *g.state()))
        g.close()
    a, b_ = results
    assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2])
    assert a[2][7] == 0 and a[0][7] == 0
    for i in range(P):
        nb = int(a[2][i])
        (al0, C0, Q0, BV0), (al1, C1, Q1, BV1) = a[3:], b_[3:]
        assert np.array_equal(al0[i][:, :nb], al1[i][:, :nb], equal_nan=True)
        assert np.array_equal(C0[i][:nb, :nb], C1[i][:nb, :nb], equal_nan=True)
        assert np.array_equal(Q0[i][:nb, :nb], Q1[i][:nb, :nb], equal_nan=True)
        assert np.array_equal(BV0[i][:nb], BV1[i][:nb], equal_nan=True)


@pytest.mark.parametrize("ny,cap,kernel,P,n", [(1, 200, "default", 203, 256), (3, 100, "default", 61, 200), (1, 12, "default", 37, 150),
                                               (1, 200, "mixed", 50, 180), (3, 40, "mixed", 33, 120), (1, 33, "geo", 29, 180),
                                               (1, 100, "fill", 21, 90), (1, -1, "mixed", 9, 60)])
def test_sparse_rows_phase_is_bit_identical(gp, ny, cap, kernel, P, n, monkeypatch):
    """The add now opens with a ROWS phase -- four patches per wave, one DPP row of 16 lanes each, while a patch needs at most 16
    basis vectors and no deletion (sparse_add_rows_kernel) -- whose unfinished patches go on a work list that the one-wave
    small-basis kernel and the regular kernel take by ticket.  With all of it, with static shares instead of the list
    (GPC_SPARSE_NO_LIST), without the rows phase (GPC_SPARSE_NO_ROWS) and with no early phase at all (GPC_SPARSE_NO_SMALL) the states, basis sizes, status
    words, point counts and the per-point decision bytes are identical: patches that never leave the rows phase (the
    reference's default hyper-parameters), patches handed over in the middle of a call (basis > 16, a geometric deletion due,
    the capacity reached), online growth over two calls, ragged and empty patches, a patch count that is not a multiple of 4."""
    monkeypatch.setenv("GPC_SPARSE_FULL", "1")     # the kernel SHAPES are compared in the full mode (the triangular passes of the four-wave shape sum a row in another order)
    capi, ctx = gp
    res = 0.15
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=23 + cap + P, ragged=True, ny=ny, n_min=1)
    cnt = np.diff(off)
    keep = np.ones(off[-1], bool)
    keep[off[5]:off[6]] = False                      # an empty patch
    cnt[5] = 0
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    x0, x1, y = x0[keep], x1[keep], np.ascontiguousarray(y[:, keep])
    perm = synth.sattolo_perms(off, seed=9)
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4 if ny == 1 else 1.0)
    if kernel == "geo":
        kw.update(sigmaf_sq=1.0, l_sq=(res * 2) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 3) ** 2, noise=1e-3 if ny == 1 else 1.0, eps_tol=1e-3)
    p = capi.default_params_sparse(ny, **kw)
    results = []
    # (GPC_SPARSE_NO_ROWS2: the one-wave small-basis kernel instead of the second rows phase -- two patches per wave, 24 rows of state, by ticket)
    # (GPC_SPARSE_NO_MID: without the one-wave kernel's second instance for bases of 25 .. 48 vectors)
    for env in (None, "GPC_SPARSE_NO_LIST", "GPC_SPARSE_NO_ROWS", "GPC_SPARSE_NO_SMALL", "GPC_SPARSE_NO_ROWS2", "GPC_SPARSE_NO_MID"):
        if env:
            monkeypatch.setenv(env, "1")
        g = capi.Sparse(ctx, p, P, ny)
        st1, tr1 = g.add(off, x0, x1, y, perm, trace=True)
        st2, tr2 = g.add(off, x0, x1, y, trace=True)          # online growth: the second call starts from trained states
        results.append((st1, st2, tr1, tr2, g.sizes(), *g.state()))
        g.close()
        if env:
            monkeypatch.delenv(env)
    a = results[0]
    assert a[4][5] == 0 and a[0][5] == 0
    if kernel == "default":
        assert a[4].max() <= 40                       # the regime the rows phase exists for
    for b_ in results[1:]:
        for q in range(5):
            assert np.array_equal(a[q], b_[q]), q
        for i in range(P):
            nb = int(a[4][i])
            (al0, C0, Q0, BV0), (al1, C1, Q1, BV1) = a[5:], b_[5:]
            assert np.array_equal(al0[i][:, :nb], al1[i][:, :nb], equal_nan=True), i
            assert np.array_equal(C0[i][:nb, :nb], C1[i][:nb, :nb], equal_nan=True), i
            assert np.array_equal(Q0[i][:nb, :nb], Q1[i][:nb, :nb], equal_nan=True), i
            assert np.array_equal(BV0[i][:nb], BV1[i][:nb], equal_nan=True), i


def test_sparse_phases_random_sweep(gp, monkeypatch):
    """tests/sparse_sweep.py: 300 random configurations (channels, capacity from 5 to unbounded, five kernel regimes, 1 .. 89 patches of
    1 .. 199 points, ragged or not, three add calls, the first in a random insertion order) through the default path -- rows phase, second
    rows phase, mid phase, regular kernel over chained work lists -- and through the regular kernel alone: states, basis sizes, status
    words, point counts and decision bytes bit for bit; and the mean on a grid through the small-basis predict kernels and through the regular
    predict kernel: the same bits.  Seed 1 on purpose: its configurations 41 and 147 are the ones that caught the
    divisions sharing one reciprocal (round 4, not kept); 2000 configurations of seeds 1 .. 4 passed on the library as committed."""
    import sparse_sweep as SW
    capi, ctx = gp
    monkeypatch.setenv("GPC_SPARSE_FULL", "1")
    bad, hist = SW.sweep(capi, synth, ctx, 300, 1)
    assert not bad, bad
    assert hist["17_24"] > 1000 and hist["25_48"] > 1000 and hist["gt48"] > 500      # every phase saw patches


@pytest.mark.parametrize("ny,cap,kernel", [(1, 200, "fill"), (3, 200, "mixed"), (1, 150, "mixed"), (1, 200, "geo"), (1, 255, "fill"), (1, 200, "default"),
                                           (1, 100, "fill"), (1, 80, "mixed")])     # (capacity <= 100: the two-wave shape; 80 < the 92 vectors the mixed kernel asks for)
def test_sparse_triangular_mode(gp, oracle, ny, cap, kernel, monkeypatch):
    """The four-wave regular kernel (capacity > 100) works on the LOWER triangles of C and Q -- half the stream of a point -- and
    mirrors them when a patch leaves it (sp_tri_pass).  Against the full passes (GPC_SPARSE_FULL=1): the matrices that come back are
    exactly symmetric; in regimes where the recursion is well conditioned the branch decisions, basis sizes and status words are
    the same and the triangular mode is as close to the CPU oracle as the full mode is (both are summation orders of the same
    recursion: the two modes differ by which of C_ij / C_ji a mat-vec reads and by the order a row is summed in -- bound written
    before the first run: twice the full mode's distance from the oracle + 1e-12, and below 1e-6); in the ill-conditioned regimes (basis-filling kernel, the reference defaults) the modes are two summation
    orders like any others and are held by the parity statistics (tests/sparse_parity.py, test_sparse_full_size_c4,
    test_sparse_c4_defaults_full_size_parity), here only by a loose bound."""
    capi, ctx = gp
    res, P, n = 0.15, 20, (300 if cap in (255, -1) else 256)
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=31 + cap, ragged=True, ny=ny, n_min=(280 if cap in (255, -1) else None))
    perm = synth.sattolo_perms(off, seed=5)
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4)
    if kernel == "geo":            # geometric deletions on bases that reach the four-wave kernel (a shorter length scale than the other geo cases)
        kw.update(sigmaf_sq=1.0, l_sq=(res * 0.6) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":          # well conditioned: noise 1e-2 on a unit-amplitude kernel
        kw.update(sigmaf_sq=1.0, l_sq=(res / 6) ** 2, noise=1e-2 if ny == 1 else 1.0, eps_tol=1e-3)
    p = capi.default_params_sparse(ny, **kw)
    xs0, xs1 = synth.grid(res, 20)
    out = []
    # (the mode is chosen per patch and call from the basis it arrives with, >= 96 vectors by default: here every patch the four-wave
    # kernel takes runs it)
    monkeypatch.setenv("GPC_SPARSE_TRI_MIN", "0")
    # (round 4: bases of 25 .. 48 vectors go through a second instance of the one-wave kernel first -- full matrices, like the small-basis
    # kernel's; this test is about the four-wave kernel's triangular passes at EVERY size it can see, so that instance stays out of it)
    monkeypatch.setenv("GPC_SPARSE_NO_MID", "1")
    for full in (False, True):
        if full:
            monkeypatch.setenv("GPC_SPARSE_FULL", "1")
        g = capi.Sparse(ctx, p, P, ny)
        st1, tr1 = g.add(off, x0, x1, y, perm, trace=True)
        st2, tr2 = g.add(off, x0, x1, y, trace=True)
        f = g.predict(xs0, xs1)[0]
        out.append((st1, st2, tr1, tr2, g.sizes(), f, *g.state()))
        g.close()
    t, u = out
    nbs = t[4]
    if kernel not in ("default", "geo"):
        assert nbs.max() > 24                          # beyond the small-basis kernel: the four-wave shape took these patches
    for i in range(P):
        nb = int(nbs[i])
        C, Q = t[7][i][:nb, :nb], t[8][i][:nb, :nb]
        if nb > 24:                                    # (a smaller basis never left the rows / small-basis kernels, whose matrices are full)
            assert np.array_equal(C, C.T, equal_nan=True) and np.array_equal(Q, Q.T, equal_nan=True), i
    fin = np.isfinite(u[5])                            # (capacity -1: a patch that ran into GPC_MAX_BV predicts NaN in both modes)
    assert np.array_equal(fin, np.isfinite(t[5]))
    scale = max(1e-300, float(np.sqrt(np.mean(u[5][fin] ** 2))))
    err = float(np.sqrt(np.mean((t[5][fin] - u[5][fin]) ** 2))) / scale
    # both modes against the CPU oracle (same two calls, same insertion orders): the triangular mode must be as close to it as the full
    # mode is -- 2 x + 1e-12 where the recursion is well conditioned (and below 1e-6 there), 3 x + 1e-9 elsewhere (in the geo regime,
    # eps_tol = 1e-14 with |Q| up to 1e9, every summation order keeps a different basis and e_full itself is O(1))
    okw = dict(p0=kw.get("sigmaf_sq", 100.0), p1=kw.get("l_sq", 1.0), capacity=cap)
    if "noise" in kw:
        okw["s20"] = kw["noise"]
    if "eps_tol" in kw:
        okw["eps_tol"] = kw["eps_tol"]
    op = oracle.sparse_params(ny, **okw)
    fo = np.zeros_like(t[5])
    for i in range(P):
        sl = slice(off[i], off[i + 1])
        h = oracle.Sparse(op, (cap + 2) if cap > 0 else 2 * n + 2)
        h.add_measurements(x0[sl], x1[sl], y[:, sl], perm[sl])
        h.add_measurements(x0[sl], x1[sl], y[:, sl], None)
        fo[i] = h.predict(xs0, xs1)[0]
    fin &= np.isfinite(fo)
    e_tri = float(np.sqrt(np.mean((t[5][fin] - fo[fin]) ** 2))) / scale
    e_full = float(np.sqrt(np.mean((u[5][fin] - fo[fin]) ** 2))) / scale
    print(f"triangular mode [{ny}-{cap}-{kernel}]: rms vs oracle tri {e_tri:.3e} full {e_full:.3e}; tri vs full {err:.3e}; bv max {nbs.max()}")
    if kernel == "mixed" and nbs.max() < cap:          # (no capacity deletions: with them, near-ties of the scores choose different vectors)
        assert np.array_equal(t[0], u[0]) and np.array_equal(t[1], u[1]) and np.array_equal(t[4], u[4])
        assert np.mean(t[2] != u[2]) < 2e-3 and np.mean(t[3] != u[3]) < 2e-3     # (a gamma within rounding of eps_tol may fall either way)
        assert e_tri <= 2.0 * e_full + 1e-12 and e_tri < 1e-6, (e_tri, e_full, err)
    else:
        assert e_tri <= 3.0 * e_full + 1e-9, (e_tri, e_full, err)

@pytest.mark.parametrize("ny,cap,kernel", [(1, 100, "fill"), (1, 80, "mixed"), (3, 100, "mixed"), (1, 100, "geo"), (1, 120, "fill"), (1, 100, "default"),
                                           (1, 70, "fill")])
def test_sparse_lds_resident_mode_is_bit_identical(gp, ny, cap, kernel, monkeypatch):
    """Round 4: for 64 < capacity <= 120 (the reference's default is 100, /root/reference/src/sparse_gp.h:48) the regular kernel keeps the
    lower triangles of C and Q of the patch in flight packed in LDS (sparse_add_kernel<.., RES>): loaded once per add call, written back
    once.  It applies the element updates of the triangular mode in the same order with the same wave shares, so against the HBM-resident
    triangular mode in the same four-wave shape (GPC_SPARSE_NO_RES=1 GPC_SPARSE_WIDE=1, triangular from the first vector on) the states,
    sizes, status words and per-point decisions are the same BIT FOR BIT -- over two add calls (the second on the state the first left in
    HBM), ragged patches, capacity and geometric deletions, three channels; and the matrices that come back are exactly symmetric."""
    capi, ctx = gp
    res, P, n = 0.15, 24, 256
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=77 + cap, ragged=True, ny=ny)
    perm = synth.sattolo_perms(off, seed=6)
    kw = dict(capacity=cap)
    if kernel == "fill":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 8) ** 2, noise=1e-4)
    if kernel == "geo":
        kw.update(sigmaf_sq=1.0, l_sq=(res * 0.6) ** 2, noise=1e-6, eps_tol=1e-14)
    if kernel == "mixed":
        kw.update(sigmaf_sq=1.0, l_sq=(res / 6) ** 2, noise=1e-2 if ny == 1 else 1.0, eps_tol=1e-3)
    p = capi.default_params_sparse(ny, **kw)
    xs0, xs1 = synth.grid(res, 20)
    monkeypatch.setenv("GPC_SPARSE_TRI_MIN", "0")
    monkeypatch.setenv("GPC_SPARSE_NO_MID", "1")    # (the four-wave shapes at every size they can see: see test_sparse_triangular_mode)
    monkeypatch.setenv("GPC_SPARSE_RES", "1")       # (measured slower than the HBM-resident shape and not the default: DESIGN 5.4c)
    out = []
    for hbm in (False, True):
        if hbm:
            monkeypatch.setenv("GPC_SPARSE_NO_RES", "1")
            monkeypatch.setenv("GPC_SPARSE_WIDE", "1")
        g = capi.Sparse(ctx, p, P, ny)
        st1, tr1 = g.add(off, x0, x1, y, perm, trace=True)
        st2, tr2 = g.add(off, x0, x1, y, trace=True)
        f, sg, _ = g.predict(xs0, xs1)
        out.append((st1, st2, tr1, tr2, g.sizes(), f, sg, *g.state()))
        g.close()
    a, b_ = out
    if kernel in ("fill", "mixed"):
        assert a[4].max() > 32                      # these patches did reach the regular kernel with a large basis
    for q in range(7):
        assert np.array_equal(a[q], b_[q], equal_nan=True), q
    for i in range(P):
        nb = int(a[4][i])
        for q in (7, 8, 9, 10):
            x_, y_ = a[q][i], b_[q][i]
            x_, y_ = (x_[:, :nb], y_[:, :nb]) if q == 7 else (x_[:nb, :nb], y_[:nb, :nb]) if q in (8, 9) else (x_[:nb], y_[:nb])
            assert np.array_equal(x_, y_, equal_nan=True), (i, q)
        if nb > 24:
            C, Q = a[8][i][:nb, :nb], a[9][i][:nb, :nb]
            assert np.array_equal(C, C.T, equal_nan=True) and np.array_equal(Q, Q.T, equal_nan=True), i


# ------------------------------------------------------------------ the tolerances, stated against the exact recursion

def _arbiter_batch(oracle, kw, off, x0, x1, y, perm, xs0, xs1, cap):
    fh, tr, bh = [], [], []
    for i in range(len(off) - 1):
        sl = slice(off[i], off[i + 1])
        h = oracle.SparseHP(oracle.sparse_params(1, **kw), cap + 2)
        tr.append(h.add_measurements(x0[sl], x1[sl], y[:, sl], perm[sl], trace=True))
        fh.append(h.predict(xs0, xs1)[0][0])
        bh.append(h.size())
    return np.array(fh), np.concatenate(tr), np.array(bh)


@pytest.mark.parametrize("regime,kw,P,n", [
    ("well-conditioned", dict(p0=1.0, p1=(0.15 / 4) ** 2, s20=1e-3, capacity=30), 16, 128),
    ("C4 basis-filling, capacity 200", dict(p0=1.0, p1=(0.15 / 8) ** 2, s20=1e-4, capacity=200), 6, 256),
    ("reference defaults", dict(), 48, 256)])
def test_sparse_gpu_vs_arbiter(gp, oracle, regime, kw, P, n):
    """What the sparse tolerances mean.  oracle/gpc_oracle_hp.c runs the same recursion in IEEE binary128; against it
      * the fp64 CPU oracle and the GPU are EQUALLY far from the exact recursion: err_gpu <= 3 err_oracle64 (+ a floor of 2e-6 of
        max|f*|) in RMS over the batch -- well conditioned (both ~1e-6), the C4 kernel that fills a 200-vector basis (both ~1e-3:
        |Q| reaches 1e9 there) and the reference's default hyper-parameters (both ~1e-2);
      * the branch decisions (full vs sparse update, deletions; gpc_sparse_set_trace) each fp64 implementation takes differently
        from the exact recursion are counted: none when well conditioned, a few per cent at the defaults -- for the GPU and
        for the CPU oracle alike, at different points."""
    capi, ctx = gp
    res = 0.15
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=77)
    perm = synth.sattolo_perms(off, seed=8)
    xs0, xs1 = synth.grid(res, 20)
    op = oracle.sparse_params(1, **kw)
    cap = op.capacity
    p = capi.default_params_sparse(1, sigmaf_sq=op.p0, l_sq=op.p1, noise=op.s20, eps_tol=op.eps_tol, capacity=cap)
    g = capi.Sparse(ctx, p, P, 1)
    st, tr_gpu = g.add(off, x0, x1, y, perm, trace=True)
    f_gpu = g.predict(xs0, xs1, want_sigma=False)[0][:, 0, :]
    b_gpu = g.sizes()
    g.close()
    f_hp, tr_hp, b_hp = _arbiter_batch(oracle, kw, off, x0, x1, y, perm, xs0, xs1, cap)
    f_orc, tr_orc = [], []
    for i in range(P):
        sl = slice(off[i], off[i + 1])
        h = oracle.Sparse(op, cap + 2)
        tr_orc.append(h.add_measurements(x0[sl], x1[sl], y[:, sl], perm[sl], trace=True))
        f_orc.append(h.predict(xs0, xs1)[0][0])
    f_orc, tr_orc = np.array(f_orc), np.concatenate(tr_orc)
    scale = np.max(np.abs(f_hp), axis=1, keepdims=True)
    rel = lambda f: np.max(np.abs(f - f_hp) / scale, axis=1)             # per patch, max-norm relative to the exact f*
    e_gpu, e_orc = rel(f_gpu), rel(f_orc)
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    w_gpu, w_orc = float(np.mean(tr_gpu != tr_hp)), float(np.mean(tr_orc != tr_hp))
    print(f"[{regime}] |f - f_exact|/max|f_exact| per patch: GPU rms {rms(e_gpu):.2e} max {e_gpu.max():.2e}; CPU oracle rms {rms(e_orc):.2e} "
          f"max {e_orc.max():.2e}; decisions differing from the exact recursion: GPU {100 * w_gpu:.2f} %, CPU oracle {100 * w_orc:.2f} %; "
          f"basis sizes exact {b_hp.min()}-{b_hp.max()}, GPU {b_gpu.min()}-{b_gpu.max()}")
    assert np.all(st == 0) and np.all(np.isfinite(f_gpu))
    assert rms(e_gpu) <= 3.0 * rms(e_orc) + 2e-6
    assert np.median(e_gpu) <= 3.0 * np.median(e_orc) + 2e-6
    if regime == "well-conditioned":
        assert w_gpu == 0.0 and w_orc == 0.0 and np.array_equal(b_gpu, b_hp) and e_gpu.max() <= 2e-5
    else:
        assert w_gpu <= 3.0 * w_orc + 0.01
