"""CPU model of the index algebra of csrc/dense_mfma.hip (-m "not gpu").

The register-tile kernel never materialises K in memory: tiles live in the v_mfma_f64_16x16x4_f64 accumulator layout,
stored transposed, and move between roles (accumulator -> B operand -> LDS operand image) without any cross-lane
shuffle.  That only works if a handful of lane maps line up; this test replays the kernel's data flow lane by lane in
NumPy -- with the MFMA lane maps measured by tools/probe_mfma_f64.hip on MI355X -- and checks the result against a
plain Cholesky solve.  It is the host-side proof that the kernel's formulas compute (K + 2 sn^2 I)^-1 y and K*^T alpha.

Lane maps (lane l, 64 lanes):  A operand: A[l & 15][l >> 4]   B operand: B[l >> 4][l & 15]
                               C/D register r: D[(l >> 4) + 4 r][l & 15]
"""
import numpy as np

import np_restatement as R
from gp_compressor_amd import synth

L = np.arange(64)
LR, LG = L & 15, L >> 4


def mfma(a, b, c, neg_a=False):
    """v_mfma_f64_16x16x4_f64: a, b (64,), c (64, 4) -> d (64, 4)."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LR, LG] = a
    B[LG, LR] = b
    P = A @ B
    if neg_a:
        P = -P
    d = c.copy()
    for r in range(4):
        d[:, r] += P[LG + 4 * r, LR]
    return d


def tiles_of_wave(NT, wave, waves=7):
    """column-major enumeration of the lower triangle dealt round-robin to the 7 worker waves (the tij[] table of the
    kernel; wave 7 is the factor wave and owns no tiles)."""
    out = []
    ntiles = NT * (NT + 1) // 2
    for t in range((ntiles + waves - 1) // waves):
        idx = t * waves + wave
        if idx >= ntiles:
            continue
        jj = 0
        for j in range(1, NT):
            if idx >= j * NT - (j * (j - 1)) // 2:
                jj = j
        ii = jj + idx - (jj * NT - (jj * (jj - 1)) // 2)
        out.append((ii, jj))
    return out


def test_tile_enumeration_covers_lower_triangle_once():
    for NT in (4, 8, 12, 16):
        seen = [tl for w in range(7) for tl in tiles_of_wave(NT, w)]
        assert sorted(seen) == sorted((i, j) for j in range(NT) for i in range(j, NT))
        # per step k the active tiles (j > k) are spread within +-1 over the waves... of column-major suffixes
        for k in range(NT - 1):
            cnt = [sum(1 for (i, j) in tiles_of_wave(NT, w) if j > k) for w in range(7)]
            assert max(cnt) - min(cnt) <= 1


def img_off(l, s):
    """operand image of csrc/dense_mfma.hip (mf_img_off): two planes of 64 lanes x 2 doubles."""
    return (s >> 1) * 128 + l * 2 + (s & 1)


def img_rc(r, c):
    return img_off(r + 16 * (c & 3), c >> 2)


IMG_LS = np.array([[img_off(l, s) for s in range(4)] for l in range(64)])   # (64, 4) gather index: lane, s -> offset


def diag_factor(Wc):
    """mf_diag_factor: in-place Gauss-Jordan on the MFMA pipe.  Wc (64, 4) is the SPD tile in the C/D layout
    (Wc[l][r] = A[(l>>4) + 4 r][l & 15]).  Pivot c = one rank-1 MFMA whose only non-zero contraction slot is
    k = c & 3: the A operand carries the multipliers of the rows below the pivot, the B operand the pivot row with
    1 added in column c, so that column c of the tile (dead after the elimination) receives column c of the unit
    lower inverse.  The reciprocal of the next pivot is formed one step ahead from two scalars of the current
    tile.  Returns the operand image of L^-1, the C/D-layout registers of L^-1 (= operand image of L^-T) and ok."""
    W = Wc.copy()
    rp = 1.0 / W[0, 0]
    for c in range(15):
        q, r, q1, r1 = c & 3, c >> 2, (c + 1) & 3, (c + 1) >> 2
        inq = LG == q
        s01 = W[16 * q + c + 1, r]              # A^(c)[c][c+1]
        s11 = W[16 * q1 + c + 1, r1]            # A^(c)[c+1][c+1]
        a_op = W[:, r] * np.where(inq & (LR > c), rp, 0.0)
        b_op = W[:, r] * inq + (inq & (LR == c))
        W = mfma(a_op, b_op, W, neg_a=True)
        t = s01 * rp
        rp = 1.0 / (s11 - t * s01)
    row = LG[:, None] + 4 * np.arange(4)[None, :]
    col = np.broadcast_to(LR[:, None], row.shape)
    pd = np.array([W[16 * (i & 3) + i, i >> 2] for i in range(16)])   # pivots: lane 16 (i&3) + i, register i >> 2
    ok = bool(np.all(pd > 0))
    rs = 1.0 / np.sqrt(np.where(pd > 0, pd, 1.0))
    fin = np.where(col < row, W * rs[row], np.where(col == row, rs[row], 0.0))
    out = np.zeros(256)
    for r in range(4):
        for l in range(64):
            out[img_rc(LG[l] + 4 * r, LR[l])] = fin[l, r]
    return out, fin, ok


def run_model(n, ny, seed, NT=None, sz=20, res=0.15):
    sf, lsq, sn = 0.0025, 9.0, 0.0016
    cexp = -0.5 / lsq
    off, x0, x1, y = synth.make_patches(1, n, seed=seed, ny=ny)
    NT = NT or (n + 15) // 16
    npad = 16 * NT
    nt = (n + 15) // 16
    px0 = np.zeros(npad); px1 = np.zeros(npad)
    px0[:n], px1[:n] = x0, x1
    yc = np.zeros((ny, npad)); yc[:, :n] = y
    zv = np.zeros((ny, npad)); av = np.zeros((ny, npad))
    # Gram tiles (transposed storage)
    acc = {}
    for j in range(NT):
        for i in range(j, NT):
            if i >= nt:
                continue
            a = np.zeros((64, 4))
            pi = 16 * i + LR
            for r in range(4):
                pj = 16 * j + LG + 4 * r
                d0, d1 = px0[pi] - px0[pj], px1[pi] - px1[pj]
                v = sf * np.exp(cexp * (d0 * d0 + d1 * d1))
                v = np.where(pi == pj, v + sn + sn, v)
                v = np.where((pi >= n) | (pj >= n), (pi == pj).astype(float), v)
                a[:, r] = v
            acc[(i, j)] = a
    Linv = np.zeros((NT, 256))
    pan = np.zeros((NT, 256))
    LinvT = np.zeros((NT, 256))
    G = np.zeros((NT, 256))
    for k in range(-1, nt):
        if k >= 0:
            lv = Linv[k][IMG_LS]
            # z_k = L_kk^-1 y_k by 16 row-threads reading the operand image: row m = 4 chunks of 4 at (m + 16 g) * 4
            for c in range(ny):
                for mrow in range(16):
                    s_ = 0.0
                    for gq in range(4):
                        ch = Linv[k][IMG_LS[mrow + 16 * gq]]
                        for s in range(4):
                            s_ += ch[s] * yc[c, 16 * k + gq + 4 * s]
                    zv[c, 16 * k + mrow] = s_
            for i in range(k + 1, nt):
                D1 = np.zeros((64, 4)); D2 = np.zeros((64, 4))
                D1 = mfma(lv[:, 0], acc[(i, k)][:, 0], D1)
                D2 = mfma(lv[:, 2], acc[(i, k)][:, 2], D2)
                D1 = mfma(lv[:, 1], acc[(i, k)][:, 1], D1)
                D2 = mfma(lv[:, 3], acc[(i, k)][:, 3], D2)
                D = D1 + D2
                acc[(i, k)] = D
                pan[i][IMG_LS] = D
            if k + 1 < nt:
                # G_k: (L_(k+1)k L_kk^-1) in C/D layout = operand image of its transpose L_kk^-T L_(k+1)k^T
                Ln, ltk = pan[k + 1][IMG_LS], LinvT[k][IMG_LS]
                Dg = np.zeros((64, 4))
                for s in range(4):
                    Dg = mfma(Ln[:, s], ltk[:, s], Dg)
                G[k][IMG_LS] = Dg
            # y_i -= L_ik z_k by row-threads reading the LDS panel
            for i in range(k + 1, nt):
                for mrow in range(16):
                    for c in range(ny):
                        s_ = 0.0
                        for gq in range(4):
                            ch = pan[i][IMG_LS[mrow + 16 * gq]]
                            for s in range(4):
                                s_ += ch[s] * zv[c, 16 * k + gq + 4 * s]
                        yc[c, 16 * i + mrow] -= s_
        if k + 1 < nt:
            a_ = acc[(k + 1, k + 1)]
            if k >= 0:
                p = pan[k + 1][IMG_LS]
                for s in range(4):
                    a_ = mfma(p[:, s], p[:, s], a_, neg_a=True)
                acc[(k + 1, k + 1)] = a_
            # hand-over in register layout (the tile is symmetric, so the transposed storage does not matter)
            Linv[k + 1], fin, ok = diag_factor(a_)
            LinvT[k + 1][IMG_LS] = fin              # mf_img_store of the C/D registers = operand image of L^-T
            assert ok
        if k >= 0:
            for j in range(k + 1, nt):
                for i in range(j, nt):
                    if (i, j) == (k + 1, k + 1):
                        continue
                    a = pan[j][IMG_LS]
                    b = pan[i][IMG_LS]
                    t = acc[(i, j)]
                    for s in range(4):
                        t = mfma(a[:, s], b[:, s], t, neg_a=True)
                    acc[(i, j)] = t
    # backward solve over tile columns: alpha_k = L_kk^-T (z_k - w_k) - G_k alpha_(k+1), w_k = sum_{i>=k+2} L_ik^T alpha_i.
    # The sub-diagonal tile is folded into G_k = L_kk^-T L_(k+1)k^T (built by the factor wave during the factorisation,
    # stored as an operand image), so the recurrence stays inside the factor wave: alpha_(k+1) is consumed in the
    # register layout the previous iteration's MFMAs left it in.
    al = np.zeros((64, 4))
    for k in range(nt - 1, -1, -1):
        ub = np.zeros((64, 4))
        for c in range(ny):
            w = np.zeros(16)
            pa = np.zeros((64, 4))
            for i in range(k + 2, nt):
                pa += acc[(i, k)] * av[c, 16 * i + LR][:, None]
            for s in range(4):
                for g in range(4):
                    w[g + 4 * s] = pa[LG == g, s].sum()          # mf_row_reduce4 + ds_add_f64
            u = zv[c, 16 * k:16 * k + 16] - w
            for q4 in range(4):
                ub[LR == c, q4] = u[LG[LR == c] + 4 * q4]
        lt = LinvT[k][IMG_LS]
        D = np.zeros((64, 4))
        for s in range(4):
            D = mfma(lt[:, s], ub[:, s], D)
        if k + 1 < nt:
            gk = G[k][IMG_LS]
            for s in range(4):
                D = mfma(gk[:, s], al[:, s], D, neg_a=True)
        al = D
        assert np.all(al[LR >= ny] == 0.0)
        for c in range(ny):
            for r in range(4):
                av[c, 16 * k + LG[LR == c] + 4 * r] = al[LR == c, r]
    # separable predictive mean on the sz x sz grid, wave w takes points [32 w, 32 w + 32)
    f = np.zeros((ny, sz * sz))
    for c in range(ny):
        red = np.zeros((8, 4, 64, 4))
        for w in range(8):
            ibase = 32 * w
            if ibase >= n:
                continue
            P = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(2)]
            for s in range(8):
                i = ibase + 4 * s + LG
                al = sf * av[c, np.minimum(i, npad - 1)]
                ea, eb = [], []
                for h in range(2):
                    pq = 16 * h + LR
                    gq = res * ((pq + 0.5) / sz - 0.5)
                    on = (pq < sz) & (i < n)
                    ii = np.minimum(i, npad - 1)
                    ea.append(np.where(on, np.exp(cexp * (gq - px1[ii]) ** 2), 0.0))
                    eb.append(np.where(on, np.exp(cexp * (gq - px0[ii]) ** 2), 0.0))
                for nl in range(2):
                    bop = eb[nl] * al
                    for mt in range(2):
                        P[mt][nl] = mfma(ea[mt], bop, P[mt][nl])
            for mt in range(2):
                for nl in range(2):
                    red[w, mt * 2 + nl] = P[mt][nl]
        for oo in range(1024):
            tile, e = oo >> 8, oo & 255
            l2, r = e >> 2, e & 3
            py, pxx = 16 * (tile >> 1) + (l2 >> 4) + 4 * r, 16 * (tile & 1) + (l2 & 15)
            if py < sz and pxx < sz:
                f[c, py * sz + pxx] = red[:, tile, l2, r].sum()
    return (x0, x1, y), av[:, :n], f, acc, nt


def _reference(x0, x1, y, sz=20, res=0.15):
    X = np.stack([x0, x1], 1)
    Lc, alpha = R.dense_fit(X, y)
    xs0, xs1 = R.grid(res, sz)
    f, _ = R.dense_predict(X, Lc, alpha, np.stack([xs0, xs1], 1))
    return Lc, alpha, f


def test_register_tile_data_flow_matches_cholesky_solve():
    for n, ny, seed, NT in ((64, 1, 1, 4), (100, 3, 2, 8), (37, 1, 3, 4), (256, 1, 4, 16), (129, 1, 5, 12)):
        (x0, x1, y), alpha, f, acc, nt = run_model(n, ny, seed, NT=NT)
        Lc, want_alpha, want_f = _reference(x0, x1, y)
        assert np.max(np.abs(alpha - want_alpha)) <= 1e-9 * np.max(np.abs(want_alpha))
        assert np.max(np.abs(f - want_f)) <= 1e-10 * np.max(np.abs(want_f))
        # the finished panel tile (i, k) is the operand image of L_ik: register r of lane l = L[16 i + (l&15)][16 k + (l>>4) + 4 r]
        Lp = np.eye(16 * nt)
        Lp[:n, :n] = Lc
        for (i, k), t in acc.items():
            if i > k:
                for r in range(4):
                    assert np.allclose(t[:, r], Lp[16 * i + LR, 16 * k + LG + 4 * r], rtol=0, atol=1e-13)


def test_transposing_row_reduction():
    """mf_row_reduce4 (csrc/dense_mfma.hip): four components summed over the 16 lanes of a DPP row in five rounds --
    row_ror:8 (l <-> l^8), row_half_mirror (l <-> 7-l inside each half row), quad_perm [1,0,3,2] and [2,3,0,1]."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(16, 4))
    l = np.arange(16)
    hi8, hi4 = (l & 8) != 0, (l & 4) != 0
    ror8 = lambda v: v[(l + 8) % 16]
    half_mirror = lambda v: v[np.where(l < 8, 7 - l, 23 - l)]
    qp = lambda v, perm: v[(l & ~3) | np.array(perm)[l & 3]]
    k0 = np.where(hi8, x[:, 2], x[:, 0]); k1 = np.where(hi8, x[:, 3], x[:, 1])
    s0 = np.where(hi8, x[:, 0], x[:, 2]); s1 = np.where(hi8, x[:, 1], x[:, 3])
    k0 = k0 + ror8(s0); k1 = k1 + ror8(s1)
    k = np.where(hi4, k1, k0); sd = np.where(hi4, k0, k1)
    k = k + half_mirror(sd)
    k = k + qp(k, [1, 0, 3, 2])
    k = k + qp(k, [2, 3, 0, 1])
    tot = x.sum(axis=0)
    for lane in (0, 4, 8, 12):
        assert abs(k[lane] - tot[lane >> 2]) < 1e-12
    sel = 2 * ((l >> 3) & 1) + ((l >> 2) & 1)
    assert np.allclose(k, tot[sel])
