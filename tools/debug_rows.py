import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_compressor_amd import capi, synth
capi.load()
ctx = capi.Context(0)
P, n = 64, 96
off, x0, x1, y = synth.make_patches(P, n, seed=4)
p = capi.default_params_sparse(1, capacity=200)
ga = capi.Sparse(ctx, p, P, 1)
gb = capi.Sparse(ctx, p, P, 1)
coff = np.arange(P + 1, dtype=np.int32)
found = False
for t in range(n):
    idx = off[:-1] + t
    os.environ.pop("GPC_SPARSE_NO_ROWS", None)
    _, tra = ga.add(coff, x0[idx], x1[idx], y[:, idx], trace=True)
    os.environ["GPC_SPARSE_NO_ROWS"] = "1"
    _, trb = gb.add(coff, x0[idx], x1[idx], y[:, idx], trace=True)
    sa, sb = ga.state(), gb.state()
    ba, bb = ga.sizes(), gb.sizes()
    if not np.array_equal(ba, bb):
        print("point", t, "sizes differ", np.where(ba != bb)[0][:5], ba[ba != bb][:5], bb[ba != bb][:5]); found = True; break
    for name, A_, B_ in zip(("alpha", "C", "Q", "BV"), sa, sb):
        for i in range(P):
            nb = int(ba[i])
            a_ = A_[i][..., :nb] if name == "alpha" else (A_[i][:nb, :nb] if name in "CQ" else A_[i][:nb])
            b_ = B_[i][..., :nb] if name == "alpha" else (B_[i][:nb, :nb] if name in "CQ" else B_[i][:nb])
            if not np.array_equal(a_, b_, equal_nan=True):
                d = np.argwhere(a_ != b_)
                print("point", t, "patch", i, name, "differs; b =", nb, "first idx", d[:4].tolist(), a_[tuple(d[0])], b_[tuple(d[0])], "trace", tra[i], trb[i])
                for nm2, A2, B2 in zip(("alpha", "C", "Q", "BV"), sa, sb):
                    u = A2[i][..., :nb] if nm2 == "alpha" else (A2[i][:nb, :nb] if nm2 in "CQ" else A2[i][:nb])
                    v = B2[i][..., :nb] if nm2 == "alpha" else (B2[i][:nb, :nb] if nm2 in "CQ" else B2[i][:nb])
                    print("   ", nm2, "differing entries", int(np.sum(u != v)), "of", u.size, "max rel", float(np.max(np.abs(u - v) / (np.abs(v) + 1e-300))))
                np.set_printoptions(precision=17, linewidth=200)
                print("    alpha rows:", sa[0][i][0, :nb], "\n               ", sb[0][i][0, :nb])
                found = True
                break
        if found: break
    if found: break
print("identical through all points" if not found else "MISMATCH")
