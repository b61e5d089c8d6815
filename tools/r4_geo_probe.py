#!/usr/bin/env python3
"""How many patches of the C4-defaults batch take a geometric deletion (decision byte bits 4-6), per add call, and how many leave each phase."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_compressor_amd import capi, synth
P, n, chunks, cap, RES = 8192, 256, 4, 200, 0.15
cn = n // chunks
ctx = capi.Context(0)
off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=4)
prm = capi.default_params_sparse(1, capacity=cap)
g = capi.Sparse(ctx, prm, P, 1)
out = []
coff = (np.arange(P + 1) * cn).astype(np.int32)
for c in range(chunks):
    idx = (off[:-1, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
    st, tr = g.add(coff, x0[idx], x1[idx], np.ascontiguousarray(y[:, idx]), trace=True)
    tr = tr.reshape(P, cn)
    geo = (tr & 0x70) != 0
    capd = (tr & 0x0e) != 0
    full = (tr & 1) != 0
    b = g.sizes()
    first_geo = np.where(geo.any(1), geo.argmax(1), cn)
    out.append({"call": c, "patches_with_geo_deletion": int(geo.any(1).sum()), "geo_deletions": int(geo.sum()), "capacity_deletions": int(capd.sum()),
                "full_updates": int(full.sum()), "mean_first_geo_point": float(first_geo[first_geo < cn].mean()) if (first_geo < cn).any() else None,
                "geo_patches_with_b_le16": int((geo.any(1) & (b <= 16)).sum()), "geo_patches_with_b_le24": int((geo.any(1) & (b <= 24)).sum()),
                "b_gt16": int((b > 16).sum()), "b_gt24": int((b > 24).sum())})
print(json.dumps(out, indent=1))
