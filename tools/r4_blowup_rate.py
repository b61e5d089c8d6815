#!/usr/bin/env python3
"""Blow-up RATE of the sparse online GP at the reference's default hyper-parameters.

A "blow-up" is a patch whose prediction leaves the data range, max|f*| > 5 max|y| (tests/sparse_parity.py).  The mechanism is a full
update taken at gamma ~ eps_tol (/root/reference/src/sparse_gp.hpp:144-163): 1/gamma ~ 1e6 enters Q and the cancellation in alpha^T k
that follows loses the digits.  One batch of 32768 patches shows 1 .. 6 of them per implementation -- Poisson noise decides whether a
one-batch gate passes.  This script measures the rate on SEEDS batches (C4 shape: P x 256 points in 4 add calls, capacity 200) for the
GPU and the fp64 CPU oracle on the same patches, names every blow-up patch with what the other implementation and the binary128
arbiter give on it, and prints the pooled counts with the two-sample test the frozen gate uses from round 4 on.

    python tools/r4_blowup_rate.py [P] [seed seed ...]      -> one JSON object on stdout"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from gp_compressor_amd import capi, synth
    import oracle_lib as O
    import sparse_parity as SP
    O.build()
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    seeds = [int(s) for s in sys.argv[2:]] or [4, 11, 12, 13, 14, 15, 16, 17]
    n, chunks, cap, RES, SZ = 256, 4, 200, 0.15, 20
    M = SZ * SZ
    ctx = capi.Context(0)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    xs0, xs1 = synth.grid(RES, SZ)
    prm = capi.default_params_sparse(1, capacity=cap)
    op = O.sparse_params(1, p0=prm.sigmaf_sq, p1=prm.l_sq, s20=prm.noise, eps_tol=prm.eps_tol, capacity=cap)
    per_seed, named = [], []
    for seed in seeds:
        keep = []

        def t(a):
            keep.append(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
            return keep[-1]
        off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=seed)
        g = capi.Sparse(ctx, prm, P, 1)
        cn = n // chunks
        coff = t((np.arange(P + 1) * cn).astype(np.int32))
        for c in range(chunks):
            idx = (off[:-1, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
            g.add_dev(coff, cn, P * cn, t(x0[idx]), t(x1[idx]), t(y[:, idx]))
        f = torch.empty((P, 1, M), dtype=torch.float64, device=dev)
        g.predict_dev(M, t(xs0), t(xs1), f)
        torch.cuda.synchronize()
        f_gpu, bv = f.cpu().numpy(), g.sizes()
        g.close()
        del keep
        t0 = time.time()
        f_or = SP.run_cpu(op, off, x0, x1, y, xs0, xs1, np.arange(P))[0]
        dt = time.time() - t0
        ymax = np.maximum(np.max(np.abs(y[0].reshape(P, n)), axis=1), 1e-300)
        r_g = np.max(np.abs(f_gpu), axis=(1, 2)) / ymax
        r_o = np.max(np.abs(f_or), axis=(1, 2)) / ymax
        bg, bo = np.where(r_g > SP.BLOWUP)[0], np.where(r_o > SP.BLOWUP)[0]
        both = np.array(sorted(set(bg.tolist()) | set(bo.tolist())), dtype=np.int64)
        if len(both):
            f_hp = SP.run_cpu(op, off, x0, x1, y, xs0, xs1, both, hp=True)[0]
            for k, i in enumerate(both):
                named.append({"seed": seed, "patch": int(i), "gpu": float(r_g[i]), "oracle": float(r_o[i]),
                              "arbiter": float(np.max(np.abs(f_hp[k])) / ymax[i])})
        per_seed.append({"seed": seed, "gpu": int(len(bg)), "oracle": int(len(bo)), "both": int(len(set(bg.tolist()) & set(bo.tolist()))),
                         "bv_mean_gpu": float(bv.mean()), "bv_max_gpu": int(bv.max()), "oracle_seconds": dt,
                         "rmse_gpu_vs_oracle": float(np.sqrt(np.mean((f_gpu - f_or) ** 2)))})
        print(f"[blowup] seed {seed}: gpu {len(bg)} oracle {len(bo)} of {P}  (oracle {dt:.1f} s)", file=sys.stderr, flush=True)
    G = sum(s["gpu"] for s in per_seed)
    Oc = sum(s["oracle"] for s in per_seed)
    N = P * len(seeds)
    # conditional (binomial) two-sample Poisson test: given G + O events, is G larger than a fair split explains?  one-sided p-value
    tot = G + Oc
    p_one = sum(math.comb(tot, k) for k in range(G, tot + 1)) / 2.0 ** tot if tot else 1.0
    out = {"what": "patches with max|f*| > 5 max|y| at the reference's default hyper-parameters, C4 shape, per implementation",
           "patches_per_batch": P, "batches": len(seeds), "patches": N,
           "gpu": {"count": G, "per_32768": G * 32768.0 / N, "per_batch": [s["gpu"] for s in per_seed]},
           "oracle": {"count": Oc, "per_32768": Oc * 32768.0 / N, "per_batch": [s["oracle"] for s in per_seed]},
           "arbiter_blowups_on_named_patches": int(sum(1 for e in named if e["arbiter"] > SP.BLOWUP)),
           "pooled_gate": {"rule": "gpu <= 3 x oracle + 2 on the pooled counts", "ok": bool(G <= SP.BLOWUP_FACTOR * Oc + 2)},
           "two_sample": {"rule": "P(X >= gpu | X ~ Binomial(gpu + oracle, 1/2))", "p_one_sided": p_one},
           "per_seed": per_seed, "named": named}
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
