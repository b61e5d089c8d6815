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

    def run_gpu(off, x0, x1, y):
        keep = []

        def t(a):
            keep.append(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
            return keep[-1]
        g = capi.Sparse(ctx, prm, P, 1)
        cn = n // chunks
        coff = t((np.arange(P + 1) * cn).astype(np.int32))
        for c in range(chunks):
            idx = (off[:-1, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
            g.add_dev(coff, cn, P * cn, t(x0[idx]), t(x1[idx]), t(y[:, idx]))
        f = torch.empty((P, 1, M), dtype=torch.float64, device=dev)
        g.predict_dev(M, t(xs0), t(xs1), f)
        torch.cuda.synchronize()
        out_ = f.cpu().numpy()
        g.close()
        return out_
    t0 = time.time()
    out = SP.blowup_counts(run_gpu, op, P, n, seeds, RES, SZ, synth)
    out["what"] = "patches with max|f*| > 5 max|y| at the reference's default hyper-parameters, C4 shape, per implementation"
    out["seconds"] = time.time() - t0
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
