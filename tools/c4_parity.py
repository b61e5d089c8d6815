#!/usr/bin/env python3
"""C4 (sparse online GP, 32768 patches x 256 points in 4 add calls, capacity 200) parity statistics at the BASELINE size: the GPU
against the fp64 CPU oracle and the binary128 arbiter (tests/sparse_parity.py).  Prints one JSON object per regime.
    python tools/c4_parity.py [defaults|fill] [P] [arbiter_sample]"""
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
    regime = sys.argv[1] if len(sys.argv) > 1 else "defaults"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    S = int(sys.argv[3]) if len(sys.argv) > 3 else (1024 if regime == "defaults" else 32)
    n, chunks, cap, RES, SZ = 256, 4, 200, 0.15, 20
    ny = int(os.environ.get("GPC_C4_NY", "1"))
    ctx = capi.Context(0)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # kernels in order with torch's copies
    keep = []                                                  # device buffers stay alive until the final synchronize

    def t(a):
        keep.append(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        return keep[-1]
    off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=4, ny=ny)
    kw = dict(sigmaf_sq=1.0, l_sq=(RES / 8) ** 2, noise=1e-4, capacity=cap) if regime == "fill" else dict(capacity=cap)
    prm = capi.default_params_sparse(ny, **kw)
    g = capi.Sparse(ctx, prm, P, ny)
    xs0, xs1 = synth.grid(RES, SZ)
    M = SZ * SZ
    cn = n // chunks
    coff = t((np.arange(P + 1) * cn).astype(np.int32))
    for c in range(chunks):
        idx = (off[:-1, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
        g.add_dev(coff, cn, P * cn, t(x0[idx]), t(x1[idx]), t(y[:, idx]))
    f = torch.empty((P, ny, M), dtype=torch.float64, device=dev)
    g.predict_dev(M, t(xs0), t(xs1), f)
    ft = torch.empty((ny, P * n), dtype=torch.float64, device=dev)
    g.predict_points_dev(t(off), P * n, t(x0), t(x1), ft)
    torch.cuda.synchronize()
    f_gpu, ft_gpu, bv = f.cpu().numpy(), ft.cpu().numpy(), g.sizes()
    op = O.sparse_params(ny, p0=prm.sigmaf_sq, p1=prm.l_sq, s20=prm.noise, eps_tol=prm.eps_tol, capacity=cap)
    t0 = time.time()
    full = regime == "defaults" or os.environ.get("GPC_C4_FULL_ORACLE") == "1"
    st = SP.stats(op, off, x0, x1, y, xs0, xs1, f_gpu, ft_gpu, np.arange(S), full_oracle=full)
    st["regime"], st["cpu_seconds"], st["bv_mean"]["gpu"], st["bv_max_gpu"], st["ny"] = regime, time.time() - t0, float(bv.mean()), int(bv.max()), ny
    print(json.dumps(st, indent=1))
    g.close()
    ctx.close()


if __name__ == "__main__":
    main()
