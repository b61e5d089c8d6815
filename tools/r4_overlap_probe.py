#!/usr/bin/env python3
"""Probe (round 4): do the phases of the sparse add path overlap when two add calls run on two streams?  C4 at the reference defaults:
one object of 32768 patches on one stream against two objects of 16384 patches on two contexts / streams issued interleaved.  If the
pair finishes clearly sooner, a tail (small-basis + regular kernel) can hide under the rows phase of the next call."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gp_compressor_amd import capi, synth  # noqa: E402

P, n, chunks, cap, res = 32768, 256, 4, 200, 0.15
dev = torch.device("cuda:0")
off, x0, x1, y = synth.make_patches(P, n, res=res, seed=4)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cn = n // chunks


def bufs_for(lo, hi):
    Pp = hi - lo
    coff = t((np.arange(Pp + 1) * cn).astype(np.int32))
    out = []
    for c in range(chunks):
        idx = (off[lo:hi, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
        out.append((t(x0[idx]), t(x1[idx]), t(y[:, idx])))
    return Pp, coff, out


def run(parts):
    """parts: list of (lo, hi); one context + stream + object per part, add calls issued round-robin"""
    objs = []
    for lo, hi in parts:
        ctx = capi.Context(0)
        s = torch.cuda.Stream()
        ctx.set_stream(s.cuda_stream)
        Pp, coff, bufs = bufs_for(lo, hi)
        g = capi.Sparse(ctx, capi.default_params_sparse(1, capacity=cap), Pp, 1)
        objs.append((ctx, s, g, Pp, coff, bufs))
    best = 1e9
    for rep in range(4):
        for o in objs:
            o[2].reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(chunks):
            for ctx, s, g, Pp, coff, bufs in objs:
                g.add_dev(coff, cn, Pp * cn, *bufs[c])
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    sizes = np.concatenate([o[2].sizes() for o in objs])
    for o in objs:
        o[2].close()
        o[0].close()
    return best, float(sizes.mean())


one, b1 = run([(0, P)])
two, b2 = run([(0, P // 2), (P // 2, P)])
four, b4 = run([(i * P // 4, (i + 1) * P // 4) for i in range(4)])
print(json.dumps({"one_object_ms": 1e3 * one, "two_objects_two_streams_ms": 1e3 * two, "four_objects_four_streams_ms": 1e3 * four,
                  "bv_mean": [b1, b2, b4]}))
