#!/usr/bin/env python3
"""Where the add calls of the C4-defaults pass spend their time: the chain of ONE patch against the launch.

The pass is 4 add calls x 3 kernels (rows phase, small-basis kernel, regular kernel), each about 0.5 .. 0.9 ms at 32768 patches.  This
probe takes the batch of the bench record, reads the basis sizes after every call, picks patches by the phase they end a call in, and
times the same four add calls for (a) each such patch ALONE (P = 1: the latency of its chain of 64 point updates per call) and (b)
subsets of the batch made of one kind only.  If (a) is close to the kernel's time in the full launch, the launch is as long as its
longest chain and only a shorter chain helps; if not, it is throughput.

    python tools/r4_tail_probe.py      -> JSON on stdout"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gp_compressor_amd import capi, synth
    P, n, chunks, cap, RES = 32768, 256, 4, 200, 0.15
    cn = n // chunks
    ctx = capi.Context(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    ctx.set_stream(st.cuda_stream)
    off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=4)
    prm = capi.default_params_sparse(1, capacity=cap)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def run(ids, reps=5):
        ids = np.asarray(ids)
        Q = len(ids)
        g = capi.Sparse(ctx, prm, Q, 1)
        coff = t((np.arange(Q + 1) * cn).astype(np.int32))
        bufs = []
        for c in range(chunks):
            idx = (off[ids, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
            bufs.append((t(x0[idx]), t(x1[idx]), t(y[:, idx])))
        times = np.zeros((reps, chunks))
        sizes = []
        for r in range(reps + 1):
            g.reset()
            evs = []
            for c in range(chunks):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g.add_dev(coff, cn, Q * cn, *bufs[c])
                b.record()
                evs.append((a, b))
                if r == 0:
                    torch.cuda.synchronize()
                    sizes.append(g.sizes().copy())
            torch.cuda.synchronize()
            if r > 0:
                times[r - 1] = [a.elapsed_time(b) for a, b in evs]
        g.close()
        return np.median(times, axis=0), np.array(sizes)

    t_all, sz = run(np.arange(P))
    out = {"full_batch_ms_per_call": t_all.round(4).tolist(),
           "basis_after_call_percentiles_50_90_99_max": [[int(np.percentile(s, q)) for q in (50, 90, 99, 100)] for s in sz],
           "patches_above_16_24_after_call": [[int((s > 16).sum()), int((s > 24).sum())] for s in sz]}
    fin = sz[-1]
    kinds = {"rows only (final basis <= 14)": np.where(fin <= 14)[0], "small-basis kernel (final basis 18 .. 24)": np.where((fin >= 18) & (fin <= 24))[0],
             "regular kernel (final basis 28 .. 36)": np.where((fin >= 28) & (fin <= 36))[0], "regular kernel (final basis > 40)": np.where(fin > 40)[0]}
    out["kinds"] = {}
    for name, ids in kinds.items():
        rec = {"patches": int(len(ids))}
        if len(ids):
            alone = [run(ids[k:k + 1])[0] for k in range(min(3, len(ids)))]
            rec["one_patch_alone_ms_per_call"] = np.median(np.array(alone), axis=0).round(4).tolist()
            for Q in (256, 4096):
                if len(ids) >= Q:
                    rec[f"{Q}_of_this_kind_ms_per_call"] = run(ids[:Q])[0].round(4).tolist()
            rec["all_of_this_kind_ms_per_call"] = run(ids)[0].round(4).tolist()
        out["kinds"][name] = rec
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
