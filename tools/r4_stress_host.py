#!/usr/bin/env python3
"""Randomised sweep of the host-pointer dense entry (round 4: eight chunks on two compute streams, workspace halves, fork only when every
chunk is the one-wave kernel's).  Random batches of >= 8192 patches -- uniform, ragged, with whole chunks of small patches, with and
without the variance, grid and point-wise X* -- through the default pipeline, the one-stream pipeline (GPC_HOST_ONE_STREAM=1) and no
pipeline at all (GPC_HOST_NO_PIPELINE=1): status words equal, values within 1e-9 of each other relative to the batch's largest value
(chunks may meet different kernels of the same algorithm: the size classes are per launch).

    python tools/r4_stress_host.py [n_configs] [seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from gp_compressor_amd import capi, synth
    ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = capi.Context(0)
    res, sz = 0.15, 8
    xs0, xs1 = synth.grid(res, sz)
    bad = []
    for k in range(ncfg):
        P = int(rng.choice([8192, 8200, 9001, 12288]))
        nmax = int(rng.choice([170, 200, 256, 257, 300, 400]))
        var = bool(rng.integers(2)) and nmax <= 256
        grid = bool(rng.integers(2)) and not var
        counts = rng.integers(max(1, nmax // 2), nmax + 1, size=P) if rng.integers(2) else np.full(P, nmax)
        for _ in range(int(rng.integers(0, 3))):            # whole chunks of small patches
            c = int(rng.integers(0, 8))
            lo, hi = P * c // 8, P * (c + 1) // 8
            counts[lo:hi] = rng.integers(20, min(nmax, int(rng.choice([100, 150, 192]))) + 1, size=hi - lo)      # (never above nmax: `keep` below cuts the data at nmax points per patch)
        counts[int(rng.integers(P))] = nmax
        off_f, x0_f, x1_f, y_f = synth.make_patches(P, nmax, res=res, seed=int(rng.integers(1 << 30)))
        keep = (np.arange(nmax)[None, :] < counts[:, None]).reshape(-1)
        off = np.zeros(P + 1, dtype=np.int32)
        off[1:] = np.cumsum(counts)
        x0, x1, y = np.ascontiguousarray(x0_f[keep]), np.ascontiguousarray(x1_f[keep]), np.ascontiguousarray(y_f[:, keep])
        prm = capi.default_params_dense(want_variance=1 if var else 0)
        outs = []
        for env in (None, "GPC_HOST_ONE_STREAM", "GPC_HOST_NO_PIPELINE"):
            if env:
                os.environ[env] = "1"
            if grid:
                f, st = ctx.dense_fit_predict_grid(prm, off, x0, x1, y, res, sz)
                v = None
            else:
                f, v, st = ctx.dense_fit_predict(prm, off, x0, x1, y, xs0, xs1)
            outs.append((f, v, st))
            if env:
                del os.environ[env]
        f0, v0, s0 = outs[0]
        ok = True
        for f, v, st in outs[1:]:
            ok = ok and np.array_equal(st, s0) and np.max(np.abs(f - f0)) <= 1e-9 * np.max(np.abs(f0))
            if var:
                ok = ok and np.max(np.abs(v - v0)) <= 1e-9 * np.max(np.abs(v0))
        ok = ok and bool(np.all(s0 == 0))
        if not ok:
            det = []
            for (f, v, st), nm in zip(outs[1:], ("one_stream", "no_pipeline")):
                w = np.where(st != s0)[0]
                det.append({"vs": nm, "status_diff": w[:5].tolist(), "st": st[w[:5]].tolist(), "st0": s0[w[:5]].tolist(),
                            "f_rel": float(np.max(np.abs(f - f0)) / np.max(np.abs(f0))),
                            "v_rel": float(np.max(np.abs(v - v0)) / np.max(np.abs(v0))) if var else None,
                            "v_argmax_patch": int(np.argmax(np.max(np.abs(v - v0), axis=1))) if var else None,
                            "n_of_that_patch": int(counts[int(np.argmax(np.max(np.abs(v - v0), axis=1)))]) if var else None})
            bad.append({"config": k, "P": P, "nmax": nmax, "var": var, "grid": grid, "nonzero_status": int((s0 != 0).sum()), "detail": det})
            print("MISMATCH", bad[-1], file=sys.stderr, flush=True)
    print(json.dumps({"configs": ncfg, "mismatches": bad}))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
