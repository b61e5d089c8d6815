#!/usr/bin/env python3
"""Randomised sweep of the sparse add's phases: tests/sparse_sweep.py on as many configurations as asked for.

    python tools/r4_stress_sparse.py [n_configs] [seed]            -> JSON summary on stdout, exit code 1 on any mismatch
    python tools/r4_stress_sparse.py  n_configs   seed  K          -> configuration K of that sequence under every phase switch: which one
                                                                      makes the default path agree with the regular kernel alone"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GPC_SPARSE_FULL"] = "1"


def main():
    from gp_compressor_amd import capi, synth
    import sparse_sweep as SW
    ncfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None
    ctx = capi.Context(0)
    if only is None:
        def progress(k, bad):
            if k % 25 == 24:
                print(f"[{k + 1}/{ncfg}] mismatches so far: {len(bad)}", file=sys.stderr, flush=True)
        bad, hist = SW.sweep(capi, synth, ctx, ncfg, seed, progress)
        print(json.dumps({"configs": ncfg, "seed": seed, "mismatches": bad, "final_basis_sizes_seen": hist}, indent=1))
        ctx.close()
        sys.exit(1 if bad else 0)
    rng = np.random.default_rng(seed)
    for k in range(only + 1):
        c = SW.draw(rng, synth)
    base = SW.run(capi, ctx, c, "GPC_SPARSE_NO_SMALL")
    report = {}
    for env in (None, "GPC_SPARSE_NO_ROWS2", "GPC_SPARSE_NO_MID", "GPC_SPARSE_NO_ROWS", "GPC_SPARSE_NO_LIST", "GPC_SPARSE_ROWS_PERSISTENT",
                "GPC_SPARSE_NO_ROWS2+GPC_SPARSE_NO_MID"):
        cur = SW.run(capi, ctx, c, env, predict=synth.grid(0.15, 12))
        report[env or "default"] = ("OK" if SW.same(cur, base, c["P"]) else "DIFF") + ("" if cur[3] else " + predict kernels disagree")
    print(json.dumps({"config": only, "ny": c["ny"], "cap": c["cap"], "kernel": c["kernel"], "P": c["P"], "n": c["n"], "report": report}, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
