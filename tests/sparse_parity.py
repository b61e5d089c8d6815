"""Parity statistics of the sparse online GP in ill-conditioned regimes (test infrastructure; imports the CPU oracle).

At the reference's default hyper-parameters (sigma_f^2 = 100, l^2 = 1 on a 0.15 m patch; /root/reference/src/rbf_kernel.h:24,
src/sparse_gp.h:48) the branch `gamma < eps_tol` (src/sparse_gp.hpp:144-163) is decided by rounding noise, so two correct fp64
implementations differ patch by patch and "GPU f* == oracle f*" is not a meaningful statement.  What IS
meaningful, and what this module computes for the GPU, the fp64 CPU oracle and the binary128 arbiter on the same patches:

  (a) "matched RMSE" (BASELINE.json north_star): reconstruction RMSE against the training targets -- predict_measurements on
      every patch's own points, the reference's training-set RMS block (src/gp_compressor.cpp:303-315, printed at :381);
  (b) per-patch max-norm error of f* on the decompression grid against the arbiter, as percentiles, for GPU and oracle;
  (c) the number of patches whose prediction leaves the data range (max|f*| > 5 max|y|) per implementation, with the worst one named.

`gate()` turns them into pass / fail: the GPU may not be worse than the fp64 oracle by more than the stated factors.  The factors
are frozen here (round 3); a kernel change that alters summation order ships with these statistics, not with a rewritten assert.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle_lib as O

BLOWUP = 5.0            # (c): max|f*| > BLOWUP * max|y| of the patch
# gate factors (GPU vs fp64 oracle on the same patches)
RMSE_TRAIN_FACTOR = 1.02      # (a) the GPU's reconstruction RMSE may exceed the oracle's by 2 %
PCT_FACTOR = 3.0              # (b) each percentile (50 / 90 / 99) of the GPU's error vs the arbiter <= 3 x the oracle's
MAX_FACTOR = 10.0             # (b) the worst patch: <= 10 x the oracle's worst (heavy tail: one patch decides it)
BLOWUP_FACTOR = 3.0           # (c) blow-ups, POOLED counts (>= 4 batches of 32768): <= 3 x the oracle's count + 2
BLOWUP_P_MIN = 1e-3           # (c) blow-ups, any sample: fail when P(X >= gpu | X ~ Binomial(gpu + oracle, share of the GPU's patches)) < 1e-3
# Round 4 -- the RATE behind (c), measured before the gate was re-frozen (tools/r4_blowup_rate.py, 8 seeded batches of 32768 x 256 at the
# reference's default hyper-parameters, C4 shape; gpurun_out/r4/blowup_rate.json -> profiles/r04_blowup_rate.json):
#   GPU     29 of 262144 = 3.6 per 32768   (per batch 4 3 7 4 0 4 2 5)
#   oracle  43 of 262144 = 5.4 per 32768   (per batch 1 8 4 6 4 9 6 5)
#   the binary128 arbiter on every one of those 72 patches: 0 blow-ups; no patch blows up in both fp64 implementations.
# The GPU does NOT blow up more often than the CPU restatement (one-sided p = 0.96 for "GPU rate > oracle rate"); round 3's "4 against 1"
# was one batch of Poisson noise.  A one-batch rule of the form gpu <= 3 oracle + 2 fails a correct build about once in a hundred
# batches at these rates, so the single-batch gate is the two-sample test above and the factor rule is applied to pooled counts only.


def _threads():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def run_cpu(op, off, x0, x1, y, xs0, xs1, idx, hp=False, fast=True, threads=None):
    """Oracle (or arbiter, hp=True) on the patches `idx` of a batch: f* (len(idx), ny, m), bv (len(idx),), f_train per patch
    (list of (ny, n_i)).  One C call per thread range."""
    idx = np.asarray(idx, dtype=np.int64)
    T = threads or _threads()
    T = max(1, min(T, len(idx)))
    # gather the selected patches into a compact batch
    cnt = (off[idx + 1] - off[idx]).astype(np.int64)
    sub = np.zeros(len(idx) + 1, dtype=np.int32)
    sub[1:] = np.cumsum(cnt)
    rows = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
    sx0, sx1, sy = np.ascontiguousarray(x0[rows]), np.ascontiguousarray(x1[rows]), np.ascontiguousarray(y[:, rows])
    bounds = [(t * len(idx)) // T for t in range(T + 1)]

    def one(t):
        lo, hi = bounds[t], bounds[t + 1]
        if hi <= lo:
            return None
        so = (sub[lo:hi + 1] - sub[lo]).astype(np.int32)
        sl = slice(int(sub[lo]), int(sub[hi]))
        return O.sparse_fit_predict_batch(op, so, sx0[sl], sx1[sl], np.ascontiguousarray(sy[:, sl]), xs0, xs1, fast=(fast and not hp),
                                          hp=hp, train=True)
    with ThreadPoolExecutor(T) as ex:
        outs = [o for o in ex.map(one, range(T)) if o is not None]
    f = np.concatenate([o[0] for o in outs], axis=0)
    bv = np.concatenate([o[2] for o in outs], axis=0)
    ft = np.concatenate([o[3] for o in outs], axis=1)
    return f, bv, ft, sub, sy


def _pct(e):
    return {"p50": float(np.percentile(e, 50)), "p90": float(np.percentile(e, 90)), "p99": float(np.percentile(e, 99)),
            "max": float(np.max(e))}


def stats(op, off, x0, x1, y, xs0, xs1, f_gpu, ft_gpu, sample, full_oracle=True, threads=None, oracle_idx=None):
    """f_gpu (P, ny, m): the GPU's grid prediction for EVERY patch; ft_gpu (ny, N): its prediction at the training points.
    sample: indices of the patches the arbiter runs on (>= 512 at the BASELINE size in the small-basis regime).  The fp64 oracle
    runs on oracle_idx (a superset of sample), or on all P patches (full_oracle: cheap in the small-basis regime), or on the sample."""
    P = len(off) - 1
    sample = np.asarray(sample, dtype=np.int64)
    ymax = np.array([np.max(np.abs(y[:, off[i]:off[i + 1]])) if off[i + 1] > off[i] else 0.0 for i in range(P)])
    f_hp, bv_hp, ft_hp, sub, sy = run_cpu(op, off, x0, x1, y, xs0, xs1, sample, hp=True, threads=threads)
    o_idx = np.asarray(oracle_idx, dtype=np.int64) if oracle_idx is not None else (np.arange(P) if full_oracle else sample)
    f_or_all, bv_or_all, ft_or_all, sub_o, _ = run_cpu(op, off, x0, x1, y, xs0, xs1, o_idx, threads=threads)
    pos = {int(i): k for k, i in enumerate(o_idx)}
    sel = np.array([pos[int(i)] for i in sample])
    f_or = f_or_all[sel]
    rows_s = np.concatenate([np.arange(off[i], off[i + 1]) for i in sample])
    ft_g = ft_gpu[:, rows_s]
    ft_o = np.concatenate([ft_or_all[:, sub_o[k]:sub_o[k + 1]] for k in sel], axis=1)

    def rm(a):
        return float(np.sqrt(np.mean((a - sy) ** 2)))
    # (b) per-patch max-norm error against the arbiter, relative to the patch's max|y|
    ys = np.maximum(ymax[sample], 1e-300)
    e_g = np.max(np.abs(f_gpu[sample] - f_hp), axis=(1, 2))
    e_o = np.max(np.abs(f_or - f_hp), axis=(1, 2))
    # (c) predictions that leave the data range
    r_g = np.max(np.abs(f_gpu), axis=(1, 2)) / np.maximum(ymax, 1e-300)
    r_o = np.max(np.abs(f_or_all), axis=(1, 2)) / np.maximum(ymax[o_idx], 1e-300)
    r_h = np.max(np.abs(f_hp), axis=(1, 2)) / ys
    # what the exact recursion does on the patches where an fp64 implementation blew up (rounding artefact or property of the data?)
    blow_g = np.where(r_g > BLOWUP)[0]
    blow_o = o_idx[np.where(r_o > BLOWUP)[0]]
    blow = np.array(sorted(set(blow_g.tolist()) | set(blow_o.tolist())), dtype=np.int64)
    if len(blow) > 32:
        # many (the basis-filling regime, where the exact recursion leaves the data range as well): the GPU's three worst and the
        # first few -- enough to say whether the others follow it there
        top = blow_g[np.argsort(r_g[blow_g])[::-1][:3]]
        blow = np.array(sorted(set(top.tolist()) | set(blow[:8].tolist())), dtype=np.int64)
    blow_list = []
    if len(blow):
        f_hb = run_cpu(op, off, x0, x1, y, xs0, xs1, blow, hp=True, threads=threads)[0]
        for k, i in enumerate(blow):
            if i in pos:
                ro_i = float(r_o[pos[int(i)]])
            else:                                      # outside the oracle's patch set: run it on this one
                ro_i = float(np.max(np.abs(run_cpu(op, off, x0, x1, y, xs0, xs1, [i], threads=1)[0])) / max(ymax[i], 1e-300))
            blow_list.append({"patch": int(i), "gpu": float(r_g[i]), "oracle": ro_i,
                              "arbiter": float(np.max(np.abs(f_hb[k])) / max(ymax[i], 1e-300))})
    wg = int(np.argmax(r_g))
    wo = int(o_idx[int(np.argmax(r_o))])
    d_go = f_gpu[o_idx] - f_or_all
    worst_diff = int(o_idx[int(np.argmax(np.max(np.abs(d_go), axis=(1, 2))))])
    out = {
        "sample_patches": int(len(sample)), "oracle_patches": int(len(o_idx)), "patches": int(P),
        "y_rms": float(np.sqrt(np.mean(sy ** 2))),
        "rmse_train": {"gpu": rm(ft_g), "oracle": rm(ft_o), "arbiter": rm(ft_hp),
                       "what": "sqrt(mean((predict_measurements(X_train) - y_train)^2)) over the sample "
                               "(the reference's training-set RMS block, src/gp_compressor.cpp:303-315)"},
        "err_vs_arbiter_abs": {"gpu": _pct(e_g), "oracle": _pct(e_o),
                               "what": "per-patch max|f* - f*_binary128| on the grid, percentiles over the sample"},
        "err_vs_arbiter_rel_ymax": {"gpu": _pct(e_g / ys), "oracle": _pct(e_o / ys)},
        "blowups": {"threshold": BLOWUP, "gpu": int(np.sum(r_g > BLOWUP)), "gpu_over": int(P),
                    "oracle": int(np.sum(r_o > BLOWUP)), "oracle_over": int(len(o_idx)),
                    "arbiter": int(np.sum(r_h > BLOWUP)), "arbiter_over": int(len(sample)),
                    "gpu_worst": {"patch": wg, "ratio": float(r_g[wg]), "max_abs_f": float(np.max(np.abs(f_gpu[wg]))),
                                  "oracle_ratio_same_patch": float(r_o[pos[wg]]) if wg in pos else None},
                    "oracle_worst": {"patch": wo, "ratio": float(np.max(r_o)), "gpu_ratio_same_patch": float(r_g[wo])},
                    "patches": blow_list,
                    "what": "patches with max|f*| > 5 max|y|; `patches`: the ratio max|f*| / max|y| of each implementation on "
                            "every patch where the GPU or the oracle exceeds the threshold"},
        "gpu_vs_oracle": {"rmse": float(np.sqrt(np.mean(d_go ** 2))), "max_abs": float(np.max(np.abs(d_go))),
                          "max_abs_patch": worst_diff,
                          "max_abs_patch_ratio_gpu": float(r_g[worst_diff]), "max_abs_patch_ratio_oracle": float(r_o[pos[worst_diff]]),
                          "f_rms": float(np.sqrt(np.mean(f_or_all ** 2)))},
        "bv_mean": {"oracle": float(np.mean(bv_or_all)), "arbiter": float(np.mean(bv_hp))},
    }
    out["gate"] = gate(out)
    return out


def blowup_p_value(g, g_over, o, o_over):
    """Two-sample Poisson test, conditional form: given g + o blow-ups in all, each falls on the GPU's side with probability
    g_over / (g_over + o_over) if the two rates are equal.  Returns P(X >= g) -- small means the GPU blows up more often."""
    import math
    n = int(g) + int(o)
    if n == 0:
        return 1.0
    q = g_over / float(g_over + o_over)
    if q >= 1.0:
        return 1.0
    # in log space (thousands of events in the basis-filling regime: the binomial coefficients overflow a double)
    lq, l1q = math.log(q), math.log1p(-q)
    logs = [math.lgamma(n + 1) - math.lgamma(k + 1) - math.lgamma(n - k + 1) + k * lq + (n - k) * l1q for k in range(int(g), n + 1)]
    top = max(logs)
    return float(min(1.0, math.exp(top) * sum(math.exp(v - top) for v in logs)))


def blowup_counts(run_gpu, op, P, n, seeds, res, sz, synth):
    """(c) over several seeded batches.  run_gpu(off, x0, x1, y) -> f* (P, ny, m) of the GPU; the fp64 oracle runs on every patch too.
    Returns per-seed counts, the pooled counts, the named patches (with the arbiter on each) and the pooled gate."""
    xs0, xs1 = synth.grid(res, sz)
    per_seed, named = [], []
    for seed in seeds:
        off, x0, x1, y = synth.make_patches(P, n, res=res, seed=seed)
        f_gpu = run_gpu(off, x0, x1, y)
        f_or = run_cpu(op, off, x0, x1, y, xs0, xs1, np.arange(P))[0]
        ymax = np.maximum(np.max(np.abs(y[0].reshape(P, n)), axis=1), 1e-300)
        r_g = np.max(np.abs(f_gpu), axis=(1, 2)) / ymax
        r_o = np.max(np.abs(f_or), axis=(1, 2)) / ymax
        bg, bo = np.where(r_g > BLOWUP)[0], np.where(r_o > BLOWUP)[0]
        both = np.array(sorted(set(bg.tolist()) | set(bo.tolist())), dtype=np.int64)
        if len(both):
            f_hp = run_cpu(op, off, x0, x1, y, xs0, xs1, both, hp=True)[0]
            for k, i in enumerate(both):
                named.append({"seed": int(seed), "patch": int(i), "gpu": float(r_g[i]), "oracle": float(r_o[i]),
                              "arbiter": float(np.max(np.abs(f_hp[k])) / ymax[i])})
        per_seed.append({"seed": int(seed), "gpu": int(len(bg)), "oracle": int(len(bo)),
                         "both": int(len(set(bg.tolist()) & set(bo.tolist()))),
                         "rmse_gpu_vs_oracle": float(np.sqrt(np.mean((f_gpu - f_or) ** 2)))})
    G, Oc, N = sum(s_["gpu"] for s_ in per_seed), sum(s_["oracle"] for s_ in per_seed), P * len(seeds)
    p_one = blowup_p_value(G, N, Oc, N)
    return {"patches_per_batch": P, "batches": len(seeds), "patches": N,
            "gpu": {"count": G, "per_32768": G * 32768.0 / N, "per_batch": [s_["gpu"] for s_ in per_seed]},
            "oracle": {"count": Oc, "per_32768": Oc * 32768.0 / N, "per_batch": [s_["oracle"] for s_ in per_seed]},
            "arbiter_blowups_on_named_patches": int(sum(1 for e in named if e["arbiter"] > BLOWUP)),
            "pooled_gate": {"rule": f"gpu <= {BLOWUP_FACTOR} x oracle + 2 on the pooled counts and one-sided p >= {BLOWUP_P_MIN}",
                            "ok": bool(G <= BLOWUP_FACTOR * Oc + 2 and p_one >= BLOWUP_P_MIN)},
            "two_sample": {"rule": "P(X >= gpu | X ~ Binomial(gpu + oracle, 1/2))", "p_one_sided": p_one},
            "per_seed": per_seed, "named": named}


def gate(s):
    """pass / fail with the reasons: the GPU against the fp64 oracle in the frozen factors above"""
    why = []
    a = s["rmse_train"]
    if not a["gpu"] <= RMSE_TRAIN_FACTOR * a["oracle"]:
        why.append(f"rmse_train gpu {a['gpu']:.4g} > {RMSE_TRAIN_FACTOR} x oracle {a['oracle']:.4g}")
    eg, eo = s["err_vs_arbiter_abs"]["gpu"], s["err_vs_arbiter_abs"]["oracle"]
    S = s["sample_patches"]
    for k in ("p50", "p90", "p99"):
        if (k == "p90" and S < 32) or (k == "p99" and S < 256):
            continue                                  # too few patches for that percentile to mean anything
        if not eg[k] <= PCT_FACTOR * eo[k]:
            why.append(f"err_vs_arbiter {k} gpu {eg[k]:.3g} > {PCT_FACTOR} x oracle {eo[k]:.3g}")
    if not eg["max"] <= MAX_FACTOR * eo["max"]:
        why.append(f"err_vs_arbiter max gpu {eg['max']:.3g} > {MAX_FACTOR} x oracle {eo['max']:.3g}")
    b = s["blowups"]
    p_one = blowup_p_value(b["gpu"], b["gpu_over"], b["oracle"], b["oracle_over"])
    b["p_gpu_rate_not_above_oracle"] = p_one
    if p_one < BLOWUP_P_MIN:
        why.append(f"blow-ups gpu {b['gpu']} of {b['gpu_over']} vs oracle {b['oracle']} of {b['oracle_over']}: one-sided p = {p_one:.2e} < {BLOWUP_P_MIN}")
    return {"ok": not why, "why": why,
            "factors": {"rmse_train": RMSE_TRAIN_FACTOR, "percentiles": PCT_FACTOR, "max": MAX_FACTOR,
                        "blowups": f"two-sample one-sided p >= {BLOWUP_P_MIN} (pooled batches: also gpu <= {BLOWUP_FACTOR} x oracle + 2)"}}
