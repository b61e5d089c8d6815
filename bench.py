#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-patch GP hot path on MI355X (contract: see the task's bench section).

Headline workload (BASELINE.json configs[1], "C2"): a 1M-point synthetic room scan = 8192 octree-leaf patches x 256
points per GPU, RBF kernel + Gaussian noise, batched Cholesky fit and predictive mean on the 20 x 20 decompression
grid (m = 400) -- i.e. gp_compressor::train_processes + the patch loop of load_compressed
(/root/reference/src/gp_compressor.cpp:121-175, 298-380) with the dense gaussian_process model
(/root/reference/src/gaussian_process.cpp:15-45) on every patch.

A "step" is one pass of the hot path over the whole batch through the C-ABI with the patch buffers resident in HBM.
  N = 1: one launch of gpc_dense_fit_predict_grid_dev.
  N > 1: the north-star path -- the job's batch (N x 8192 patches, rank-seeded shards) is partitioned over the ranks by
         gpc_partition_patches (longest-processing-time, fixed-size padded slots), every rank fits + predicts its slots,
         ONE RCCL all-gather reassembles f_star, a device gather un-permutes it to patch order; the result is checked
         against the rank's own slots.  Weak scaling (8192 patches per GPU); the gather of step k overlaps the kernel of k+1.

Besides the headline the same JSON line carries "secondary": driver-timed records of the other BASELINE configurations,
each with its own config.workload, roofline and (N = 1) cpu_baseline:
  N = 1: C3-per-GPU (8192 x 512 dense), C4 (sparse online, 32768 x 256 in 4 chunks, capacity 200: the basis-filling
         kernel and the reference's default hyper-parameters), C5 (probit IRLS, 4096 x 1024).
  N > 1: C3 (N x 8192 patches x 512 points -- 65536 at N = 8, BASELINE configs[2]) through the same sharded path.

    python bench.py                       # N = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix == vector peak, AMD public spec (256 CUs x 4 SIMD x 32 FLOP/clk x 2.4 GHz)
HBM_PEAK_GBPS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
RES, SZ = 0.15, 20
M = SZ * SZ


def _log(msg):
    """progress on stderr (GPC_BENCH_VERBOSE=1): the JSON line on stdout stays the only output there"""
    if os.environ.get("GPC_BENCH_VERBOSE") == "1":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def algorithmic_flops(n, m):
    """dense fit + predictive mean, kernel evaluation = 7 flops, symmetric K counted once."""
    return 3.5 * n * n + n ** 3 / 3.0 + 2.0 * n * n + 9.0 * n * m


def irls_flops(n, m, iters):
    """config 5: every Newton step is one Gram build + factorisation + two triangular solves; one predictive mean at the end"""
    return iters * (3.5 * n * n + n ** 3 / 3.0 + 2.0 * n * n) + 9.0 * n * m


def host_cores():
    """Threads to use for the CPU baseline: the cgroup CPU quota if there is one (a 1-GPU box gets a 16-CPU share of
    the host), else the affinity mask, capped at 64."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("GPC_CPU_THREADS", "16"))))


def _traffic(key, field="hbm_bytes_per_launch"):
    """HBM bytes per launch from the PMC passes (profiles/traffic.json, collected and corrected as the microarch guide
    prescribes; tools/collect_profiles_r04.py) or None.  Other fields of the same record: the rocprofv3 kernel time of the
    timed launches, the counted wave-instructions of a pass."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {}).get(field)
    except Exception:
        return None


def _kstats(ms):
    """median / mean / min / max of per-step kernel times (HIP events on the launch stream)"""
    a = np.asarray(ms, dtype=np.float64)
    if a.size == 0:
        return {"median": float("nan"), "mean": float("nan"), "min": float("nan"), "max": float("nan"), "steps": 0}
    return {"median": float(np.median(a)), "mean": float(np.mean(a)), "min": float(np.min(a)), "max": float(np.max(a)), "steps": int(a.size)}


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.build()
    return O


def _timed_threads(fn, n_items, cores, budget_s, probe=2):
    """Runs fn(lo, hi) over a bounded sample: one probe on a single thread (the per-item cost, which is also how the
    reference itself runs -- it has no threads), then one chunk per thread sized for ~budget_s.  Returns
    (outputs, items_done, wall_s, single_thread_items_per_s)."""
    from concurrent.futures import ThreadPoolExecutor
    fn(0, 1)                                   # warm-up (library load, page faults)
    probe = max(1, min(probe, n_items))
    t0 = time.perf_counter()
    fn(0, probe)
    per_item = (time.perf_counter() - t0) / probe
    per_thread = max(1, min(n_items // cores, int(budget_s / max(per_item, 1e-9))))
    chunks = [(i * per_thread, (i + 1) * per_thread) for i in range(cores) if (i + 1) * per_thread <= n_items]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(len(chunks)) as ex:
        outs = list(ex.map(lambda c: fn(*c), chunks))
    dt = time.perf_counter() - t0
    return outs, per_thread * len(chunks), dt, 1.0 / per_item


def cpu_baseline_dense(off, x0, x1, y, f_gpu, budget_s):
    """Times the CPU oracle (oracle/gpc_oracle.c, -O3 -march=native build; kind "port": the reference's own Eigen code
    cannot be built here) on a bounded sample of the same workload, one thread per host core, and returns
    the baseline record plus the RMSE of the GPU result against it on that sample."""
    O = _oracle()
    p = O.dense_params()
    xs0, xs1 = O.grid(RES, SZ)
    P = len(off) - 1
    n = int(off[1] - off[0])

    def run(lo, hi):
        sub = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        sl = slice(int(off[lo]), int(off[hi]))
        return O.dense_fit_predict_batch(p, sub, np.ascontiguousarray(x0[sl]), np.ascontiguousarray(x1[sl]),
                                         np.ascontiguousarray(y[:, sl]), xs0, xs1, fast=True)[0]
    cores = host_cores()
    outs, done, dt, single = _timed_threads(run, P, cores, budget_s, probe=min(64, max(2, 4096 // n)))
    f_cpu = np.concatenate(outs, axis=0)
    diff = f_gpu[:done] - f_cpu
    rec = {"value": done / dt, "unit": "patches/s", "cores": cores, "kind": "port", "single_thread_value": single,
           "sample": f"{done} of the {P} patches x {n} pts (same buffers), oracle/gpc_oracle.c -O3 -march=native, "
                     f"{cores} threads, {dt:.1f} s"}
    rm = {"rmse": float(np.sqrt(np.mean(diff * diff))), "max_abs": float(np.max(np.abs(diff))),
          "f_rms": float(np.sqrt(np.mean(f_cpu * f_cpu))), "what": "GPU f* vs CPU oracle f* on the cpu_baseline sample"}
    return rec, rm


# ------------------------------------------------------------------------------------------------ dense (C2 headline, C3)

def bench_dense(env, P, n, steps, warmup, seed=2):
    """One GPU's (or, N > 1, the sharded job's) dense fit + predict.  Returns a dict of measurements; `P` is patches PER GPU."""
    import torch
    import torch.distributed as dist
    from gp_compressor_amd import capi, dist as gdist, synth
    ctx, dev, world, rank, use_dist = env["ctx"], env["dev"], env["world"], env["rank"], env["use_dist"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    prm = capi.default_params_dense()                         # gaussian_process defaults, reference double-noise (F5)
    if not use_dist:
        off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=seed)
        slots = None
        S, Pg = P, P
    else:
        # the whole job's batch from rank-seeded shards, then the LPT partition of the C-ABI: this rank's slots
        goff, gx0, gx1, gy = gdist.global_batch(world, P, n, res=RES, seed=seed)
        Pg = len(goff) - 1
        slots, off, x0, x1, y = gdist.shard_batch(goff, gx0, gx1, gy, world, rank)
        S = slots.shape[1]
        del gx0, gx1, gy
    N = int(off[-1])
    n_max = int(np.max(np.diff(off))) if S else 0
    d_off, d_x0, d_x1, d_y = t(off), t(x0), t(x1), t(y)
    nb = 2 if use_dist else 1
    f_bufs = [torch.empty((S, 1, M), dtype=torch.float64, device=dev) for _ in range(nb)]
    status = torch.empty((S,), dtype=torch.int32, device=dev)
    gathers, exchange = None, None
    if use_dist:
        # the exchange goes through the C-ABI's own communicator (gpc_comm_create + gpc_allgather_fstar_dev) -- what a reference-side
        # binding would call; torch.distributed only if RCCL cannot be bound (or GPC_BENCH_TORCH_GATHER=1, for comparison)
        g0, exchange = gdist.make_gather(slots, Pg, f_bufs[0], world, rank, dev.index or 0,
                                         prefer_cabi=os.environ.get("GPC_BENCH_TORCH_GATHER") != "1")
        g1 = (gdist.CabiGather(slots, Pg, f_bufs[0], world, rank, dev.index or 0, share=g0) if isinstance(g0, gdist.CabiGather)
              else gdist.ShardedGather(slots, Pg, f_bufs[0], world))
        gathers = [g0, g1]
    pending = [None]

    def step(k, ev=None):
        b = k & 1 if use_dist else 0
        if ev is not None:
            ev[0].record()
        ctx.dense_fit_predict_grid_dev(prm, S, d_off, n_max, N, d_x0, d_x1, d_y, 1, RES, SZ, f_bufs[b], status=status)
        if ev is not None:
            ev[1].record()
        if use_dist:
            gathers[b].start(f_bufs[b], async_op=True)        # RCCL's stream; overlaps the un-permute below and the next kernel
            if pending[0] is not None:
                gathers[pending[0]].finish()                  # step k-1: wait for its gather, un-permute to patch order
            pending[0] = b

    def fence():
        if use_dist and pending[0] is not None:
            gathers[pending[0]].finish()
            pending[0] = None
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(warmup):
        step(k)
    fence()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        step(k, events[k])
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kt = _kstats([a.elapsed_time(b) for a, b in events])
    kern_ms = kt["median"]                       # the roofline uses the MEDIAN of the per-step HIP-event times; mean / min / max beside it
    last = (steps - 1) & 1 if use_dist and steps > 0 else 0
    st = status.cpu().numpy()
    f_host = f_bufs[last].cpu().numpy()
    ok = bool(np.all(st == 0)) and bool(np.all(np.isfinite(f_host)))
    if use_dist:
        ok = ok and gathers[last].own_rows_match(f_bufs[last], rank, slots)
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    kernel = ctx.last_dense_kernel()
    flops = algorithmic_flops(n, M) * P
    achieved = flops / (kern_ms * 1e-3) / 1e12
    if gathers:
        torch.cuda.synchronize()
        for g_ in gathers[::-1]:
            if hasattr(g_, "close"):
                g_.close()
    del d_off, d_x0, d_x1, d_y, f_bufs, gathers
    return {"value": world * P * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "kern_ms": kern_ms, "kern_stats": kt, "ok": ok,
            "kernel": kernel, "achieved": achieved, "exchange": exchange, "host": (off, x0, x1, y, f_host) if not use_dist else None}


def dense_record(name, r, P, n, world, steps, warmup):
    tkey = f"dense_mfma_big@n{n}" if ("dense_mfma_big" in r["kernel"] and n > 272) else r["kernel"]
    par = (f"{world * P} patches partitioned over {world} ranks by gpc_partition_patches (LPT), 1 all-gather of f_star + un-permute"
           if world > 1 else "1 GPU")
    return {"metric": "patches/sec (compress+predict)", "value": r["value"], "unit": "patches/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}: {P} patches x {n} pts per GPU, RBF + Gaussian noise, dense Cholesky fit + predictive mean "
                                   f"on the {SZ}x{SZ} grid (m={M})",
                       "patches_per_gpu": P, "points_per_patch": n, "grid_points": M, "parallelism": par, "kernel": r["kernel"],
                       "results_ok": r["ok"], **({"exchange": r["exchange"]} if r.get("exchange") else {})},
            "roofline": {"bound": "mfma", "achieved": r["achieved"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": r["achieved"] / FP64_PEAK_TFLOPS, "traffic": _traffic(tkey), "kernel_ms": r["kern_ms"],
                         "kernel_ms_stats": r["kern_stats"], "kernel_ms_rocprof": _traffic(tkey, "kernel_ms_timed_rocprof"),
                         "flops_per_patch": algorithmic_flops(n, M),
                         "what": "F(n, m) x patches / MEDIAN of the per-step HIP-event times of the launch (kernel_ms_stats: mean, min, max, "
                                 "steps); kernel_ms_rocprof: the timed median of the same command under rocprofv3 --kernel-trace "
                                 "(profiles/, collected by tools/profile_r04.sh)"}}


def bench_dense_variance(env, P, n, steps, budget_s):
    """C2 with the predictive variance the reference always computes (gaussian_process::predict_measurements,
    src/gaussian_process.cpp:35-43): point-wise X* = the grid, f* and V* through gpc_dense_fit_predict_dev."""
    import torch
    from gp_compressor_amd import capi, synth
    ctx, dev = env["ctx"], env["dev"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=2)
    xs0, xs1 = synth.grid(RES, SZ)
    prm = capi.default_params_dense(want_variance=1)
    d_off, d_x0, d_x1, d_y, d_xs0, d_xs1 = t(off), t(x0), t(x1), t(y), t(xs0), t(xs1)
    d_f = torch.empty((P, 1, M), dtype=torch.float64, device=dev)
    d_v = torch.empty((P, M), dtype=torch.float64, device=dev)
    d_st = torch.empty((P,), dtype=torch.int32, device=dev)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        ctx.dense_fit_predict_dev(prm, P, d_off, n, P * n, d_x0, d_x1, d_y, 1, M, d_xs0, d_xs1, d_f, v_star=d_v, status=d_st)
        if ev is not None:
            ev[1].record()
    step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        step(ev[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kt = _kstats([a.elapsed_time(b) for a, b in ev])
    kern_ms = kt["median"]
    v_host, st = d_v.cpu().numpy(), d_st.cpu().numpy()
    ok = bool(np.all(st == 0)) and bool(np.all(np.isfinite(v_host))) and bool(np.all(v_host > -1e-12))
    fl = algorithmic_flops(n, M) + float(n) * n * M + 2.0 * n * M
    achieved = fl * P / (kern_ms * 1e-3) / 1e12
    rec = {"metric": "patches/sec (compress+predict)", "value": P * steps / elapsed, "unit": "patches/s", "n_gpus": 1, "steps": steps,
           "warmup": 1, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"C2 + predictive variance: {P} patches x {n} pts, dense fit + mean AND variance on the {SZ}x{SZ} grid "
                                  f"(V* = k** - |L^-1 k*|^2, what gaussian_process::predict_measurements computes)",
                      "patches_per_gpu": P, "points_per_patch": n, "grid_points": M, "kernel": ctx.last_dense_kernel(), "results_ok": ok},
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                        "traffic": _traffic("dense_variance@C2"), "kernel_ms": kern_ms, "kernel_ms_stats": kt, "flops_per_patch": fl,
                        "what": "F(n, m) + n^2 m + 2 n m flops per patch / HIP-event time of fit + variance kernels"}}
    if budget_s > 0:
        O = _oracle()
        p_ = O.dense_params()

        def run(lo, hi):
            sub = (off[lo:hi + 1] - off[lo]).astype(np.int32)
            sl = slice(int(off[lo]), int(off[hi]))
            return O.dense_fit_predict_batch(p_, sub, x0[sl], x1[sl], np.ascontiguousarray(y[:, sl]), xs0, xs1, variance=True, fast=True)[1]
        cores = host_cores()
        outs, done, dt, single = _timed_threads(run, P, cores, budget_s, probe=4)
        v_cpu = np.concatenate(outs, axis=0)
        diff = v_host[:done] - v_cpu
        rec["cpu_baseline"] = {"value": done / dt, "unit": "patches/s", "cores": cores, "kind": "port", "single_thread_value": single,
                               "sample": f"{done} of the {P} patches (same buffers), oracle/gpc_oracle.c -O3 -march=native, {cores} threads, {dt:.1f} s"}
        rec["rmse_vs_ref"] = {"rmse": float(np.sqrt(np.mean(diff * diff))), "max_abs": float(np.max(np.abs(diff))),
                              "f_rms": float(np.sqrt(np.mean(v_cpu * v_cpu))), "what": "GPU V* vs CPU oracle V* on the cpu_baseline sample"}
        rec["speedup_vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]
    return rec


# ------------------------------------------------------------------------------------------------ C4: sparse online GP

ISSUE_PEAK_GINST = 256 * 4 * 2.4     # wave-instructions / ns the part can issue: one per SIMD and cycle, 1024 SIMDs, 2.4 GHz


def sparse_add_bytes(sizes, cn, cap):
    """Algorithmic bytes of the add calls of one pass.  B_add = 32 b^2 bytes per point (C and Q read once, written once),
    b interpolated linearly between the measured basis sizes at the chunk boundaries (`sizes`: chunks + 1 arrays over the patches).
    A patch that arrives at an add call with >= 32 basis vectors runs that call on the LOWER TRIANGLES of C and Q (csrc/sparse.hip,
    sp_tri_pass: the regular kernel's two- and four-wave shapes, i.e. capacity > 64, Gaussian noise, GPC_SPARSE_FULL unset):
    16 b^2 bytes per point + one mirror pass (16 b^2) when it leaves the kernel -- the bytes the path has to move."""
    tri_on = cap > 64 and os.environ.get("GPC_SPARSE_FULL") is None
    tri_min = int(os.environ.get("GPC_SPARSE_TRI_MIN", "32"))
    total = 0.0
    frac = (np.arange(cn) + 0.5) / cn
    for c in range(len(sizes) - 1):
        b = sizes[c][:, None] + (sizes[c + 1] - sizes[c])[:, None] * frac[None, :]
        tri = (sizes[c] >= tri_min) & tri_on
        per_b2 = np.where(tri, 16.0, 32.0)[:, None]
        total += float(np.sum(per_b2 * b * b)) + float(np.sum(np.where(tri, 16.0, 0.0) * sizes[c + 1] ** 2))
    return total, tri_on


def sparse_add_roofline(regime, ny, bytes_total, add_stats, P, point_updates, cap=200):
    """The roofline object of a sparse record's add calls.
    fill: the basis reaches the capacity and the pass streams C and Q -- bound = HBM, achieved = algorithmic bytes / time.
    defaults (the reference's hyper-parameters: ~13 basis vectors, the blocks live in LDS): the pass moves ~1 GB and is bound by
    INSTRUCTION ISSUE -- achieved = wave-instructions the add kernels issue per pass (SQ_INSTS_* of the PMC passes,
    profiles/traffic.json, scaled by this record's point updates) / time, peak = one instruction per SIMD and cycle.  The byte
    figure stays beside it as `hbm_algorithmic_GBps`: Sigma 32 b^2 over blocks that never leave LDS is NOT what the kernels move."""
    add_ms = add_stats["median"]
    tkey = f"sparse_add@C4_{regime}" + ("" if ny == 1 else "_ny3") + ("" if cap == 200 else f"_cap{cap}")
    gbps = bytes_total / (add_ms * 1e-3) / 1e9
    if regime == "fill":
        return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "traffic": _traffic(tkey), "kernel_ms": add_ms, "kernel_ms_stats": add_stats, "bytes_per_patch": bytes_total / P,
                "what": "the add calls of one pass (rows phase + small-basis phase + regular kernel): sum over points of 32 b_t^2 bytes "
                        "(16 b_t^2 where the regular kernel works on the lower triangles of C and Q, + its mirror pass) / MEDIAN of their "
                        "HIP-event times over the timed passes"}
    per_pt = _traffic(tkey, "wave_insts_per_point_update")
    src = "profiles/traffic.json (SQ_INSTS_VALU + SALU + LDS + SMEM + VMEM of the add kernels, PMC passes)"
    if per_pt is None:
        per_pt, src = 650.0 / 4.0, "model: ~650 wave-instructions per wave-point for four patches (DESIGN 5.4a); no PMC pass on file"
    ginst = per_pt * point_updates / (add_ms * 1e-3) / 1e9
    # the VALU side of the same counters: a 64-lane FP64 operation occupies its SIMD's VALU for FOUR cycles, so the fraction of the
    # chip's VALU cycles the add kernels use is 4 x the VALU share of `frac` -- the number that says how close to instruction-bound they are
    valu_pt = _traffic(tkey, "valu_insts_per_point_update")
    valu_busy = None if valu_pt is None else 4.0 * valu_pt * point_updates / (add_ms * 1e-3) / 1e9 / ISSUE_PEAK_GINST
    return {"bound": "issue", "achieved": ginst, "peak": ISSUE_PEAK_GINST, "unit": "G wave-instructions/s", "frac": ginst / ISSUE_PEAK_GINST,
            "traffic": _traffic(tkey), "kernel_ms": add_ms, "kernel_ms_stats": add_stats,
            "wave_insts_per_point_update": per_pt, "wave_insts_source": src, "point_updates": point_updates,
            "valu_insts_per_point_update": valu_pt, "valu_cycles_frac": valu_busy,
            "hbm_algorithmic_GBps": gbps, "bytes_per_patch": bytes_total / P,
            "what": "instruction-issue roofline of the add calls (rows phase + small-basis phase + regular kernel): counted wave-instructions "
                    "per point update x point updates of the pass / MEDIAN HIP-event time, against 1024 SIMDs x 1 instruction / cycle x 2.4 GHz; "
                    "with ~13 basis vectors the state lives in LDS and HBM traffic (`traffic`) is noise"}


def bench_sparse_c4(env, regime, P, n, chunks, cap, steps, budget_s, ny=1, with_sigma=False):
    """BASELINE configs[3]: sparse_gp online updates, patches of n points streamed in `chunks` add calls, capacity `cap`,
    then predict on the grid.  regime "fill": kernel parameters under which the basis reaches the capacity (l = res/8,
    sigma_f^2 = 1, s20 = 1e-4); "defaults": the reference's own hyper-parameters (the basis stays at ~13).
    ny = 3: the colour GP the reference trains beside every depth GP (sparse_gp_field, src/gp_compressor.cpp:163, 334) at ITS
    defaults (s20 = 1e2f, eps_tol = 1e-4f).
    with_sigma: a SECOND record of the same workload whose predict also returns sigma = sqrt(s20 + k* + k^T C k), as the reference's
    predict_measurements always computes it (/root/reference/src/sparse_gp.hpp:299-351; call sites src/gp_compressor.cpp:333-334) --
    its own timed passes, its own CPU baseline (the oracle with sigma), the predict kernel's MFMA roofline on F_pred = m (2 b^2 + 9 b).
    Returns a list of records.
    Parity is stated the way tests/sparse_parity.py defines it (the regime decides branches by rounding noise): reconstruction
    RMSE against the training targets, per-patch error against the binary128 arbiter as percentiles, blow-up counts -- for the
    GPU and the fp64 oracle side by side; results_ok fails when the GPU is worse than the oracle by the frozen factors."""
    import torch
    from gp_compressor_amd import capi, synth
    ctx, dev = env["ctx"], env["dev"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=4, ny=ny)
    kw = dict(sigmaf_sq=1.0, l_sq=(RES / 8) ** 2, noise=1e-4, capacity=cap) if regime == "fill" else dict(capacity=cap)
    prm = capi.default_params_sparse(ny, **kw)
    g = capi.Sparse(ctx, prm, P, ny)
    xs0, xs1 = synth.grid(RES, SZ)
    d_xs0, d_xs1 = t(xs0), t(xs1)
    f = torch.empty((P, ny, M), dtype=torch.float64, device=dev)
    d_sig = torch.empty((P, M), dtype=torch.float64, device=dev) if with_sigma else None
    cn = n // chunks
    coff = t((np.arange(P + 1) * cn).astype(np.int32))
    bufs = []
    for c in range(chunks):
        idx = (off[:-1, None].astype(np.int64) + np.arange(c * cn, (c + 1) * cn)[None, :]).reshape(-1)
        bufs.append((t(x0[idx]), t(x1[idx]), t(y[:, idx])))
    # untimed pass: warm-up + the basis sizes at the chunk boundaries (for the algorithmic byte count)
    sizes = [np.zeros(P)]
    for c in range(chunks):
        g.add_dev(coff, cn, P * cn, *bufs[c])
        torch.cuda.synchronize()
        sizes.append(g.sizes().astype(np.float64))
    g.predict_dev(M, d_xs0, d_xs1, f, sigma=d_sig)
    torch.cuda.synchronize()
    bytes_total, tri_on = sparse_add_bytes(sizes, cn, cap)

    def timed_passes(sigma):
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(chunks + 1)] for _ in range(steps)]
        t_tot = 0.0
        for k in range(steps):
            g.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in range(chunks):
                ev[k][c][0].record()
                g.add_dev(coff, cn, P * cn, *bufs[c])
                ev[k][c][1].record()
            ev[k][chunks][0].record()
            g.predict_dev(M, d_xs0, d_xs1, f, sigma=sigma)
            ev[k][chunks][1].record()
            torch.cuda.synchronize()
            t_tot += time.perf_counter() - t0
        add = _kstats([sum(a.elapsed_time(b) for a, b in ev[k][:chunks]) for k in range(steps)])
        pred = _kstats([ev[k][chunks][0].elapsed_time(ev[k][chunks][1]) for k in range(steps)])
        return t_tot, add, pred

    t_tot, add_stats, pred_stats = timed_passes(None)
    bv = g.sizes()
    f_host = f.cpu().numpy()
    ok = bool(np.all(np.isfinite(f_host)))
    what = "sparse_gp" if ny == 1 else "sparse_gp_field (3 colour channels)"
    hyp = (" -- basis-filling kernel l=res/8, sigma_f^2=1, s20=1e-4" if regime == "fill"
           else (" -- the reference's default hyper-parameters (sigma_f^2=100, l^2=1, s20=0.1)" if ny == 1
                 else " -- the reference's default hyper-parameters of the colour GP (sigma_f^2=100, l^2=1, s20=100, eps_tol=1e-4)"))
    kern = ("sparse_add_rows_kernel<16, ny> (rows phase: four patches per wave, bases <= 16) + sparse_add_rows_kernel<32, ny, 24, true> (second "
            "rows phase: two per wave by ticket, bases <= 24) + sparse_add_kernel<true, ., ., ., 48> (mid phase: one wave per patch, bases <= 48, "
            "deletions in place) + "
            + ("sparse_add_kernel<false, false, true> (triangular passes from 32 basis vectors on)" if tri_on
               else "sparse_add_kernel<false, false>") + (" in its two-wave shape" if 64 < cap <= 100 else "")
            + " + sparse_predict_small_kernel<16 / 32> (patches of at most 32 basis vectors) + sparse_predict_kernel")

    def record(t_tot_, add_, pred_, sigma):
        return {"metric": "patches/sec (compress+predict)", "value": P * steps / t_tot_, "unit": "patches/s", "n_gpus": 1, "steps": steps,
                "warmup": 1, "ms_per_step": 1e3 * t_tot_ / steps, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"C4 {what} online ({regime}){' + sigma' if sigma else ''}: {P} patches x {n} pts streamed in {chunks} add calls, "
                                       f"capacity {cap}, then predictive mean{' AND sigma (what predict_measurements computes)' if sigma else ''} "
                                       f"on the {SZ}x{SZ} grid" + hyp,
                           "patches_per_gpu": P, "points_per_patch": n, "capacity": cap, "channels": ny, "bv_mean": float(bv.mean()),
                           "bv_max": int(bv.max()), "kernel": kern, "predict_ms": pred_["median"], "results_ok": ok},
                "roofline": sparse_add_roofline(regime, ny, bytes_total, add_, P, float(P) * n, cap)}

    rec = record(t_tot, add_stats, pred_stats, False)
    recs = [rec]
    O = op = run_threads = None
    if budget_s > 0:
        O = _oracle()
        import sparse_parity as SP
        op = O.sparse_params(ny, p0=prm.sigmaf_sq, p1=prm.l_sq, s20=prm.noise, eps_tol=prm.eps_tol, capacity=cap)

        def run_threads(sigma):
            def run(lo, hi):
                # one C call per thread range (orc_sparse_fit_predict_batch: add_measurements + predict per patch; ctypes drops the GIL)
                sub = (off[lo:hi + 1] - off[lo]).astype(np.int32)
                sl = slice(int(off[lo]), int(off[hi]))
                r_ = O.sparse_fit_predict_batch(op, sub, x0[sl], x1[sl], np.ascontiguousarray(y[:, sl]), xs0, xs1, sigma=sigma, fast=True)
                return r_[1] if sigma else r_[0]
            cores = host_cores()
            outs, done, dt, single = _timed_threads(run, P, cores, budget_s, probe=4 if regime == "fill" else 256)
            base = {"value": done / dt, "unit": "patches/s", "cores": cores, "kind": "port", "single_thread_value": single,
                    "sample": f"{done} of the {P} patches (same buffers, same insertion order), orc_sparse_fit_predict_batch "
                              f"(oracle/gpc_oracle.c -O3 -march=native{', with sigma' if sigma else ''}), one C call per thread, "
                              f"{cores} threads, {dt:.1f} s"}
            return np.concatenate(outs, axis=0), done, base
        f_cpu, done, rec["cpu_baseline"] = run_threads(False)
        diff = f_host[:done] - f_cpu
        rec["rmse_vs_ref"] = {"rmse": float(np.sqrt(np.mean(diff * diff))), "max_abs": float(np.max(np.abs(diff))),
                              "f_rms": float(np.sqrt(np.mean(f_cpu * f_cpu))),
                              "what": "GPU f* vs CPU oracle f* on the cpu_baseline sample -- in this regime two correct fp64 implementations "
                                      "differ patch by patch; the parity statement is `parity` below"}
        rec["speedup_vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]
        # the parity statement: GPU and fp64 oracle against the binary128 arbiter and against the training targets
        ft = torch.empty((ny, P * n), dtype=torch.float64, device=dev)
        d_off, d_x0, d_x1 = t(off), t(x0), t(x1)
        g.predict_points_dev(d_off, P * n, d_x0, d_x1, ft)
        torch.cuda.synchronize()
        n_arb = 512 if regime == "defaults" else 16                # the arbiter costs ~5 s per patch and thread with a full basis of 200
        o_idx = np.arange(P) if regime == "defaults" else np.arange(min(P, 512))
        par = SP.stats(op, off, x0, x1, y, xs0, xs1, f_host, ft.cpu().numpy(), np.arange(min(P, n_arb)), oracle_idx=o_idx)
        rec["parity"] = par
        rec["config"]["results_ok"] = ok = ok and par["gate"]["ok"]
        del ft, d_off, d_x0, d_x1
    if with_sigma:
        t2, add2, pred2 = timed_passes(d_sig)
        s_host = d_sig.cpu().numpy()
        ok_s = ok and bool(np.all(np.isfinite(s_host))) and bool(np.all(s_host >= 0.0))
        rs = record(t2, add2, pred2, True)
        rs["config"]["results_ok"] = ok_s
        b_ = bv.astype(np.float64)
        fl = float(np.sum(M * (2.0 * b_ * b_ + 9.0 * b_)))
        ach = fl / (pred2["median"] * 1e-3) / 1e12
        rs["roofline_predict"] = {"bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                                  "traffic": _traffic(f"sparse_predict_sigma@C4_{regime}"), "kernel_ms": pred2["median"], "kernel_ms_stats": pred2,
                                  "flops_per_patch": fl / P,
                                  "what": "sparse_predict_kernel with sigma: F_pred = m (2 b^2 + 9 b) flops per patch (SURVEY 8(d), b = the patch's "
                                          "basis size) / MEDIAN HIP-event time of the predict call; the k^T C k part runs on the MFMA pipe (DESIGN 5.6)"}
        if budget_s > 0:
            s_cpu, done, rs["cpu_baseline"] = run_threads(True)
            diff = s_host[:done] - s_cpu
            rs["rmse_vs_ref"] = {"rmse": float(np.sqrt(np.mean(diff * diff))), "max_abs": float(np.max(np.abs(diff))),
                                 "f_rms": float(np.sqrt(np.mean(s_cpu * s_cpu))),
                                 "what": "GPU sigma vs CPU oracle sigma on the cpu_baseline sample (same caveat as the mean: the states differ "
                                         "patch by patch in this regime; `parity` of the mean-only record is the statement)"}
            rs["speedup_vs_cpu_baseline"] = rs["value"] / rs["cpu_baseline"]["value"]
        recs.append(rs)
    g.close()
    return recs


def bench_sparse_c4_sharded(env, regime, P, n, chunks, cap, steps):
    """BASELINE configs[3] at N > 1 ("sparse_gp online update ... 1 -> 8 GPU scaling"): the job's patches (N x P, rank-seeded
    shards) are dealt to the ranks by gpc_partition_patches(sparse_capacity = cap) ONCE -- the per-patch state (alpha, C, Q, BV)
    lives on its GPU across the add calls (fixed affinity, as gp_mapping::train_processes keeps adding to trained GPs,
    /root/reference/src/gp_mapping.cpp:293-343) -- then `chunks` add calls, predict, and the one exchange of the path: the
    all-gather of f_star through the C-ABI's communicator + un-permutation.  P is patches PER GPU (weak scaling)."""
    import torch
    import torch.distributed as dist
    from gp_compressor_amd import capi, dist as gdist, synth
    ctx, dev, world, rank = env["ctx"], env["dev"], env["world"], env["rank"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    goff, gx0, gx1, gy = gdist.global_batch(world, P, n, res=RES, seed=4)
    Pg = len(goff) - 1
    slots, off, x0, x1, y = gdist.shard_batch(goff, gx0, gx1, gy, world, rank, sparse_capacity=cap)
    del gx0, gx1, gy
    S = slots.shape[1]
    kw = dict(sigmaf_sq=1.0, l_sq=(RES / 8) ** 2, noise=1e-4, capacity=cap) if regime == "fill" else dict(capacity=cap)
    prm = capi.default_params_sparse(1, **kw)
    g = capi.Sparse(ctx, prm, S, 1)
    xs0, xs1 = synth.grid(RES, SZ)
    d_xs0, d_xs1 = t(xs0), t(xs1)
    f = torch.empty((S, 1, M), dtype=torch.float64, device=dev)
    cnt = np.diff(off)
    cn = n // chunks
    bufs = []
    for c in range(chunks):
        ccnt = np.minimum(np.maximum(cnt - c * cn, 0), cn)                 # points of this chunk per slot (padding slots: 0)
        coff = np.zeros(S + 1, dtype=np.int32)
        coff[1:] = np.cumsum(ccnt)
        idx = np.concatenate([np.arange(off[i] + c * cn, off[i] + c * cn + ccnt[i]) for i in range(S)]) if S else np.zeros(0, np.int64)
        bufs.append((t(coff), int(ccnt.max()) if S else 0, int(coff[-1]), t(x0[idx]), t(x1[idx]), t(y[:, idx])))
    gather, exchange = gdist.make_gather(slots, Pg, f, world, rank, dev.index or 0,
                                         prefer_cabi=os.environ.get("GPC_BENCH_TORCH_GATHER") != "1")

    sizes = [np.zeros(S)]
    evs = []

    def one_pass(record_sizes=False, timed=False):
        g.reset()
        ev = []
        for c in range(chunks):
            if timed:
                ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                ev[-1][0].record()
            g.add_dev(*bufs[c])
            if timed:
                ev[-1][1].record()
            if record_sizes:
                torch.cuda.synchronize()
                sizes.append(g.sizes().astype(np.float64))
        g.predict_dev(M, d_xs0, d_xs1, f)
        gather.start(f, async_op=False)
        if timed:
            evs.append(ev)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()
    one_pass(record_sizes=True)         # warm-up + the basis sizes at the chunk boundaries (this rank's slots)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(timed=True)
    fence()
    elapsed = time.perf_counter() - t0
    add_stats = _kstats([sum(a.elapsed_time(b) for a, b in ev) for ev in evs])
    if dist.is_initialized():
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    bv = g.sizes()
    ok = bool(torch.isfinite(gather.out).all().item()) and gather.own_rows_match(f, rank, slots)
    if dist.is_initialized():
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    # roofline of rank 0's add calls (per GPU: its own slots, its own HIP-event times) -- the same model as the one-GPU records
    bytes_total, _ = sparse_add_bytes(sizes, cn, cap)
    roof = sparse_add_roofline(regime, 1, bytes_total, add_stats, max(S, 1), float(int(cnt.sum())), cap)
    roof["what"] = "rank 0's GPU: " + roof["what"]
    rec = {"metric": "patches/sec (compress+predict)", "value": world * P * steps / elapsed, "unit": "patches/s", "n_gpus": world, "steps": steps,
           "warmup": 1, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"C4 sparse_gp online ({regime}), sharded: {world * P} patches x {n} pts, {P} per GPU, streamed in {chunks} add calls "
                                  f"with fixed patch -> GPU affinity, capacity {cap}, predictive mean on the {SZ}x{SZ} grid, 1 all-gather of f_star",
                      "patches_per_gpu": P, "points_per_patch": n, "capacity": cap, "bv_mean_rank0": float(bv.mean()),
                      "parallelism": f"gpc_partition_patches(sparse_capacity={cap}) -> {S} slots per rank; state stays on its GPU across the add calls",
                      "exchange": exchange, "results_ok": ok},
           "roofline": roof}
    if hasattr(gather, "close"):
        gather.close()
    g.close()
    return rec


# ------------------------------------------------------------------------------------------------ C5: probit IRLS

def bench_irls_c5(env, P, n, steps, budget_s):
    """BASELINE configs[4]: probit occupancy-GP variant, dense Newton / IRLS loop on the GPU (gpc_dense_irls_fit_predict_dev)."""
    import torch
    from gp_compressor_amd import capi, synth
    ctx, dev = env["ctx"], env["dev"]
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    off, x0, x1, y = synth.make_patches(P, n, res=RES, seed=5)
    lab = synth.occupancy_labels(off, y[0])
    s20, l_sq, sf = 0.25, (RES / 3) ** 2, 1.0
    prm = capi.default_params_dense(sigmaf_sq=sf, l_sq=l_sq, noise=s20, noise_model=2)
    ir = capi.default_params_irls(max_iter=20, tol=1e-9)
    d_off, d_x0, d_x1, d_y = t(off), t(x0), t(x1), t(lab)
    d_f = torch.empty((P, M), dtype=torch.float64, device=dev)
    d_it = torch.empty((P,), dtype=torch.int32, device=dev)
    d_st = torch.empty((P,), dtype=torch.int32, device=dev)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        ctx.dense_irls_fit_predict_dev(prm, ir, P, d_off, n, P * n, d_x0, d_x1, d_y, M, None, None, RES, SZ, d_f, None, None, d_it, d_st)
        if ev is not None:
            ev[1].record()
    step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        step(ev[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kt = _kstats([a.elapsed_time(b) for a, b in ev])
    kern_ms = kt["median"]
    it, st, f_host = d_it.cpu().numpy(), d_st.cpu().numpy(), d_f.cpu().numpy()
    ok = bool(np.all(st == 0)) and bool(np.all(np.isfinite(f_host))) and int(it.max()) < ir.max_iter
    flops = float(np.sum(irls_flops(n, M, it.astype(np.float64))))
    achieved = flops / (kern_ms * 1e-3) / 1e12
    rec = {"metric": "patches/sec (compress+predict)", "value": P * steps / elapsed, "unit": "patches/s", "n_gpus": 1, "steps": steps,
           "warmup": 1, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"C5 probit occupancy GP: {P} patches x {n} labelled pts (+-1), RBF kernel, probit likelihood (CDF variant), "
                                  f"Newton/IRLS loop on the GPU (tol {ir.tol:g}, <= {ir.max_iter} steps) + latent mean on the {SZ}x{SZ} grid",
                      "patches_per_gpu": P, "points_per_patch": n, "newton_steps_mean": float(it.mean()), "newton_steps_max": int(it.max()),
                      "kernel": ctx.last_dense_kernel(), "results_ok": ok},
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                        "traffic": _traffic("dense_mfma_big_irls@n1024"), "kernel_ms": kern_ms, "kernel_ms_stats": kt, "flops_per_patch": flops / P,
                        "what": "sum over patches of newton_steps x (3.5 n^2 + n^3/3 + 2 n^2) + 9 n m flops / HIP-event time of the launch"}}
    if budget_s > 0:
        O = _oracle()
        op = O.dense_params(sigmaf_sq=sf, l_sq=l_sq, sigman_sq=s20)
        xs0, xs1 = O.grid(RES, SZ)

        def run(lo, hi):
            sub = (off[lo:hi + 1] - off[lo]).astype(np.int32)
            sl = slice(int(off[lo]), int(off[hi]))
            return O.dense_irls_fit_predict_batch(op, 2, sub, x0[sl], x1[sl], lab[sl], xs0, xs1, max_iter=ir.max_iter, tol=ir.tol, fast=True)[0]
        cores = host_cores()
        outs, done, dt, single = _timed_threads(run, P, cores, budget_s, probe=1)
        f_cpu = np.concatenate(outs, axis=0)
        diff = f_host[:done] - f_cpu
        rec["cpu_baseline"] = {"value": done / dt, "unit": "patches/s", "cores": cores, "kind": "port", "single_thread_value": single,
                               "sample": f"{done} of the {P} patches x {n} pts (same buffers), orc_dense_irls_fit -O3 -march=native, "
                                         f"{cores} threads, {dt:.1f} s"}
        rec["rmse_vs_ref"] = {"rmse": float(np.sqrt(np.mean(diff * diff))), "max_abs": float(np.max(np.abs(diff))),
                              "f_rms": float(np.sqrt(np.mean(f_cpu * f_cpu))), "what": "GPU latent mean vs CPU oracle on the cpu_baseline sample"}
        rec["speedup_vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=8192, help="patches per GPU (C2: 8192)")
    ap.add_argument("--points", type=int, default=256, help="points per patch (C2: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (profiling passes)")
    ap.add_argument("--only", default="", help="profiling: run just one workload -- c3 | c4fill | c4defaults | c4defaults3 | c4fills | c4defaultss (with sigma) | c4fill100 (capacity 100) | c5 | c2var -- and print its record")
    args = ap.parse_args()

    # stdout carries ONE JSON line: whatever else writes to fd 1 while the bench runs (RCCL prints a version banner there when a
    # communicator is created) is sent to stderr, and fd 1 comes back for the line itself
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from gp_compressor_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("GPC_BENCH_FORCE_DIST") == "1"      # the latter: 1-rank rehearsal of the N > 1 path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx = capi.Context(local_rank)
    # Everything of the bench runs on ONE non-default stream: torch's current stream is the legacy default stream otherwise, and on this
    # runtime a kernel on the legacy stream does not overlap with work on any other stream (measured) -- at N > 1 the
    # all-gather of step k, on its side stream, would serialise with the kernel of step k + 1 instead of hiding under it.
    if os.environ.get("GPC_BENCH_LEGACY_STREAM") != "1":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # kernels and events share torch's current stream
    env = {"ctx": ctx, "dev": dev, "world": world, "rank": rank, "use_dist": use_dist}
    cpu = world == 1 and not use_dist and not args.no_cpu_baseline
    sec_steps = max(1, min(args.steps, 5))

    if args.only:
        if args.only == "c3":
            r = bench_dense(env, 8192, 512, sec_steps, 1, seed=3)
            out = dense_record("C3 outdoor scan (one GPU's share)", r, 8192, 512, world, sec_steps, 1)
        elif args.only == "c4fill100":
            out = bench_sparse_c4(env, "fill", int(os.environ.get("GPC_C4_P", "32768")), 256, 4, 100, int(os.environ.get("GPC_C4_STEPS", "1")),
                                  float(os.environ.get("GPC_C4_CPU_S", "0")))[0]
        elif args.only in ("c4fill", "c4defaults", "c4defaults3", "c4fills", "c4defaultss"):
            sig = args.only.endswith("s")                      # c4fills / c4defaultss: the record with sigma (profiling its predict kernel)
            reg = args.only[2:].rstrip("3s")
            recs = bench_sparse_c4(env, reg, int(os.environ.get("GPC_C4_P", "32768")), 256, 4, 200, int(os.environ.get("GPC_C4_STEPS", "1")),
                                   float(os.environ.get("GPC_C4_CPU_S", "0")), ny=3 if args.only.endswith("3") else 1, with_sigma=sig)
            out = recs[-1]
        elif args.only == "c5":
            out = bench_irls_c5(env, 4096, 1024, 1, 0.0)
        elif args.only == "c2var":
            out = bench_dense_variance(env, 8192, 256, 3, 0.0)
        else:
            raise SystemExit("--only: c3 | c4fill | c4defaults | c4defaults3 | c5 | c2var")
        if rank == 0:
            emit(out)
        ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    P, n = args.patches, args.points
    _log(f"headline: dense {P} x {n}, world {world}, use_dist {use_dist}")
    r = bench_dense(env, P, n, args.steps, args.warmup)
    _log(f"headline done: {r['value']:.0f} patches/s, exchange: {r.get('exchange')}")
    name = "C2 room scan" if (P, n) == (8192, 256) else "room scan (non-default size)"
    out = dense_record(name, r, P, n, world, args.steps, args.warmup)
    if cpu:
        off, x0, x1, y, f_host = r["host"]
        # the host-pointer entry of the C-ABI (H2D of the patch buffers + kernel + D2H of f*), for the record: this
        # PCIe-inclusive rate is NOT `value` (inputs are HBM-resident in the timed region above)
        prm = capi.default_params_dense()
        ctx.dense_fit_predict_grid(prm, off, x0, x1, y, RES, SZ)
        th = time.perf_counter()
        for _ in range(3):
            ctx.dense_fit_predict_grid(prm, off, x0, x1, y, RES, SZ)
        th = (time.perf_counter() - th) / 3
        pageable = {"value": P / th, "unit": "patches/s", "ms_per_call": 1e3 * th,
                    "what": "same call on pageable numpy arrays: staged through the context's pinned buffers by a 4-thread memcpy"}
        # the entry as the reference-side binding uses it: the batch assembled in page-locked memory from
        # gpc_host_alloc, transferred in place; 8-chunk pipeline of H2D / kernel / D2H (copy-in, two compute, copy-out streams), synchronous for the caller
        pin = {k: ctx.host_array(a.shape, a.dtype) for k, a in (("off", off), ("x0", x0), ("x1", x1), ("y", y))}
        for k, a in (("off", off), ("x0", x0), ("x1", x1), ("y", y)):
            pin[k][...] = a
        pf = ctx.host_array((P, 1, M))
        pst = ctx.host_array((P,), np.int32)
        import ctypes as C_

        def pinned_call():
            rc_ = ctx.lib.gpc_dense_fit_predict_grid(ctx.h, C_.byref(prm), P, pin["off"].ctypes.data, pin["x0"].ctypes.data,
                                                     pin["x1"].ctypes.data, pin["y"].ctypes.data, 1, RES, SZ, pf.ctypes.data, None,
                                                     pst.ctypes.data)
            assert rc_ == 0, rc_
        pinned_call()
        tp = time.perf_counter()
        for _ in range(5):
            pinned_call()
        tp = (time.perf_counter() - tp) / 5
        out["host_pointer_entry"] = {"value": P / tp, "unit": "patches/s", "ms_per_call": 1e3 * tp,
                                     "results_equal_device_entry": bool(np.max(np.abs(pf - f_host)) <= 1e-12 * np.max(np.abs(f_host))),
                                     "what": "gpc_dense_fit_predict_grid with HOST buffers from gpc_host_alloc (page-locked): PCIe-inclusive, "
                                             "H2D / kernel / D2H pipelined in 8 chunks whose kernels alternate between two streams; never `value`",
                                     "pageable": pageable}
        rec, rm = cpu_baseline_dense(off, x0, x1, y, f_host, 12.0)
        out["cpu_baseline"] = rec
        out["rmse_vs_ref"] = rm
        out["speedup_vs_cpu_baseline"] = out["value"] / rec["value"]
    r["host"] = None

    secondary = []
    if not args.no_secondary:
        # C3: BASELINE configs[2] -- 8192 patches x 512 points per GPU (65536 at N = 8), same (sharded, N > 1) path
        r3 = bench_dense(env, 8192, 512, sec_steps, 1, seed=3)
        rec3 = dense_record("C3 outdoor scan" + (" (one GPU's share)" if world == 1 else ""), r3, 8192, 512, world, sec_steps, 1)
        if cpu:
            off, x0, x1, y, f_host = r3["host"]
            rec3["cpu_baseline"], rec3["rmse_vs_ref"] = cpu_baseline_dense(off, x0, x1, y, f_host, 4.0)
            rec3["speedup_vs_cpu_baseline"] = rec3["value"] / rec3["cpu_baseline"]["value"]
        r3["host"] = None
        secondary.append(rec3)
        if use_dist:
            # BASELINE configs[3] sharded: fixed affinity over the add calls, one gather (8192 patches per GPU: 5.7 GB of state each)
            for regime in ("fill", "defaults"):
                _log(f"sharded sparse C4 ({regime})")
                secondary.append(bench_sparse_c4_sharded(env, regime, 8192, 256, 4, 200, 2))
                torch.cuda.empty_cache()
        if world == 1 and not use_dist:
            secondary.append(bench_dense_variance(env, 8192, 256, 3, 3.0 if cpu else 0.0))
            torch.cuda.empty_cache()
            for regime, ny_ in (("fill", 1), ("defaults", 1), ("defaults", 3)):
                # depth plane: a second record with sigma, as the reference's predict_measurements computes it (sparse_gp.hpp:299-351)
                secondary.extend(bench_sparse_c4(env, regime, 32768, 256, 4, 200, 2, 4.0 if cpu else 0.0, ny=ny_, with_sigma=(ny_ == 1)))
                torch.cuda.empty_cache()
            # the basis-filling regime at the reference's DEFAULT capacity (100, /root/reference/src/sparse_gp.h:48): the two-wave shape
            secondary.extend(bench_sparse_c4(env, "fill", 32768, 256, 4, 100, 2, 3.0 if cpu else 0.0))
            torch.cuda.empty_cache()
            secondary.append(bench_irls_c5(env, 4096, 1024, 2, 3.0 if cpu else 0.0))
    _log("printing the line")
    if rank == 0:
        if secondary:
            out["secondary"] = secondary
        emit(out)
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
