#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-patch GP hot path on MI355X (contract: see the task's bench section).

Workload (BASELINE.json configs[1], "C2"): a 1M-point synthetic room scan = 8192 octree-leaf patches x 256 points
per GPU, RBF kernel + Gaussian noise, batched Cholesky fit and predictive mean on the 20 x 20 decompression grid
(m = 400) -- i.e. gp_compressor::train_processes + the patch loop of load_compressed
(/root/reference/src/gp_compressor.cpp:121-175, 298-380) with the dense gaussian_process model
(/root/reference/src/gaussian_process.cpp:15-45) on every patch.

A "step" is one pass of the hot path over the whole batch: one launch of the fused fit+predict kernel through the
C-ABI (gpc_dense_fit_predict_grid_dev) with the patch buffers already resident in HBM, followed -- when N > 1 -- by
the single RCCL all-gather that reassembles the decompressed grid values of all ranks.  Patches shard across ranks
with no other exchange (weak scaling: 8192 patches per GPU).

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


def algorithmic_flops(n, m):
    """dense fit + predictive mean, kernel evaluation = 7 flops, symmetric K counted once."""
    return 3.5 * n * n + n ** 3 / 3.0 + 2.0 * n * n + 9.0 * n * m


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


def cpu_baseline(off, x0, x1, y, res, sz, f_gpu, budget_s=12.0):
    """Times the CPU oracle (oracle/gpc_oracle.c, -O3 -march=native build; kind "port": the reference's own Eigen code
    cannot be built here) on a bounded sample of the same workload, one thread per host core, and returns
    the baseline record plus the RMSE of the GPU result against it on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from concurrent.futures import ThreadPoolExecutor
    O.build()
    p = O.dense_params()
    xs0, xs1 = O.grid(res, sz)
    P = len(off) - 1
    n = int(off[1] - off[0])

    def run(lo, hi):
        sub = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        sl = slice(int(off[lo]), int(off[hi]))
        f, _, st = O.dense_fit_predict_batch(p, sub, np.ascontiguousarray(x0[sl]), np.ascontiguousarray(x1[sl]),
                                             np.ascontiguousarray(y[:, sl]), xs0, xs1, fast=True)
        return f

    run(0, 2)                                  # warm-up (library load, page faults)
    n1 = min(64, P)
    t0 = time.perf_counter()
    run(0, n1)
    per_patch = (time.perf_counter() - t0) / n1    # single thread: how the reference itself runs (it has no threads)
    cores = host_cores()
    per_thread = max(2, min(P // cores, int(budget_s / per_patch)))
    chunks = [(i * per_thread, (i + 1) * per_thread) for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        outs = list(ex.map(lambda c: run(*c), chunks))
    dt = time.perf_counter() - t0
    done = per_thread * cores
    f_cpu = np.concatenate(outs, axis=0)
    diff = f_gpu[:done] - f_cpu
    rmse = float(np.sqrt(np.mean(diff * diff)))
    rec = {"value": done / dt, "unit": "patches/s", "cores": cores, "kind": "port", "single_thread_value": 1.0 / per_patch,
           "sample": f"{done} of the {P} patches x {n} pts (same buffers), oracle/gpc_oracle.c -O3 -march=native, "
                     f"{cores} threads, {dt:.1f} s"}
    return rec, rmse, float(np.max(np.abs(diff))), float(np.sqrt(np.mean(f_cpu * f_cpu)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=8192, help="patches per GPU (C2: 8192)")
    ap.add_argument("--points", type=int, default=256, help="points per patch (C2: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gp_compressor_amd import capi, synth

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
    if world > 1 or os.environ.get("GPC_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    P, n, res, sz = args.patches, args.points, 0.15, 20
    m = sz * sz
    # every rank owns its own 8192 patches of the scan (seeded per rank): weak scaling, no data-path exchange
    off, x0, x1, y = synth.make_patches(P, n, res=res, seed=2 + 1000 * rank)
    N = int(off[-1])
    t = lambda a: torch.from_numpy(a).to(dev)
    d_off, d_x0, d_x1, d_y = t(off), t(x0), t(x1), t(y)
    # two output buffers: with N > 1 the all-gather of step k runs on RCCL's stream while the kernel of step k+1 computes
    use_dist = world > 1 or os.environ.get("GPC_BENCH_FORCE_DIST") == "1"      # the latter: 1-rank rehearsal of the N > 1 path
    f_bufs = [torch.empty((P, 1, m), dtype=torch.float64, device=dev) for _ in range(2 if use_dist else 1)]
    status = torch.empty((P,), dtype=torch.int32, device=dev)
    g_bufs = [torch.empty((world * P, 1, m), dtype=torch.float64, device=dev) for _ in range(2)] if use_dist else None
    works = [None, None]

    ctx = capi.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # kernel and events share torch's current stream
    prm = capi.default_params_dense()                         # gaussian_process defaults, reference double-noise (F5)

    def step(k, ev=None):
        b = k & 1 if use_dist else 0
        if use_dist and works[b] is not None:
            works[b].wait()                                   # the gather that still reads this buffer (stream-level wait)
        if ev is not None:
            ev[0].record()
        ctx.dense_fit_predict_grid_dev(prm, P, d_off, n, N, d_x0, d_x1, d_y, 1, res, sz, f_bufs[b], status=status)
        if ev is not None:
            ev[1].record()
        if use_dist:
            works[b] = dist.all_gather_into_tensor(g_bufs[b], f_bufs[b], async_op=True)

    def fence():
        if use_dist:
            for w in works:
                if w is not None:
                    w.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, events[k])
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    last = (args.steps - 1) & 1 if use_dist and args.steps > 0 else 0
    f_star = f_bufs[last]
    gathered = g_bufs[last] if use_dist else None

    st = status.cpu().numpy()
    f_host = f_star.cpu().numpy()
    ok = bool(np.all(st == 0)) and bool(np.all(np.isfinite(f_host)))
    if gathered is not None:
        mine = gathered[rank * P:(rank + 1) * P].cpu().numpy()
        ok = ok and bool(np.array_equal(mine, f_host))

    if rank == 0:
        flops = algorithmic_flops(n, m) * P
        achieved = flops / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(ctx.last_dense_kernel(), {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "patches/sec (compress+predict)", "value": world * P * args.steps / elapsed, "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'C2 room scan' if (P, n) == (8192, 256) else 'room scan (non-default size)'}: {P} patches x {n} pts per GPU, RBF + Gaussian noise, dense Cholesky "
                                   f"fit + predictive mean on the {sz}x{sz} grid (m={m})",
                       "patches_per_gpu": P, "points_per_patch": n, "grid_points": m,
                       "parallelism": f"patches sharded over {world} rank(s), 1 all-gather of f_star" if world > 1 else "1 GPU",
                       "kernel": ctx.last_dense_kernel(), "results_ok": ok},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel_ms": kern_ms, "flops_per_patch": algorithmic_flops(n, m)},
        }
        if world == 1 and not args.no_cpu_baseline:
            # the host-pointer entry of the C-ABI (H2D of the patch buffers + kernel + D2H of f*), for the record: this
            # PCIe-inclusive rate is NOT `value` (inputs are HBM-resident in the timed region above)
            ctx.dense_fit_predict_grid(prm, off, x0, x1, y, res, sz)
            th = time.perf_counter()
            for _ in range(3):
                ctx.dense_fit_predict_grid(prm, off, x0, x1, y, res, sz)
            th = (time.perf_counter() - th) / 3
            out["host_pointer_entry"] = {"value": P / th, "unit": "patches/s", "ms_per_call": 1e3 * th,
                                         "what": "gpc_dense_fit_predict_grid with host buffers: H2D + kernel + D2H, synchronous"}
            rec, rmse, maxabs, frms = cpu_baseline(off, x0, x1, y, res, sz, f_host)
            out["cpu_baseline"] = rec
            out["rmse_vs_ref"] = {"rmse": rmse, "max_abs": maxabs, "f_rms": frms,
                                  "what": "GPU f* vs CPU oracle f* on the cpu_baseline sample"}
            out["speedup_vs_cpu_baseline"] = out["value"] / rec["value"]
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
