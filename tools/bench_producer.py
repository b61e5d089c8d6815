#!/usr/bin/env python3
"""Side measurement (not the headline bench): the patch producer of row f2 on the cloud behind BASELINE config 2's per-GPU
batch (~2.1 M points -> ~8100 leaves of ~256 points, res 0.15, sz 20): gpc_project_cloud_dev on a device-resident cloud,
with the C++ host producer (1 thread) timed beside it.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gp_compressor_amd import capi, host_api, synth  # noqa: E402

N = int(os.environ.get("N", "2100000"))
reps = int(os.environ.get("REPS", "5"))
res, sz = 0.15, 20
side = res * int(round((N / 259.0) ** 0.5))
xyz, rgb = synth.plane_cloud(N, seed=11, extent=side)
ctx = capi.Context(0)
cloud = ctx.make_cloud(xyz, rgb)
d_cloud = torch.from_numpy(cloud.view(np.uint8).reshape(-1, 32)).cuda()
torch.cuda.synchronize()
times = []
for r in range(reps + 1):
    t0 = time.perf_counter()
    pt = ctx.project_cloud(d_cloud, res, sz, n=N)          # synchronous: the result's sizes depend on the data
    times.append(time.perf_counter() - t0)
    v = pt.view
    P, n_total, n_max = v.P, v.n_total, v.n_max
    pt.close()
t_gpu = float(np.median(times[1:]))
t0 = time.perf_counter()
pt = ctx.project_cloud(cloud, res, sz)                     # host cloud: PCIe upload included
t_gpu_h2d = time.perf_counter() - t0
pt.close()
out = {"workload": f"project_cloud: {N} points, res {res}, sz {sz}", "P": P, "n_total": n_total, "n_max": n_max,
       "gpu_s": t_gpu, "gpu_points_per_s": N / t_gpu, "gpu_host_cloud_s": t_gpu_h2d}
if os.environ.get("CPU", "1") == "1":
    g = host_api.GpCompressor(xyz, rgb, res=res, sz=sz)
    t0 = time.perf_counter()
    g.L.gpc_host_project(g.h)
    t_cpu = time.perf_counter() - t0
    out.update(cpu_host_producer_s=t_cpu, cpu_points_per_s=N / t_cpu, speedup=t_cpu / t_gpu)
print(json.dumps(out))
