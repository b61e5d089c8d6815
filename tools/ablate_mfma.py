#!/usr/bin/env python3
"""Timing-only ablation study of dense_mfma_kernel (diagnostic; results of the ablated builds are wrong by design).
  python tools/ablate_mfma.py --build     (container: builds gp_compressor_amd/abl/libgpc_<variant>.so, all with coarse stamps)
  python tools/ablate_mfma.py             (GPU box: runs each variant on the C2 workload, prints the coarse phase table)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "gp_compressor_amd", "abl")
ALLC = ("-DMF_ABL_DIAG=1", "-DMF_ABL_FWD=1", "-DMF_ABL_PASS1=1", "-DMF_ABL_UPD=1", "-DMF_ABL_TRSM=1", "-DMF_ABL_YROWS=1")
VARIANTS = {
    "base": (),
    "skeleton": ALLC,
    "skeleton_nowait": ALLC + ("-DMF_ABL_NOWAIT=1",),
    "skeleton_nobar": ALLC + ("-DMF_ABL_NOBAR=1",),
    "skeleton_nowait_nobar": ALLC + ("-DMF_ABL_NOWAIT=1", "-DMF_ABL_NOBAR=1"),
    "nodiag": ("-DMF_ABL_DIAG=1",),
    "noupd": ("-DMF_ABL_UPD=1",),
    "onlydiag": ("-DMF_ABL_FWD=1", "-DMF_ABL_PASS1=1", "-DMF_ABL_UPD=1", "-DMF_ABL_TRSM=1", "-DMF_ABL_YROWS=1"),
    "onlymfma": ("-DMF_ABL_DIAG=1", "-DMF_ABL_FWD=1", "-DMF_ABL_YROWS=1"),
    "onlymfma_nowait_nobar": ("-DMF_ABL_DIAG=1", "-DMF_ABL_FWD=1", "-DMF_ABL_YROWS=1", "-DMF_ABL_NOWAIT=1", "-DMF_ABL_NOBAR=1"),
}
if "--build" in sys.argv:
    from gp_compressor_amd import build
    os.makedirs(ABL, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    todo = {k: v for k, v in VARIANTS.items() if not only or k in only}
    def one(item):
        name, flags = item
        build.build(lib=os.path.join(ABL, f"libgpc_{name}.so"), extra_flags=("-DMF_STAMPS=1",) + flags)
        print("built", name, flush=True)
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(one, todo.items()))
    sys.exit(0)
only = [a for a in sys.argv[1:] if not a.startswith("--")]
for name in VARIANTS:
    if (only and name not in only) or not os.path.exists(os.path.join(ABL, f"libgpc_{name}.so")):
        continue
    env = dict(os.environ, GPC_LIB_PATH=os.path.join(ABL, f"libgpc_{name}.so"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamp_mfma.py")], env=env, capture_output=True, text=True)
    lines = [l for l in (out.stderr + out.stdout).splitlines() if l.startswith(("load+gram", "post-loop", "backward", "predict", "total"))]
    print(f"=== {name}")
    for l in lines[-5:]:
        print("   ", l)
    sys.stdout.flush()
