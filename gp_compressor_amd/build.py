"""In-tree build of libgpc_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc.  No JIT cache: the .so lands next to
this file so that it travels to the GPU box with the repo snapshot."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpc_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        [os.path.join(os.path.dirname(HERE), "include", "gpc.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=None):
    """Compile every HIP source for gfx950 into gp_compressor_amd/libgpc_hip.so (cross-compiles without a GPU).
    `lib` + `extra_flags` build a diagnostic variant (e.g. -DMF_STAMPS -> libgpc_hip_stamps.so); its objects carry the
    variant's name, so several variants can be built concurrently."""
    if lib is None:
        if not force and not _stale():
            return LIB
        target, tag = LIB, ""
    else:
        target = lib
        tag = "." + os.path.splitext(os.path.basename(lib))[0]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libgpc_hip.so (and there is no CPU fallback)")
    # an object is rebuilt when its source, any header of csrc/ or include/gpc.h is newer (or always with force / for a variant);
    # the translation units compile side by side (the big kernels take a minute each)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        [os.path.join(os.path.dirname(HERE), "include", "gpc.h"), os.path.abspath(__file__)]
    t_hdr = max(os.path.getmtime(h) for h in hdrs)
    objs, jobs = [], []
    for src in sources():
        obj = os.path.splitext(src)[0] + tag + ".o"
        objs.append(obj)
        if not force and not tag and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), t_hdr):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-c", src, "-o", obj, *extra_flags]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append(cmd)
    nproc = max(1, min(len(jobs), int(os.environ.get("GPC_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))))
    running = []
    while jobs or running:
        while jobs and len(running) < nproc:
            c = jobs.pop(0)
            running.append((c, subprocess.Popen(c)))
        c, pr = running.pop(0)
        if pr.wait() != 0:
            for _, other in running:
                other.kill()
            raise subprocess.CalledProcessError(pr.returncode, c)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target + ".tmp", *objs, "-ldl"]
    subprocess.check_call(cmd)
    os.replace(target + ".tmp", target)
    if tag:
        for o in objs:
            os.remove(o)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
