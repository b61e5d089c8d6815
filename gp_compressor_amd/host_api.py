"""ctypes door onto the host-side C++ mirror of the reference classes (gp_compressor_amd/host -> libgpc_host.so).
Used by the tests; a C++ caller includes gp_compressor_amd/host/gp_compressor.hpp directly."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgpc_host.so")
_lib = None


def build():
    from . import build as b
    b.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host")])


def load():
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401  (see capi.load: torch's libamdhip64 must be mapped first)
        except ImportError:
            pass
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.gpc_host_create.restype = vp
        L.gpc_host_create.argtypes = [vp, vp, i, d, i, i, i]
        L.gpc_host_destroy.argtypes = [vp]
        L.gpc_host_seed.argtypes = [vp, C.c_uint]
        L.gpc_host_set_sparse_kernel.argtypes = [vp, d, d, d, d, i]
        L.gpc_host_project.argtypes = [vp]
        L.gpc_host_project_device.argtypes = [vp, vp, i]
        L.gpc_host_set_gpu_producer.argtypes = [vp, i]
        L.gpc_host_set_devices.argtypes = [vp, vp, i]
        L.gpc_host_patch_count.argtypes = [vp]
        L.gpc_host_point_count.argtypes = [vp]
        L.gpc_host_get_batch.argtypes = [vp] * 9
        L.gpc_host_get_mask.argtypes = [vp, vp]
        L.gpc_host_roundtrip.argtypes = [vp, vp, vp, i, vp, vp, vp, i]
        L.gpc_host_save_model.restype = C.c_longlong
        L.gpc_host_save_model.argtypes = [vp, C.c_char_p, vp, i]
        L.gpc_host_decompress_file.argtypes = [C.c_char_p, i, vp, vp, i, vp, i]
        _lib = L
    return _lib


class GpCompressor:
    """gp_compressor(cloud, res, sz): save_compressed() / load_compressed() (src/gp_compressor.h:65-67)."""

    def __init__(self, xyz, rgb, res=0.1, sz=10, model="sparse", device=0, seed=None):
        self.L = load()
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        self.rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        self.sz = sz
        self.h = self.L.gpc_host_create(self.xyz.ctypes.data, self.rgb.ctypes.data, len(self.xyz), float(res), int(sz),
                                        1 if model == "dense" else 0, device)
        if seed is not None:
            self.L.gpc_host_seed(self.h, seed)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.gpc_host_destroy(self.h)
            self.h = None

    def set_gpu_producer(self, on):
        """save_compressed(): cut the patches on the GPU and keep the whole round trip on the device (default), or on the host"""
        self.L.gpc_host_set_gpu_producer(self.h, int(bool(on)))

    def set_devices(self, devices):
        """multi-GPU mode of the dense model: one process drives these devices (gp_compressor::set_devices); [] switches it off"""
        d = np.ascontiguousarray(devices, dtype=np.int32)
        self.L.gpc_host_set_devices(self.h, d.ctypes.data if len(d) else None, len(d))

    def set_sparse_kernel(self, sigmaf_sq, l_sq, s20_depth, s20_rgb, capacity):
        self.L.gpc_host_set_sparse_kernel(self.h, sigmaf_sq, l_sq, s20_depth, s20_rgb, capacity)

    def project_cloud(self, device=False):
        """project_cloud() on the host, or (device=True) the same batch cut on the GPU"""
        if device:
            err = C.create_string_buffer(512)
            if self.L.gpc_host_project_device(self.h, C.addressof(err), 512) != 0:
                raise RuntimeError(f"project_cloud_device failed: {err.value.decode()}")
        else:
            assert self.L.gpc_host_project(self.h) == 0
        P, N = self.L.gpc_host_patch_count(self.h), self.L.gpc_host_point_count(self.h)
        off = np.zeros(P + 1, dtype=np.int32)
        x0, x1, y = np.zeros(N), np.zeros(N), np.zeros(N)
        rgb = np.zeros((3, N))
        R, mean, cm = np.zeros((P, 9)), np.zeros((P, 3)), np.zeros((P, 3))
        self.L.gpc_host_get_batch(self.h, *[a.ctypes.data for a in (off, x0, x1, y, rgb, R, mean, cm)])
        W = np.zeros((P, self.sz * self.sz), dtype=np.uint8)
        self.L.gpc_host_get_mask(self.h, W.ctypes.data)
        return dict(off=off, x0=x0, x1=x1, y=y, rgb=rgb, R=R.reshape(P, 3, 3).transpose(0, 2, 1).copy(), mean=mean, rgb_mean=cm, W=W)

    def roundtrip(self):
        """save_compressed("...") then load_compressed(): returns (xyz float32 (M,3), rgb uint8 (M,3), mean_added, max_added)."""
        cap = max(1, self.L.gpc_host_patch_count(self.h)) if False else 0
        assert self.L.gpc_host_project(self.h) == 0
        cap = self.L.gpc_host_patch_count(self.h) * self.sz * self.sz
        oxyz = np.zeros((max(cap, 1), 3), dtype=np.float32)
        orgb = np.zeros((max(cap, 1), 3), dtype=np.uint8)
        mean_added = C.c_double(0)
        max_added = C.c_int(0)
        err = C.create_string_buffer(512)
        n = self.L.gpc_host_roundtrip(self.h, oxyz.ctypes.data, orgb.ctypes.data, cap, C.addressof(mean_added),
                                      C.addressof(max_added), C.addressof(err), 512)
        if n < 0:
            raise RuntimeError(f"gp_compressor round trip failed ({n}): {err.value.decode()}")
        return oxyz[:n], orgb[:n], mean_added.value, max_added.value


    def save_model(self, path):
        """write the trained sparse model (frames + (BV, alpha) per patch); returns the file size in bytes"""
        err = C.create_string_buffer(512)
        n = self.L.gpc_host_save_model(self.h, os.fsencode(path), C.addressof(err), 512)
        if n < 0:
            raise RuntimeError(f"save_model failed: {err.value.decode()}")
        return int(n)


def decompress_file(path, capacity_pts, device=0):
    """reconstruct the cloud from a model file alone: returns (xyz float32 (M,3), rgb uint8 (M,3))"""
    L = load()
    oxyz = np.zeros((max(capacity_pts, 1), 3), dtype=np.float32)
    orgb = np.zeros((max(capacity_pts, 1), 3), dtype=np.uint8)
    err = C.create_string_buffer(512)
    n = L.gpc_host_decompress_file(os.fsencode(path), device, oxyz.ctypes.data, orgb.ctypes.data, capacity_pts, C.addressof(err), 512)
    if n < 0:
        raise RuntimeError(f"decompress_file failed ({n}): {err.value.decode()}")
    return oxyz[:n], orgb[:n]


def synthetic_plane_cloud(n=10000, seed=1, extent=1.2):
    """BASELINE config 1 / SURVEY section 8(d) C1 (see synth.plane_cloud)"""
    from . import synth
    return synth.plane_cloud(n, seed, extent)
