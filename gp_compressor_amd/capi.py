"""ctypes binding of the C-ABI in include/gpc.h (gp_compressor_amd/libgpc_hip.so).

This is exactly the binding a Python caller of the reference would add; the C++ host classes in
gp_compressor_amd/host/ use the same entry points.  There is NO fallback: if the shared library has not been
built (gp_compressor_amd.build.build() / __graft_entry__.build()) or no HIP device is present, calls raise.
"""
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPC_LIB_PATH", os.path.join(HERE, "libgpc_hip.so"))   # override: diagnostic builds only

GPC_OK, GPC_EINVAL, GPC_ENOMEM, GPC_ENODEV, GPC_EHIP, GPC_ERANGE = 0, -22, -12, -19, -5, -34
STATUS_OK, STATUS_NOT_SPD, STATUS_NAN, STATUS_SIGMA_CLAMPED, STATUS_OVERFLOW, STATUS_NOT_CONVERGED = 0, 1, 2, 3, 4, 5
MAX_POINTS, MAX_BV = 1024, 256

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class Params(C.Structure):
    """struct gpc_params (include/gpc.h)."""
    _fields_ = [("sigmaf_sq", C.c_double), ("l_sq", C.c_double), ("noise", C.c_double), ("eps_tol", C.c_double),
                ("capacity", C.c_int32), ("noise_model", C.c_int32), ("ref_double_noise", C.c_int32),
                ("ref_field_delete_bug", C.c_int32), ("want_variance", C.c_int32), ("reserved", C.c_int32)]


class IrlsParams(C.Structure):
    """struct gpc_irls_params (include/gpc.h)."""
    _fields_ = [("max_iter", C.c_int32), ("reserved", C.c_int32), ("tol", C.c_double), ("f_init", C.c_double)]


class PatchesView(C.Structure):
    """struct gpc_patches_view (include/gpc.h): sizes + device addresses of a patch batch."""
    _fields_ = [("P", C.c_int32), ("n_total", C.c_int32), ("n_max", C.c_int32), ("m", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("off", "x0", "x1", "y", "rgb", "rotations", "means", "rgb_means", "W", "src")]


class GpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gpc error {code}: {msg}")
        self.code = code


# every symbol include/gpc.h declares, with its prototype (the CPU test-suite checks the library exports all of them)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
PROTOTYPES = {
    "gpc_version": (C.c_int, []),
    "gpc_default_params_dense": (None, [C.POINTER(Params)]),
    "gpc_default_params_sparse": (None, [C.POINTER(Params), _i]),
    "gpc_ctx_create": (C.c_int, [C.POINTER(_vp), _i]),
    "gpc_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "gpc_ctx_synchronize": (C.c_int, [_vp]),
    "gpc_dev_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "gpc_dev_free": (C.c_int, [_vp, _vp]),
    "gpc_dev_memcpy": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _i]),
    "gpc_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "gpc_host_free": (C.c_int, [_vp, _vp]),
    "gpc_ctx_destroy": (None, [_vp]),
    "gpc_last_error": (C.c_char_p, [_vp]),
    "gpc_last_dense_kernel": (C.c_char_p, [_vp]),
    "gpc_dense_fit_predict": (C.c_int, [_vp, C.POINTER(Params), _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gpc_dense_fit_predict_dev": (C.c_int, [_vp, C.POINTER(Params), _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp,
                                            _vp, _vp, _vp, _vp]),
    "gpc_dense_fit_predict_grid": (C.c_int, [_vp, C.POINTER(Params), _i, _vp, _vp, _vp, _vp, _i, _d, _i, _vp, _vp, _vp]),
    "gpc_dense_fit_predict_grid_dev": (C.c_int, [_vp, C.POINTER(Params), _i, _vp, _i, _i, _vp, _vp, _vp, _i, _d, _i,
                                                 _vp, _vp, _vp]),
    "gpc_default_params_irls": (None, [C.POINTER(IrlsParams)]),
    "gpc_dense_irls_fit_predict": (C.c_int, [_vp, C.POINTER(Params), C.POINTER(IrlsParams), _i, _vp, _vp, _vp, _vp, _i, _vp, _vp,
                                             _d, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpc_dense_irls_fit_predict_dev": (C.c_int, [_vp, C.POINTER(Params), C.POINTER(IrlsParams), _i, _vp, _i, _i, _vp, _vp, _vp, _i,
                                                 _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpc_noise_eval": (C.c_int, [_vp, _i, _d, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpc_sparse_create": (C.c_int, [_vp, C.POINTER(Params), _i, _i, C.POINTER(_vp)]),
    "gpc_sparse_destroy": (None, [_vp]),
    "gpc_sparse_reset": (C.c_int, [_vp]),
    "gpc_sparse_set_trace": (C.c_int, [_vp, _vp]),
    "gpc_sparse_add": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gpc_sparse_add_dev": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpc_sparse_predict": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "gpc_sparse_predict_dev": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "gpc_sparse_predict_points": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "gpc_sparse_predict_points_dev": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "gpc_sparse_likelihood": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gpc_sparse_likelihood_dev": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "gpc_sparse_train_sigmaf": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp]),
    "gpc_sparse_train_sigmaf_dev": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp]),
    "gpc_sparse_sizes": (C.c_int, [_vp, _vp]),
    "gpc_sparse_get_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "gpc_sparse_ld": (C.c_int, [_vp]),
    "gpc_sparse_set_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gpc_reproject": (C.c_int, [_vp, _i, _i] + [_vp] * 10),
    "gpc_reproject_dev": (C.c_int, [_vp, _i, _i] + [_vp] * 10),
    "gpc_project_cloud": (C.c_int, [_vp, _vp, _i, _d, _i, C.POINTER(_vp)]),
    "gpc_project_cloud_dev": (C.c_int, [_vp, _vp, _i, _d, _i, C.POINTER(_vp)]),
    "gpc_patches_view_dev": (C.c_int, [_vp, _vp]),
    "gpc_patches_fetch": (C.c_int, [_vp] * 11),
    "gpc_patches_destroy": (None, [_vp]),
    "gpc_partition_patches": (C.c_int, [_i, _vp, _i, _i, _vp]),
    "gpc_comm_unique_id": (C.c_int, [_vp]),
    "gpc_comm_create": (C.c_int, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "gpc_comm_create_all": (C.c_int, [_i, C.POINTER(_vp), C.POINTER(_vp)]),
    "gpc_comm_adopt": (C.c_int, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "gpc_comm_destroy": (None, [_vp]),
    "gpc_comm_world": (C.c_int, [_vp]),
    "gpc_comm_rank": (C.c_int, [_vp]),
    "gpc_comm_library": (C.c_char_p, []),
    "gpc_comm_set_partition": (C.c_int, [_vp, _i, _vp]),
    "gpc_group_start": (C.c_int, []),
    "gpc_group_end": (C.c_int, []),
    "gpc_allgather_fstar_dev": (C.c_int, [_vp, _i, _vp, _vp, _vp]),
    "gpc_unpermute_fstar_dev": (C.c_int, [_vp, _i, _vp, _vp]),
    "gpc_test_exp_host": (None, [_vp, _vp, _i]),
    "gpc_test_exp_small_host": (None, [_vp, _vp, _i]),
    "gpc_test_par_memcpy": (None, [_vp, _vp, C.c_size_t]),
}

_lib = None


def load():
    """Load libgpc_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy is mapped first
        # serves the whole process, and torch refuses to see the GPU when the system copy won: load torch's first.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise GpcError(GPC_ENODEV, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    """Address of a numpy array (host), a torch tensor (host or device), an int, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous(), "tensor must be contiguous"
        return a.data_ptr()
    raise TypeError(type(a))


def default_params_dense(**kw):
    p = Params()
    load().gpc_default_params_dense(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_params_sparse(ny=1, **kw):
    p = Params()
    load().gpc_default_params_sparse(C.byref(p), ny)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_params_irls(**kw):
    p = IrlsParams()
    load().gpc_default_params_irls(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Context:
    """gpc_ctx: one per process per GPU."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = _vp()
        rc = self.lib.gpc_ctx_create(C.byref(h), int(device))
        if rc != GPC_OK:
            raise GpcError(rc, "gpc_ctx_create failed (no HIP device?) -- there is no CPU fallback")
        self.h = h
        self._children = weakref.WeakSet()   # sparse handles must be destroyed before their context
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "h", None):
            for ch in list(self._children):
                ch.close()
            self.lib.gpc_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != GPC_OK:
            raise GpcError(rc, self.lib.gpc_last_error(self.h).decode())

    def set_stream(self, stream):
        """stream: raw hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream; 0 is HIP's default stream),
        or None for the context's own non-blocking stream (GPC_STREAM_OWN)."""
        self._check(self.lib.gpc_ctx_set_stream(self.h, C.c_void_p(-1) if stream is None else C.c_void_p(int(stream))))

    def synchronize(self):
        self._check(self.lib.gpc_ctx_synchronize(self.h))

    def host_array(self, shape, dtype=np.float64):
        """a numpy array in page-locked memory (gpc_host_alloc): host-pointer entries transfer such buffers in place.
        The memory lives until the process ends or free_host_array(a) is called."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _vp()
        self._check(self.lib.gpc_host_alloc(self.h, max(n, 8), C.byref(p)))
        buf = (C.c_char * max(n, 8)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def free_host_array(self, a):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p is not None:
            self._check(self.lib.gpc_host_free(self.h, p))

    def last_dense_kernel(self):
        return self.lib.gpc_last_dense_kernel(self.h).decode()

    # ---- dense, host (numpy) buffers -----------------------------------------------------------------------
    def dense_fit_predict(self, params, off, x0, x1, y, xs0, xs1, want_alpha=False):
        """Host-pointer entry (gpc_dense_fit_predict).  y: (ny, N).  Returns f_star (P, ny, m), v_star|None, status, [alpha]."""
        off = np.ascontiguousarray(off, dtype=np.int32)
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        xs0 = np.ascontiguousarray(xs0, dtype=np.float64)
        xs1 = np.ascontiguousarray(xs1, dtype=np.float64)
        P, ny, m = off.shape[0] - 1, y.shape[0], xs0.shape[0]
        f = np.full((P, ny, m), np.nan)
        v = np.full((P, m), np.nan) if params.want_variance else None
        st = np.full(P, -1, dtype=np.int32)
        al = np.full_like(y, np.nan) if want_alpha else None
        self._check(self.lib.gpc_dense_fit_predict(self.h, C.byref(params), P, _ptr(off), _ptr(x0), _ptr(x1), _ptr(y), ny,
                                                   m, _ptr(xs0), _ptr(xs1), _ptr(f), _ptr(v), _ptr(al), _ptr(st)))
        return (f, v, st, al) if want_alpha else (f, v, st)

    def dense_fit_predict_grid(self, params, off, x0, x1, y, res, sz, want_alpha=False):
        off = np.ascontiguousarray(off, dtype=np.int32)
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        P, ny, m = off.shape[0] - 1, y.shape[0], sz * sz
        f = np.full((P, ny, m), np.nan)
        st = np.full(P, -1, dtype=np.int32)
        al = np.full_like(y, np.nan) if want_alpha else None
        self._check(self.lib.gpc_dense_fit_predict_grid(self.h, C.byref(params), P, _ptr(off), _ptr(x0), _ptr(x1), _ptr(y),
                                                        ny, float(res), int(sz), _ptr(f), _ptr(al), _ptr(st)))
        return (f, st, al) if want_alpha else (f, st)

    # ---- dense GP + probit functor, Newton / IRLS loop (BASELINE config 5) ---------------------------------------------
    def dense_irls_fit_predict(self, params, irls, off, x0, x1, y, xs0=None, xs1=None, res=0.0, sz=0):
        """Host-pointer entry (gpc_dense_irls_fit_predict).  y: (N,) labels +-1.  X* point-wise (xs0, xs1) or the sz x sz grid.
        Returns f_star (P, m), alpha (N,), fhat (N,), iters (P,), status (P,)."""
        off = np.ascontiguousarray(off, dtype=np.int32)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        if xs0 is not None:
            xs0 = np.ascontiguousarray(xs0, dtype=np.float64)
            xs1 = np.ascontiguousarray(xs1, dtype=np.float64)
            m = xs0.shape[0]
        else:
            m = sz * sz
        P = off.shape[0] - 1
        f = np.full((P, m), np.nan)
        al = np.full_like(y, np.nan)
        fh = np.full_like(y, np.nan)
        it = np.full(P, -1, dtype=np.int32)
        st = np.full(P, -1, dtype=np.int32)
        self._check(self.lib.gpc_dense_irls_fit_predict(self.h, C.byref(params), C.byref(irls), P, _ptr(off), _ptr(x0), _ptr(x1), _ptr(y),
                                                        m, _ptr(xs0), _ptr(xs1), float(res), int(sz), _ptr(f), _ptr(al), _ptr(fh),
                                                        _ptr(it), _ptr(st)))
        return f, al, fh, it, st

    def dense_irls_fit_predict_dev(self, params, irls, P, off, n_max, n_total, x0, x1, y, m, xs0, xs1, res, sz, f_star,
                                   alpha_out=None, fhat_out=None, iters=None, status=None):
        self._check(self.lib.gpc_dense_irls_fit_predict_dev(self.h, C.byref(params), C.byref(irls), P, _ptr(off), n_max, n_total,
                                                            _ptr(x0), _ptr(x1), _ptr(y), m, _ptr(xs0), _ptr(xs1), float(res), int(sz),
                                                            _ptr(f_star), _ptr(alpha_out), _ptr(fhat_out), _ptr(iters), _ptr(status)))

    def noise_eval(self, noise_model, s20, y, x, sigma_x):
        """gpc_noise_eval: the device's dx_ln / dx2_ln on arrays of (y, x, sigma_x) -> q, r"""
        y, x, sx = (np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in (y, x, sigma_x))
        q = np.full_like(y, np.nan)
        r = np.full_like(y, np.nan)
        self._check(self.lib.gpc_noise_eval(self.h, int(noise_model), float(s20), len(y), _ptr(y), _ptr(x), _ptr(sx), _ptr(q), _ptr(r)))
        return q, r

    # ---- reprojection + colour clamp (row f3): predicted grids -> pcl::PointXYZRGB records ------------------------------
    POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"),
                            ("a", "u1"), ("pad", "<f4", (3,))])

    def reproject(self, xs0, xs1, f_star, rotations, means, c_star=None, rgb_means=None, bv_count=None):
        """gpc_reproject: f_star (P, m) [+ c_star (P, 3, m)] -> structured array of 32-byte points, compacted over trained patches"""
        f_star = np.ascontiguousarray(f_star, dtype=np.float64)
        P, m = f_star.shape
        xs0 = np.ascontiguousarray(xs0, dtype=np.float64)
        xs1 = np.ascontiguousarray(xs1, dtype=np.float64)
        R = np.ascontiguousarray(rotations, dtype=np.float64).reshape(P, 9)
        mu = np.ascontiguousarray(means, dtype=np.float64).reshape(P, 3)
        cs = None if c_star is None else np.ascontiguousarray(c_star, dtype=np.float64).reshape(P, 3, m)
        cm = None if rgb_means is None else np.ascontiguousarray(rgb_means, dtype=np.float64).reshape(P, 3)
        bv = None if bv_count is None else np.ascontiguousarray(bv_count, dtype=np.int32)
        cloud = np.zeros(max(P * m, 1), dtype=self.POINT_DTYPE)
        npts = np.zeros(1, dtype=np.int32)
        self._check(self.lib.gpc_reproject(self.h, P, m, _ptr(bv), _ptr(xs0), _ptr(xs1), _ptr(f_star), _ptr(cs), _ptr(R), _ptr(mu),
                                           _ptr(cm), _ptr(cloud), _ptr(npts)))
        return cloud[:int(npts[0])]

    def make_cloud(self, xyz, rgb):
        """(n, 3) float32 + (n, 3) uint8 -> array of pcl::PointXYZRGB records"""
        xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
        rgb = np.asarray(rgb, dtype=np.uint8).reshape(-1, 3)
        c = np.zeros(len(xyz), dtype=self.POINT_DTYPE)
        c["x"], c["y"], c["z"], c["w"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], 1.0
        c["r"], c["g"], c["b"], c["a"] = rgb[:, 0], rgb[:, 1], rgb[:, 2], 255
        return c

    def project_cloud(self, cloud, res, sz, n=None):
        """gpc_project_cloud[_dev]: a host record array (make_cloud) or a device buffer of n records -> Patches"""
        h = _vp()
        if isinstance(cloud, np.ndarray):
            assert cloud.dtype == self.POINT_DTYPE
            cloud = np.ascontiguousarray(cloud)
            self._check(self.lib.gpc_project_cloud(self.h, _ptr(cloud) if len(cloud) else None, len(cloud), float(res), int(sz),
                                                   C.byref(h)))
        else:
            self._check(self.lib.gpc_project_cloud_dev(self.h, _ptr(cloud), int(n), float(res), int(sz), C.byref(h)))
        return Patches(self, h)

    def reproject_dev(self, P, m, bv_count, xs0, xs1, f_star, c_star, rotations, means, rgb_means, cloud, n_points):
        self._check(self.lib.gpc_reproject_dev(self.h, P, m, _ptr(bv_count), _ptr(xs0), _ptr(xs1), _ptr(f_star), _ptr(c_star),
                                               _ptr(rotations), _ptr(means), _ptr(rgb_means), _ptr(cloud), _ptr(n_points)))

    # ---- dense, device buffers (torch tensors or raw addresses), asynchronous on the context's stream ---------
    def dense_fit_predict_dev(self, params, P, off, n_max, n_total, x0, x1, y, ny, m, xs0, xs1, f_star,
                              v_star=None, alpha_out=None, status=None):
        self._check(self.lib.gpc_dense_fit_predict_dev(self.h, C.byref(params), P, _ptr(off), n_max, n_total, _ptr(x0),
                                                       _ptr(x1), _ptr(y), ny, m, _ptr(xs0), _ptr(xs1), _ptr(f_star),
                                                       _ptr(v_star), _ptr(alpha_out), _ptr(status)))

    def dense_fit_predict_grid_dev(self, params, P, off, n_max, n_total, x0, x1, y, ny, res, sz, f_star,
                                   alpha_out=None, status=None):
        self._check(self.lib.gpc_dense_fit_predict_grid_dev(self.h, C.byref(params), P, _ptr(off), n_max, n_total,
                                                            _ptr(x0), _ptr(x1), _ptr(y), ny, float(res), int(sz),
                                                            _ptr(f_star), _ptr(alpha_out), _ptr(status)))


class Patches:
    """gpc_patches: the batch gp_compressor::project_cloud produces, resident on the device."""

    def __init__(self, ctx, h):
        self.ctx, self.lib, self.h = ctx, ctx.lib, h
        self.view = PatchesView()
        ctx._check(self.lib.gpc_patches_view_dev(h, C.byref(self.view)))
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpc_patches_destroy(self.h)
            self.h = None

    __del__ = close

    def fetch(self):
        """host copy as a dict of arrays; R[i] is the 3x3 matrix (columns normal, u, v)"""
        v = self.view
        P, N, m = v.P, v.n_total, v.m
        o = dict(off=np.zeros(P + 1, np.int32), x0=np.zeros(N), x1=np.zeros(N), y=np.zeros(N), rgb=np.zeros((3, N)),
                 R=np.zeros((P, 9)), mean=np.zeros((P, 3)), rgb_mean=np.zeros((P, 3)), W=np.zeros((P, m), np.uint8),
                 src=np.zeros(N, np.int32))
        self.ctx._check(self.lib.gpc_patches_fetch(self.h, *[_ptr(o[k]) for k in ("off", "x0", "x1", "y", "rgb", "R", "mean",
                                                                                  "rgb_mean", "W", "src")]))
        o["R"] = o["R"].reshape(P, 3, 3).transpose(0, 2, 1).copy()
        return o


class Sparse:
    """gpc_sparse: P independent sparse_gp (ny=1) / sparse_gp_field (ny=3) states resident on the device."""

    def __init__(self, ctx, params, P, ny=1):
        self.ctx, self.lib, self.P, self.ny = ctx, ctx.lib, P, ny
        h = _vp()
        ctx._check(self.lib.gpc_sparse_create(ctx.h, C.byref(params), P, ny, C.byref(h)))
        self.h = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpc_sparse_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        self.ctx._check(self.lib.gpc_sparse_reset(self.h))

    def add(self, off, x0, x1, y, perm=None, trace=False):
        """trace=True: also returns the decision bytes of the call (gpc_sparse_set_trace), (N,) uint8 in insertion order"""
        off = np.ascontiguousarray(off, dtype=np.int32)
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        assert y.shape[0] == self.ny and off.shape[0] == self.P + 1
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        st = np.full(self.P, -1, dtype=np.int32)
        tr, d_tr = None, _vp()
        if trace:
            tr = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
            self.ctx._check(self.lib.gpc_dev_malloc(self.ctx.h, tr.nbytes, C.byref(d_tr)))
            self.ctx._check(self.lib.gpc_dev_memcpy(self.ctx.h, d_tr, _ptr(tr), tr.nbytes, 1))
            self.ctx._check(self.lib.gpc_sparse_set_trace(self.h, d_tr))
        try:
            self.ctx._check(self.lib.gpc_sparse_add(self.h, _ptr(off), _ptr(x0), _ptr(x1), _ptr(y), _ptr(pm), _ptr(st)))
            if trace:
                self.ctx._check(self.lib.gpc_dev_memcpy(self.ctx.h, _ptr(tr), d_tr, tr.nbytes, 2))
        finally:
            if trace:
                self.lib.gpc_sparse_set_trace(self.h, None)
                self.lib.gpc_dev_free(self.ctx.h, d_tr)
        return (st, tr[:int(off[-1])]) if trace else st

    def add_dev(self, off, n_max, n_total, x0, x1, y, perm=None, status=None):
        self.ctx._check(self.lib.gpc_sparse_add_dev(self.h, _ptr(off), n_max, n_total, _ptr(x0), _ptr(x1), _ptr(y),
                                                    _ptr(perm), _ptr(status)))

    def predict(self, xs0, xs1, want_sigma=True, conf=False):
        xs0 = np.ascontiguousarray(xs0, dtype=np.float64)
        xs1 = np.ascontiguousarray(xs1, dtype=np.float64)
        m = xs0.shape[0]
        f = np.full((self.P, self.ny, m), np.nan)
        s = np.full((self.P, m), np.nan) if want_sigma else None
        st = np.full(self.P, -1, dtype=np.int32)
        self.ctx._check(self.lib.gpc_sparse_predict(self.h, m, _ptr(xs0), _ptr(xs1), _ptr(f), _ptr(s), int(conf), _ptr(st)))
        return f, s, st

    def predict_dev(self, m, xs0, xs1, f_star, sigma=None, conf=False, status=None):
        self.ctx._check(self.lib.gpc_sparse_predict_dev(self.h, m, _ptr(xs0), _ptr(xs1), _ptr(f_star), _ptr(sigma),
                                                        int(conf), _ptr(status)))

    def predict_points(self, off, x0, x1, want_sigma=False, conf=False):
        """predict_measurements with every patch on its own (ragged) point set: f (ny, N), sigma (N,)|None, status (P,)"""
        off = np.ascontiguousarray(off, dtype=np.int32)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        N = int(off[-1])
        assert off.shape[0] == self.P + 1 and x0.shape[0] == N and x1.shape[0] == N
        f = np.full((self.ny, N), np.nan)
        s = np.full(N, np.nan) if want_sigma else None
        st = np.full(self.P, -1, dtype=np.int32)
        self.ctx._check(self.lib.gpc_sparse_predict_points(self.h, _ptr(off), _ptr(x0), _ptr(x1), _ptr(f), _ptr(s), int(conf), _ptr(st)))
        return f, s, st

    def predict_points_dev(self, off, n_total, x0, x1, f, sigma=None, conf=False, status=None):
        self.ctx._check(self.lib.gpc_sparse_predict_points_dev(self.h, _ptr(off), int(n_total), _ptr(x0), _ptr(x1), _ptr(f),
                                                               _ptr(sigma), int(conf), _ptr(status)))

    def likelihood(self, off, x0, x1, y, want_dx=True, want_l=True):
        """compute_derivatives + compute_likelihoods (src/sparse_gp.h:44-45) on a ragged batch: dX (N, 3), l (N)"""
        off = np.ascontiguousarray(off, dtype=np.int32)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        N = int(off[-1])
        assert y.shape == (self.ny, N) and off.shape[0] == self.P + 1
        dX = np.full((N, 3), np.nan) if want_dx else None
        l = np.full(N, np.nan) if want_l else None
        self.ctx._check(self.lib.gpc_sparse_likelihood(self.h, _ptr(off), _ptr(x0), _ptr(x1), _ptr(y), _ptr(dX), _ptr(l)))
        return dX, l

    def likelihood_dev(self, off, n_total, x0, x1, y, dX=None, l=None):
        self.ctx._check(self.lib.gpc_sparse_likelihood_dev(self.h, _ptr(off), int(n_total), _ptr(x0), _ptr(x1), _ptr(y),
                                                           _ptr(dX), _ptr(l)))

    def train_sigmaf(self, off, x0, x1, y, step=float(np.float32(1e-4)), max_counter=100):
        """the live part of train_parameters (src/sparse_gp.hpp:586-640) per patch: returns p0 (P,), iters (P,),
        ls (P, max_counter + 2), delta (P, 2)"""
        off = np.ascontiguousarray(off, dtype=np.int32)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        x1 = np.ascontiguousarray(x1, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        assert off.shape[0] == self.P + 1 and y.shape[0] == int(off[-1])
        p0 = np.full(self.P, np.nan)
        iters = np.full(self.P, -1, dtype=np.int32)
        ls = np.zeros((self.P, max_counter + 2))
        delta = np.full((self.P, 2), np.nan)
        self.ctx._check(self.lib.gpc_sparse_train_sigmaf(self.h, _ptr(off), _ptr(x0), _ptr(x1), _ptr(y), float(step), int(max_counter),
                                                         _ptr(p0), _ptr(iters), _ptr(ls), _ptr(delta)))
        return p0, iters, ls, delta

    def train_sigmaf_dev(self, off, n_total, x0, x1, y, step, max_counter, p0, iters, ls, delta):
        self.ctx._check(self.lib.gpc_sparse_train_sigmaf_dev(self.h, _ptr(off), int(n_total), _ptr(x0), _ptr(x1), _ptr(y), float(step),
                                                             int(max_counter), _ptr(p0), _ptr(iters), _ptr(ls), _ptr(delta)))

    def sizes(self):
        b = np.zeros(self.P, dtype=np.int32)
        self.ctx._check(self.lib.gpc_sparse_sizes(self.h, _ptr(b)))
        return b

    def ld(self):
        return self.lib.gpc_sparse_ld(self.h)

    def state(self):
        ld = self.ld()
        alpha = np.zeros((self.P, self.ny, ld))
        Cc = np.zeros((self.P, ld, ld))
        Q = np.zeros((self.P, ld, ld))
        BV = np.zeros((self.P, ld, 2))
        self.ctx._check(self.lib.gpc_sparse_get_state(self.h, _ptr(alpha), _ptr(Cc), _ptr(Q), _ptr(BV)))
        # C, Q are column-major per patch: [p][j][i] = M(i, j)
        return alpha, np.transpose(Cc, (0, 2, 1)).copy(), np.transpose(Q, (0, 2, 1)).copy(), BV


def _sparse_set_state(self, bv_count, alpha, BV, C_=None, Q=None):
    """gpc_sparse_set_state: alpha (P, ny, ld), BV (P, ld, 2), optional C/Q (P, ld, ld) as returned by state() (row-major views)"""
    ld = self.ld()
    bv = np.ascontiguousarray(bv_count, dtype=np.int32)
    al = np.ascontiguousarray(alpha, dtype=np.float64).reshape(self.P, self.ny, ld)
    bvv = np.ascontiguousarray(BV, dtype=np.float64).reshape(self.P, ld, 2)
    cc = None if C_ is None else np.ascontiguousarray(np.transpose(C_, (0, 2, 1)))
    qq = None if Q is None else np.ascontiguousarray(np.transpose(Q, (0, 2, 1)))
    self.ctx._check(self.lib.gpc_sparse_set_state(self.h, _ptr(bv), _ptr(al), _ptr(cc), _ptr(qq), _ptr(bvv)))


Sparse.set_state = _sparse_set_state


class Comm:
    """gpc_comm: the communicator of the single all-gather (one per context).  Comm(ctx, world, rank, unique_id) creates one
    through RCCL (unique_id: the 128 bytes of Comm.unique_id() on rank 0); Comm.all(ctxs) is the one-process form."""

    def __init__(self, ctx, world=1, rank=0, unique_id=None, handle=None):
        self.ctx, self.lib = ctx, ctx.lib
        if handle is None:
            uid = unique_id if unique_id is not None else Comm.unique_id()
            buf = (C.c_char * 128).from_buffer_copy(uid)
            handle = _vp()
            ctx._check(self.lib.gpc_comm_create(ctx.h, world, rank, C.addressof(buf), C.byref(handle)))
        self.h = handle
        ctx._children.add(self)

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        rc = load().gpc_comm_unique_id(C.addressof(buf))
        if rc != GPC_OK:
            raise GpcError(rc, "gpc_comm_unique_id (RCCL not found?)")
        return bytes(buf)

    @staticmethod
    def all(ctxs):
        lib = load()
        n = len(ctxs)
        hs = (_vp * n)(*[c.h for c in ctxs])
        outs = (_vp * n)()
        rc = lib.gpc_comm_create_all(n, hs, outs)
        if rc != GPC_OK:
            raise GpcError(rc, lib.gpc_last_error(ctxs[0].h).decode())
        return [Comm(ctxs[i], handle=_vp(outs[i])) for i in range(n)]

    def close(self):
        if getattr(self, "h", None):
            self.lib.gpc_comm_destroy(self.h)
            self.h = None

    __del__ = close

    def set_partition(self, P, slots):
        sl = np.ascontiguousarray(slots, dtype=np.int32).reshape(-1)
        self.ctx._check(self.lib.gpc_comm_set_partition(self.h, int(P), _ptr(sl)))

    def allgather_fstar_dev(self, row_doubles, local_f, gathered, f_star=None):
        self.ctx._check(self.lib.gpc_allgather_fstar_dev(self.h, int(row_doubles), _ptr(local_f), _ptr(gathered), _ptr(f_star)))

    def unpermute_fstar_dev(self, row_doubles, gathered, f_star):
        self.ctx._check(self.lib.gpc_unpermute_fstar_dev(self.h, int(row_doubles), _ptr(gathered), _ptr(f_star)))


def partition_patches(off, world, sparse_capacity=0):
    """gpc_partition_patches: returns slot_patch (world, S) with patch ids (-1 = padding)."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    P = off.shape[0] - 1
    S = (P + world - 1) // world if world > 0 else 0
    out = np.zeros(max(world * S, 1), dtype=np.int32)
    rc = load().gpc_partition_patches(P, _ptr(off), world, sparse_capacity, _ptr(out))
    if rc != GPC_OK:
        raise GpcError(rc, "gpc_partition_patches")
    return out[:world * S].reshape(world, S)


def exp_host(x, small=False):
    """the kernels' exp() evaluated on the host: table-driven (any x) or the small-argument polynomial (-2^-5 <= x <= 0)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    (load().gpc_test_exp_small_host if small else load().gpc_test_exp_host)(_ptr(x), _ptr(out), x.size)
    return out
