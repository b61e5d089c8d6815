"""ctypes door onto the CPU oracle (oracle/liboracle.so) and the compiled reference noise objects
(oracle/_ref/libref_noise.so).  Test infrastructure only -- the product never imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


def _dp(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


def _ip(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ip)


class DenseParams(C.Structure):
    _fields_ = [("sigmaf_sq", C.c_double), ("l_sq", C.c_double), ("sigman_sq", C.c_double),
                ("ref_double_noise", C.c_int)]


class Patches(C.Structure):
    _fields_ = [("P", C.c_int), ("n_total", C.c_int), ("off", c_ip), ("x0", c_dp), ("x1", c_dp), ("y", c_dp), ("rgb", c_dp),
                ("R", c_dp), ("mean", c_dp), ("rgb_mean", c_dp), ("W", C.POINTER(C.c_uint8)), ("src", c_ip)]


class SparseParams(C.Structure):
    _fields_ = [("p0", C.c_double), ("p1", C.c_double), ("s20", C.c_double), ("eps_tol", C.c_double),
                ("capacity", C.c_int), ("ny", C.c_int), ("noise_model", C.c_int), ("field_delete_bug", C.c_int)]


def build(fast=False):
    """(Re)build the oracle with its Makefile; the _ref target is a no-op when /root/reference is absent."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)


def _load(name):
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


_lib = None
_fast = None
_ref = None


def lib(fast=False):
    global _lib, _fast
    if fast:
        if _fast is None:
            _fast = _bind(_load("liboracle_fast.so"))
        return _fast
    if _lib is None:
        _lib = _bind(_load("liboracle.so"))
    return _lib


def ref_noise():
    """The reference's own gaussian_noise/probit_noise objects, or None when oracle/_ref was not built."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_noise.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        for n in ("ref_gaussian_dx_ln", "ref_gaussian_dx2_ln", "ref_probit_dx_ln", "ref_probit_dx2_ln"):
            f = getattr(_ref, n)
            f.restype = C.c_double
            f.argtypes = [C.c_double] * 4
    return _ref


def _bind(L):
    d = C.c_double
    L.orc_rbf_kernel.restype = d
    L.orc_rbf_kernel.argtypes = [d] * 6
    L.orc_rbf_construct_covariance_fast.restype = None
    L.orc_rbf_construct_covariance_fast.argtypes = [d, d, C.c_int, c_dp, c_dp, C.c_int, c_dp, c_dp]
    for n in ("orc_gaussian_dx_ln", "orc_gaussian_dx2_ln", "orc_probit_dx_ln", "orc_probit_dx2_ln",
              "orc_probit_std_dx_ln", "orc_probit_std_dx2_ln"):
        f = getattr(L, n)
        f.restype = d
        f.argtypes = [d] * 4
    L.orc_dense_irls_fit_predict_batch.restype = C.c_int
    L.orc_dense_irls_fit_predict_batch.argtypes = [C.POINTER(DenseParams), C.c_int, C.c_int, d, d, C.c_int, c_ip, c_dp, c_dp, c_dp,
                                                   C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip]
    L.orc_gaussian3d_dx_ln.restype = None
    L.orc_gaussian3d_dx_ln.argtypes = [d, C.c_int, c_dp, c_dp, d, c_dp]
    L.orc_gaussian3d_dx2_ln.restype = d
    L.orc_gaussian3d_dx2_ln.argtypes = [d, d]
    L.orc_dense_default_params.argtypes = [C.POINTER(DenseParams)]
    L.orc_dense_fit.restype = C.c_int
    L.orc_dense_fit.argtypes = [C.POINTER(DenseParams), C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp, c_dp]
    L.orc_dense_predict.restype = None
    L.orc_dense_predict.argtypes = [C.POINTER(DenseParams), C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int,
                                    C.c_int, c_dp, c_dp, c_dp, c_dp]
    L.orc_dense_fit_predict_batch.restype = C.c_int
    L.orc_dense_fit_predict_batch.argtypes = [C.POINTER(DenseParams), C.c_int, c_ip, c_dp, c_dp, c_dp, C.c_int,
                                              C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp]
    L.orc_sparse_default_params.argtypes = [C.POINTER(SparseParams), C.c_int]
    L.orc_sparse_create.restype = C.c_void_p
    L.orc_sparse_create.argtypes = [C.POINTER(SparseParams), C.c_int]
    L.orc_sparse_destroy.argtypes = [C.c_void_p]
    L.orc_sparse_reset.argtypes = [C.c_void_p]
    L.orc_sparse_size.restype = C.c_int
    L.orc_sparse_size.argtypes = [C.c_void_p]
    L.orc_sparse_total_count.restype = C.c_int
    L.orc_sparse_total_count.argtypes = [C.c_void_p]
    L.orc_sparse_add.argtypes = [C.c_void_p, d, d, c_dp]
    L.orc_sparse_add_measurements.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_ip]
    L.orc_sparse_add_measurements_trace.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_ip, C.c_void_p]
    L.orc_sparse_delete_bv.argtypes = [C.c_void_p, C.c_int]
    L.orc_sparse_predict.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int]
    L.orc_sparse_likelihood.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.orc_sparse_train_sigmaf.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, d, C.c_int, c_dp, c_ip, c_dp, c_dp]
    L.orc_sparse_get_state.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp]
    L.orc_sparse_fit_predict_batch.restype = C.c_int
    L.orc_sparse_fit_predict_batch.argtypes = [C.POINTER(SparseParams), C.c_int, C.c_int, c_ip, c_dp, c_dp, c_dp, c_ip,
                                               C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp]
    L.orc_sparse_get_counters.argtypes = [C.c_void_p, c_ip, c_ip, c_ip]
    L.orc_shuffle_libc.argtypes = [C.c_int, c_ip]
    L.orc_shuffle_stream.argtypes = [C.c_int, C.POINTER(C.c_uint32), c_ip]
    L.orc_grid.argtypes = [d, C.c_int, c_dp, c_dp]
    L.orc_reproject.argtypes = [c_dp, c_dp, d, d, d, C.POINTER(C.c_float)]
    L.orc_flatten_colors.argtypes = [c_dp, C.POINTER(C.c_uint8)]
    L.orc_project_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_int, d, C.c_int, C.POINTER(Patches)]
    L.orc_patches_free.argtypes = [C.POINTER(Patches)]
    L.orc_compute_rotation.argtypes = [c_dp, C.c_int, c_dp]
    return L


# ---------------------------------------------------------------- pythonic wrappers

def dense_params(sigmaf_sq=None, l_sq=None, sigman_sq=None, ref_double_noise=1):
    p = DenseParams()
    lib().orc_dense_default_params(C.byref(p))
    if sigmaf_sq is not None:
        p.sigmaf_sq = sigmaf_sq
    if l_sq is not None:
        p.l_sq = l_sq
    if sigman_sq is not None:
        p.sigman_sq = sigman_sq
    p.ref_double_noise = ref_double_noise
    return p


def sparse_params(ny=1, **kw):
    p = SparseParams()
    lib().orc_sparse_default_params(C.byref(p), ny)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def grid(res, sz):
    xs0 = np.empty(sz * sz)
    xs1 = np.empty(sz * sz)
    lib().orc_grid(res, sz, _dp(xs0), _dp(xs1))
    return xs0, xs1


def dense_fit(p, x0, x1, y):
    """y: (ny, n).  Returns info, L (n x n, [row, col]), alpha (ny, n)."""
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
    ny, n = y.shape
    Lcm = np.zeros((n, n))  # column-major storage: Lcm[j, i] = L(i, j)
    alpha = np.zeros((ny, n))
    info = lib().orc_dense_fit(C.byref(p), n, _dp(x0), _dp(x1), _dp(y), ny, _dp(Lcm), _dp(alpha))
    return info, Lcm.T.copy(), alpha


def dense_predict(p, x0, x1, L, alpha, xs0, xs1, variance=False):
    n = x0.shape[0]
    ny = alpha.shape[0]
    m = xs0.shape[0]
    Lcm = np.ascontiguousarray(L.T)
    f = np.zeros((ny, m))
    v = np.zeros(m) if variance else None
    lib().orc_dense_predict(C.byref(p), n, _dp(x0), _dp(x1), _dp(Lcm), _dp(np.ascontiguousarray(alpha)), ny,
                            m, _dp(xs0), _dp(xs1), _dp(f), _dp(v))
    return f, v


def dense_fit_predict_batch(p, off, x0, x1, y, xs0, xs1, variance=False, fast=False, want_alpha=False):
    """y: (ny, N) planes.  Returns f_star (P, ny, m), v_star (P, m)|None, status (P,), [alpha (ny, N)]."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    P = off.shape[0] - 1
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
    ny = y.shape[0]
    m = xs0.shape[0]
    f = np.zeros((P, ny, m))
    v = np.zeros((P, m)) if variance else None
    st = np.zeros(P, dtype=np.int32)
    al = np.zeros_like(y) if want_alpha else None
    rc = lib(fast).orc_dense_fit_predict_batch(C.byref(p), P, _ip(off), _dp(x0), _dp(x1), _dp(y), ny,
                                               m, _dp(xs0), _dp(xs1), _dp(f), _dp(v), _ip(st), _dp(al))
    assert rc == 0
    if want_alpha:
        return f, v, st, al
    return f, v, st


def dense_irls_fit_predict_batch(p, noise_model, off, x0, x1, y, xs0, xs1, max_iter=20, tol=1e-9, f_init=0.0, fast=False):
    """BASELINE config 5 (dense GP + probit functor, Newton / IRLS loop; oracle/gpc_oracle.c).  p.sigman_sq is s20.
    y: (N,) labels.  Returns f_star (P, m), alpha (N,), fhat (N,), iters (P,), status (P,)."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    P = off.shape[0] - 1
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    m = xs0.shape[0]
    f = np.zeros((P, m))
    al = np.zeros_like(y)
    fh = np.zeros_like(y)
    it = np.zeros(P, dtype=np.int32)
    st = np.zeros(P, dtype=np.int32)
    rc = lib(fast).orc_dense_irls_fit_predict_batch(C.byref(p), int(noise_model), int(max_iter), float(tol), float(f_init), P,
                                                    _ip(off), _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(x1)), _dp(y), m,
                                                    _dp(np.ascontiguousarray(xs0)), _dp(np.ascontiguousarray(xs1)), _dp(f), _dp(al),
                                                    _dp(fh), _ip(it), _ip(st))
    assert rc == 0
    return f, al, fh, it, st


class Sparse:
    """sparse_gp / sparse_gp_field restatement handle."""

    def __init__(self, params, max_bv, fast=False):
        self.L = lib(fast)
        self.p = params
        self.ny = params.ny
        self.h = self.L.orc_sparse_create(C.byref(params), max_bv)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_sparse_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.orc_sparse_reset(self.h)

    def size(self):
        return self.L.orc_sparse_size(self.h)

    def add(self, x0, x1, y):
        yy = np.ascontiguousarray(np.atleast_1d(y), dtype=np.float64)
        self.L.orc_sparse_add(self.h, float(x0), float(x1), _dp(yy))

    def add_measurements(self, x0, x1, y, perm=None, trace=False):
        """trace=True: returns the decision bytes (one per point, insertion order; layout in oracle/gpc_oracle_hp.c)"""
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        assert y.shape[0] == self.ny
        n = y.shape[1]
        pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        tr = np.zeros(n, dtype=np.uint8) if trace else None
        self.L.orc_sparse_add_measurements_trace(self.h, n, _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(x1)),
                                                 _dp(y), _ip(pp), None if tr is None else tr.ctypes.data)
        return tr

    def delete_bv(self, loc):
        self.L.orc_sparse_delete_bv(self.h, loc)

    def predict(self, xs0, xs1, conf=False):
        m = xs0.shape[0]
        f = np.zeros((self.ny, m))
        s = np.zeros(m)
        self.L.orc_sparse_predict(self.h, m, _dp(np.ascontiguousarray(xs0)), _dp(np.ascontiguousarray(xs1)),
                                  _dp(f), _dp(s), int(conf))
        return f, s

    def likelihood(self, x0, x1, y):
        """compute_derivatives + compute_likelihoods on one point set: returns dX (n, 3), l (n)"""
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        assert y.shape[0] == self.ny
        n = y.shape[1]
        dX = np.zeros((n, 3))
        l = np.zeros(n)
        self.L.orc_sparse_likelihood(self.h, n, _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(x1)), _dp(y),
                                     _dp(dX), _dp(l))
        return dX, l

    def train_sigmaf(self, x0, x1, y, step=float(np.float32(1e-4)), max_counter=100):
        """the live part of train_parameters: returns p0, iters, ls (max_counter + 2), delta (2)"""
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        p0 = np.zeros(1)
        iters = np.zeros(1, dtype=np.int32)
        ls = np.zeros(max_counter + 2)
        delta = np.zeros(2)
        self.L.orc_sparse_train_sigmaf(self.h, len(y), _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(x1)), _dp(y),
                                       float(step), int(max_counter), _dp(p0), _ip(iters), _dp(ls), _dp(delta))
        return float(p0[0]), int(iters[0]), ls, delta

    def state(self):
        b = self.size()
        alpha = np.zeros((self.ny, b))
        Ccm = np.zeros((b, b))
        Qcm = np.zeros((b, b))
        BV = np.zeros((b, 2))
        self.L.orc_sparse_get_state(self.h, _dp(alpha), _dp(Ccm), _dp(Qcm), _dp(BV))
        return alpha, Ccm.T.copy(), Qcm.T.copy(), BV

    def counters(self):
        a = np.zeros(3, dtype=np.int32)
        self.L.orc_sparse_get_counters(self.h, _ip(a[0:1]), _ip(a[1:2]), _ip(a[2:3]))
        return tuple(int(v) for v in a)


def sparse_fit_predict_batch(p, off, x0, x1, y, xs0, xs1, perm=None, max_bv=None, sigma=False, fast=False, hp=False, train=False):
    """Per patch of a ragged batch: add_measurements (patch-local insertion order `perm`, None = identity) + predict on the grid,
    in ONE C call (oracle/gpc_oracle.c orc_sparse_fit_predict_batch; hp=True: the binary128 arbiter's twin).  y: (ny, N) planes.
    ctypes releases the GIL for the call, so thread pools over patch ranges scale.  Returns f_star (P, ny, m), sigma (P, m)|None,
    bv_count (P,) [, f_train (ny, N): the prediction at every patch's own points, when train=True]."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    P = off.shape[0] - 1
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
    assert y.shape[0] == p.ny and y.shape[1] == int(off[-1]) and int(off[0]) == 0
    m = xs0.shape[0]
    if max_bv is None:
        max_bv = (p.capacity + 2) if p.capacity > 0 else int(np.max(np.diff(off))) + 1
    f = np.zeros((P, p.ny, m))
    s = np.zeros((P, m)) if sigma else None
    bv = np.zeros(P, dtype=np.int32)
    ft = np.zeros_like(y) if train else None
    pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
    fn = hp_lib().hp_sparse_fit_predict_batch if hp else lib(fast).orc_sparse_fit_predict_batch
    rc = fn(C.byref(p), int(max_bv), P, _ip(off), _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(x1)), _dp(y), _ip(pp),
            m, _dp(np.ascontiguousarray(xs0)), _dp(np.ascontiguousarray(xs1)), _dp(f), _dp(s), _ip(bv), _dp(ft))
    assert rc == 0
    if train:
        return f, s, bv, ft
    return f, s, bv


_hp = None


def hp_lib():
    global _hp
    if _hp is None:
        L = _load("liboracle_hp.so")
        L.hp_sparse_create.restype = C.c_void_p
        L.hp_sparse_create.argtypes = [C.POINTER(SparseParams), C.c_int]
        L.hp_sparse_destroy.argtypes = [C.c_void_p]
        L.hp_sparse_size.restype = C.c_int
        L.hp_sparse_size.argtypes = [C.c_void_p]
        L.hp_sparse_add_measurements.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_ip, C.c_void_p]
        L.hp_sparse_predict.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, c_dp]
        L.hp_sparse_get_state.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp]
        L.hp_sparse_fit_predict_batch.restype = C.c_int
        L.hp_sparse_fit_predict_batch.argtypes = [C.POINTER(SparseParams), C.c_int, C.c_int, c_ip, c_dp, c_dp, c_dp, c_ip,
                                                  C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp]
        _hp = L
    return _hp


class SparseHP:
    """The binary128 arbiter of the sparse recursion (oracle/gpc_oracle_hp.c): same interface as Sparse, Gaussian noise."""

    def __init__(self, params, max_bv):
        assert params.noise_model == 0
        self.L = hp_lib()
        self.ny = params.ny
        self.h = self.L.hp_sparse_create(C.byref(params), max_bv)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hp_sparse_destroy(self.h)
            self.h = None

    def size(self):
        return self.L.hp_sparse_size(self.h)

    def add_measurements(self, x0, x1, y, perm=None, trace=False):
        y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
        assert y.shape[0] == self.ny
        n = y.shape[1]
        pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        tr = np.zeros(n, dtype=np.uint8) if trace else None
        self.L.hp_sparse_add_measurements(self.h, n, _dp(np.ascontiguousarray(x0)), _dp(np.ascontiguousarray(x1)), _dp(y), _ip(pp),
                                          None if tr is None else tr.ctypes.data)
        return tr

    def predict(self, xs0, xs1):
        m = xs0.shape[0]
        f = np.zeros((self.ny, m))
        s = np.zeros(m)
        self.L.hp_sparse_predict(self.h, m, _dp(np.ascontiguousarray(xs0)), _dp(np.ascontiguousarray(xs1)), _dp(f), _dp(s))
        return f, s

    def state(self):
        b = self.size()
        alpha = np.zeros((self.ny, b))
        Ccm = np.zeros((b, b))
        Qcm = np.zeros((b, b))
        BV = np.zeros((b, 2))
        self.L.hp_sparse_get_state(self.h, _dp(alpha), _dp(Ccm), _dp(Qcm), _dp(BV))
        return alpha, Ccm.T.copy(), Qcm.T.copy(), BV


def shuffle_stream(n, rs):
    rs = np.ascontiguousarray(rs, dtype=np.uint32)
    ind = np.zeros(n, dtype=np.int32)
    lib().orc_shuffle_stream(n, rs.ctypes.data_as(C.POINTER(C.c_uint32)), _ip(ind))
    return ind


def project_cloud(xyz, rgb, res, sz):
    """gp_compressor::project_cloud (src/gp_compressor.cpp:177-249): the patch batch as a dict of arrays; R[i] is the
    3x3 matrix (columns normal, u, v)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    o = Patches()
    rc = lib().orc_project_cloud(xyz.ctypes.data, rgb.ctypes.data, len(xyz), float(res), int(sz), C.byref(o))
    assert rc == 0
    P, N, m = o.P, o.n_total, sz * sz

    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n > 0 else np.zeros(0, dt)
    out = dict(off=arr(o.off, P + 1, np.int32), x0=arr(o.x0, N, np.float64), x1=arr(o.x1, N, np.float64),
               y=arr(o.y, N, np.float64), rgb=arr(o.rgb, 3 * N, np.float64).reshape(3, N),
               R=arr(o.R, 9 * P, np.float64).reshape(P, 3, 3).transpose(0, 2, 1).copy(),
               mean=arr(o.mean, 3 * P, np.float64).reshape(P, 3), rgb_mean=arr(o.rgb_mean, 3 * P, np.float64).reshape(P, 3),
               W=arr(o.W, P * m, np.uint8).reshape(P, m), src=arr(o.src, N, np.int32))
    lib().orc_patches_free(C.byref(o))
    return out


def compute_rotation(M, k):
    """compute_rotation (src/gp_compressor.cpp:29-64) from the 4x4 moment matrix of the k homogeneous points"""
    A = np.ascontiguousarray(M, dtype=np.float64).copy()
    R = np.zeros(9)
    lib().orc_compute_rotation(_dp(A.reshape(-1)), int(k), _dp(R))
    return R.reshape(3, 3).T.copy()
