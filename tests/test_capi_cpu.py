"""CPU tests (-m "not gpu") of the C-ABI library: it loads, exports every symbol include/gpc.h declares, the host
logic (defaults, partition, argument checks, exp accuracy) is right, and there is NO CPU compute fallback."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import np_restatement as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from gp_compressor_amd import build, capi as m
    build.build()
    m.load()
    return m


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "gpc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gpc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gpc_params", "gpc_ctx", "gpc_sparse"}
    assert declared, "no declarations parsed"
    lib = C.CDLL(capi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding covers the same set
    assert declared == set(capi.PROTOTYPES), declared ^ set(capi.PROTOTYPES)
    assert lib.gpc_version() == 100


def test_params_struct_and_defaults_match_reference_constants(capi, oracle):
    p = capi.default_params_dense()
    o = oracle.dense_params()
    assert (p.sigmaf_sq, p.l_sq, p.noise) == (o.sigmaf_sq, o.l_sq, o.sigman_sq) == (0.05 * 0.05, 9.0, 0.04 * 0.04)
    assert p.ref_double_noise == 1 and p.want_variance == 0
    for ny in (1, 3):
        p = capi.default_params_sparse(ny)
        o = oracle.sparse_params(ny)
        assert (p.sigmaf_sq, p.l_sq, p.noise, p.eps_tol, p.capacity) == (o.p0, o.p1, o.s20, o.eps_tol, o.capacity)
        assert p.ref_field_delete_bug == 1
    p1, p3 = capi.default_params_sparse(1), capi.default_params_sparse(3)
    assert p1.noise == R.F(1e-1) and p1.eps_tol == R.F(1e-6) and p3.noise == 100.0 and p3.eps_tol == R.F(1e-4)
    assert C.sizeof(capi.Params) == 56


def test_no_cpu_fallback(capi):
    """Without a HIP device the context cannot be created: the product never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.GpcError) as e:
        capi.Context(0)
    assert e.value.code == capi.GPC_ENODEV


def test_exp_table_accuracy(capi):
    """The device exp() (host-compiled from the same header) stays within 2 ulp of libm over the RBF argument range."""
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.uniform(0, 50, 200000), -10.0 ** rng.uniform(-12, 2.8, 100000), rng.uniform(0, 5, 1000),
                        np.array([0.0, -0.0, -1e-300, -708.0, -745.0, -746.0, -1e6, 709.0, 710.0, 1e6])])
    got = capi.exp_host(x)
    want = np.exp(x)
    fin = np.isfinite(want) & (want > 1e-300)
    ulp = np.abs(got[fin] - want[fin]) / np.spacing(want[fin])
    assert ulp.max() <= 2.0, ulp.max()
    assert np.mean(ulp <= 0.5 + 1e-9) > 0.6
    assert got[x == -1e6][0] == 0.0 and got[x == 1e6][0] == np.inf and got[x == 0.0][0] == 1.0
    sub = (want <= 1e-300) & (want > 0)
    assert np.all(np.abs(got[sub] - want[sub]) <= 4 * np.spacing(want[sub]) + 5e-324)
    assert np.isnan(capi.exp_host(np.array([np.nan]))[0])


def test_exp_small_argument_polynomial(capi):
    """The degree-7 polynomial the register-tile kernel uses when |c| d^2 <= 2^-5 stays within 1 ulp of libm."""
    rng = np.random.default_rng(1)
    x = -np.concatenate([rng.uniform(0, 2.0 ** -5, 200000), 10.0 ** rng.uniform(-18, -1.6, 50000), [0.0, 2.0 ** -5]])
    x = x[x >= -2.0 ** -5]
    got = capi.exp_host(x, small=True)
    want = np.exp(x)
    ulp = np.abs(got - want) / np.spacing(want)
    assert ulp.max() <= 1.0, ulp.max()
    assert got[x == 0.0][0] == 1.0


def test_partition_patches(capi):
    rng = np.random.default_rng(3)
    counts = rng.integers(1, 513, size=1000)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    for world in (1, 2, 4, 8):
        slots = capi.partition_patches(off, world)
        S = (1000 + world - 1) // world
        assert slots.shape == (world, S)
        ids = slots[slots >= 0]
        assert sorted(ids.tolist()) == list(range(1000))           # every patch exactly once
        cost = counts.astype(float) ** 3
        load = np.array([cost[r[r >= 0]].sum() for r in slots])
        assert load.max() <= 1.02 * load.mean() + cost.max()       # LPT balance
    # sparse cost model and padding
    slots = capi.partition_patches(off[:11], 4, sparse_capacity=100)
    assert slots.shape == (4, 3) and (slots < 0).sum() == 2
    with pytest.raises(capi.GpcError):
        capi.partition_patches(off, 0)


def test_staging_copy_is_safe_from_two_threads(capi):
    """The staging copy of the host-pointer entries is one process-wide thread pool, while the entries serialise per context
    (include/gpc.h: thread-safe per context): two threads on two contexts reach it together.  Two threads hammer it with different
    buffers; every copy must arrive intact and nobody may dead-lock (the request fields were shared unguarded)."""
    import threading
    lib = capi.load()
    n = (3 << 20) // 8 + 123                     # > 1 MB: the multi-threaded path
    bad, done = [], []

    def worker(seed):
        rng = np.random.default_rng(seed)
        for it in range(40):
            src = rng.integers(0, 2 ** 62, size=n, dtype=np.int64)
            dst = np.zeros_like(src)
            lib.gpc_test_par_memcpy(dst.ctypes.data, src.ctypes.data, src.nbytes)
            if not np.array_equal(src, dst):
                bad.append((seed, it))
        done.append(seed)
    ths = [threading.Thread(target=worker, args=(s,), daemon=True) for s in (1, 2, 3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert sorted(done) == [1, 2, 3], "a copy never returned (lost wake-up)"
    assert not bad, bad
