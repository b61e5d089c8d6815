"""GPU parity of gpc_reproject (reprojection + colour clamp fused, src/gp_compressor.cpp:335-373)
against the CPU oracle (orc_reproject, orc_flatten_colors).  Integer / byte / float-cast work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from gp_compressor_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    from gp_compressor_amd import capi
    capi.load()
    ctx = capi.Context(0)
    yield capi, ctx
    ctx.close()


def _oracle_cloud(oracle, xs0, xs1, f, R, mu, cs, cm, bv):
    L = oracle.lib()
    P, m = f.shape
    xyz_all, rgb_all = [], []
    xyz = (C.c_float * 3)()
    rgb = (C.c_uint8 * 3)()
    for i in range(P):
        if bv is not None and bv[i] == 0:
            continue
        Ri, mi = np.ascontiguousarray(R[i]), np.ascontiguousarray(mu[i])
        for q in range(m):
            L.orc_reproject(oracle._dp(Ri), oracle._dp(mi), float(f[i, q]), float(xs0[q]), float(xs1[q]), xyz)
            xyz_all.append((xyz[0], xyz[1], xyz[2]))
            if cs is not None:
                c3 = np.ascontiguousarray(cs[i, :, q] + cm[i])
                L.orc_flatten_colors(oracle._dp(c3), rgb)
                rgb_all.append((rgb[0], rgb[1], rgb[2]))
            else:
                rgb_all.append((0, 0, 0))
    return np.array(xyz_all, dtype=np.float32).reshape(-1, 3), np.array(rgb_all, dtype=np.uint8).reshape(-1, 3)


@pytest.mark.parametrize("P,sz,with_rgb,with_bv", [(7, 6, True, True), (3, 20, True, False), (5, 4, False, True), (1, 1, True, False)])
def test_reproject_bit_exact(gp, oracle, P, sz, with_rgb, with_bv):
    capi, ctx = gp
    rng = np.random.default_rng(100 + P)
    res, m = 0.15, sz * sz
    xs0, xs1 = oracle.grid(res, sz)
    f = rng.normal(0, 0.01, (P, m))
    # random rotations (column-major 9 doubles: columns = normal, u, v) and centres far from the origin (float rounding matters)
    R = np.zeros((P, 9))
    for i in range(P):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        R[i] = Q.T.reshape(-1)          # column-major storage of Q
    mu = rng.uniform(-50, 50, (P, 3))
    cs = cm = None
    if with_rgb:
        cs = rng.normal(0, 90, (P, 3, m))
        cm = rng.uniform(0, 255, (P, 3))
        # the clamp's corner cases: NaN / inf -> 255, negative -> 0, > 255 -> 255, > 32767 wraps as short, out of int range -> 0
        specials = [np.nan, np.inf, -np.inf, -1e-9, -0.9999, 255.0, 255.9999, 256.0, 32767.9, 32768.0, 40000.0, 65535.5, 65536.0, 65791.0,
                    1e9, 3e9, -3e9, 1e300, -1e300, 0.0, 127.5]
        flat = cs.reshape(-1)
        idx = rng.choice(flat.size, size=min(len(specials), flat.size), replace=False)
        for k, j in enumerate(idx):
            flat[j] = specials[k] - cm.reshape(-1)[(j // m) % (3 * P)] if np.isfinite(specials[k]) else specials[k]
    bv = None
    if with_bv:
        bv = rng.integers(0, 3, P).astype(np.int32)
        bv[0] = 0
        bv[-1] = 5
    cloud = ctx.reproject(xs0, xs1, f, R, mu, cs, cm, bv)
    xyz_o, rgb_o = _oracle_cloud(oracle, xs0, xs1, f, R, mu, cs, cm, bv)
    assert cloud.shape[0] == xyz_o.shape[0] == m * (P if bv is None else int(np.count_nonzero(bv)))
    got_xyz = np.stack([cloud["x"], cloud["y"], cloud["z"]], 1)
    assert np.array_equal(got_xyz.view(np.uint32), xyz_o.view(np.uint32))          # bit-exact floats
    assert np.array_equal(np.stack([cloud["r"], cloud["g"], cloud["b"]], 1), rgb_o)
    assert np.all(cloud["w"] == 1.0) and np.all(cloud["a"] == 255)
    assert cloud.dtype.itemsize == 32


def test_reproject_empty_and_errors(gp):
    capi, ctx = gp
    xs0, xs1 = synth.grid(0.15, 3)
    out = ctx.reproject(xs0, xs1, np.zeros((2, 9)), np.tile(np.eye(3).reshape(-1), (2, 1)), np.zeros((2, 3)), bv_count=np.zeros(2, np.int32))
    assert out.shape[0] == 0
    with pytest.raises(capi.GpcError) as e:        # colours without their means
        ctx.reproject(xs0, xs1, np.zeros((1, 9)), np.eye(3).reshape(1, 9), np.zeros((1, 3)), c_star=np.zeros((1, 3, 9)))
    assert e.value.code == capi.GPC_EINVAL
