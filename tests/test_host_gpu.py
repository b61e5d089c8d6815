"""GPU test of the reference-shaped host flow (BASELINE config 1 plumbing: test_gp_compress on a synthetic 10k-point
planar cloud): gp_compressor(cloud, 0.15, 20) -> save_compressed -> load_compressed (src/test_gp_compress.cpp:21-24),
with the per-patch GP loops running batched through the C-ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from gp_compressor_amd import host_api
    host_api.load()
    return host_api


def _surface(x, y):
    return 0.02 * np.sin(3 * x) * np.cos(2 * y)


def _oracle_colour_error(oracle, batch, res, sz, kw, seed):
    """mean |red - texture| of the CPU oracle's reconstruction of the SAME colour model on the SAME patch batch
    (sparse_gp_field with the test's kernel, one seeded insertion order per patch): what the colour bound is derived from"""
    from gp_compressor_amd import synth
    off = batch["off"]
    perm = synth.sattolo_perms(off, seed=seed)
    xs0, xs1 = oracle.grid(res, sz)
    errs = []
    for i in range(len(off) - 1):
        sl = slice(off[i], off[i + 1])
        if off[i + 1] == off[i]:
            continue
        g = oracle.Sparse(oracle.sparse_params(3, **kw), kw["capacity"] + 2)
        g.add_measurements(batch["x0"][sl], batch["x1"][sl], batch["rgb"][:, sl], perm[sl])
        c_star, _ = g.predict(xs0, xs1)
        red = np.clip(np.trunc(c_star[0] + batch["rgb_mean"][i, 0]), 0, 255)
        Rm, mu = batch["R"][i], batch["mean"][i]
        wx = mu[0] + Rm[0, 1] * xs0 + Rm[0, 2] * xs1                    # world x of the grid points (the depth term is < 1 mm here)
        errs.append(np.abs(red - np.clip(127 + 100 * np.sin(10 * wx), 0, 255)))
    return float(np.mean(np.concatenate(errs)))


@pytest.mark.parametrize("model", ["dense", "sparse"])
def test_compress_roundtrip_c1(H, model, oracle):
    res, sz = 0.15, 20
    xyz, rgb = H.synthetic_plane_cloud(10000, seed=1)
    g = H.GpCompressor(xyz, rgb, res=res, sz=sz, model=model, seed=7)
    if model == "sparse":
        # a kernel that lets the sparse GP follow the surface (at the reference defaults every K_ij is within 2 % of
        # sigma_f^2 and the BV set stays tiny -- covered by tests/test_sparse_gpu.py)
        g.set_sparse_kernel(1.0, (res / 2) ** 2, 1e-2, 25.0, 40)      # the CPU oracle reconstructs to 0.6 mm rms with these
    oxyz, orgb, mean_added, max_added = g.roundtrip()
    P = 64
    assert len(oxyz) == P * sz * sz                          # every patch decompresses to a full sz x sz grid
    # geometric reconstruction error against the noise-free surface (sensor noise sigma = 2 mm)
    err = oxyz[:, 2].astype(np.float64) - _surface(oxyz[:, 0].astype(np.float64), oxyz[:, 1].astype(np.float64))
    rms = float(np.sqrt(np.mean(err ** 2)))
    assert rms < 2.0e-3, rms
    # colours: the bound is DERIVED, not guessed -- the CPU oracle reconstructs the same colour model (sparse_gp_field, the
    # flow's own kernel: sigma_f^2 = 1 against colour excursions of +-100 grey levels, so the field GP smooths heavily) on
    # the same patch batch with two other insertion orders; the GPU flow's mean error must sit within 15 % + 1 grey level of
    # what the oracle gets (round 1 measured 12.7 on the GPU; the insertion order alone moves the oracle's figure by ~0.5)
    want_r = np.clip(127 + 100 * np.sin(10 * oxyz[:, 0].astype(np.float64)), 0, 255)
    e_gpu = float(np.mean(np.abs(orgb[:, 0].astype(np.float64) - want_r)))
    batch = g.project_cloud()
    kw = dict(p0=1.0, p1=(res / 2) ** 2, s20=25.0, capacity=40) if model == "sparse" else None
    if model == "sparse":
        e_orc = [_oracle_colour_error(oracle, batch, res, sz, kw, seed) for seed in (3, 4)]
        print(f"mean |red - texture|: GPU flow {e_gpu:.2f}, CPU oracle (two insertion orders) {e_orc[0]:.2f} / {e_orc[1]:.2f}")
        assert e_gpu <= 1.15 * max(e_orc) + 1.0 and e_gpu >= 0.85 * min(e_orc) - 1.0, (e_gpu, e_orc)
    else:
        # dense model: the colour planes go through the same dense GP (gaussian_process defaults) -- no insertion order, so the
        # oracle's reconstruction of the batch must give the same figure to within the clamp's rounding
        xs0, xs1 = oracle.grid(res, sz)
        c_star, _, _ = oracle.dense_fit_predict_batch(oracle.dense_params(), batch["off"], batch["x0"], batch["x1"], batch["rgb"], xs0, xs1)
        errs = []
        for i in range(len(batch["off"]) - 1):
            red = np.clip(np.trunc(c_star[i, 0] + batch["rgb_mean"][i, 0]), 0, 255)
            wx = batch["mean"][i, 0] + batch["R"][i][0, 1] * xs0 + batch["R"][i][0, 2] * xs1
            errs.append(np.abs(red - np.clip(127 + 100 * np.sin(10 * wx), 0, 255)))
        e_orc = float(np.mean(np.concatenate(errs)))
        print(f"mean |red - texture|: GPU flow {e_gpu:.2f}, CPU oracle {e_orc:.2f}")
        assert abs(e_gpu - e_orc) <= 0.5, (e_gpu, e_orc)
    if model == "sparse":
        assert 1 <= mean_added <= 40 and max_added <= 40     # "Mean added" / "Max added" (src/gp_compressor.cpp:173-174)
    else:
        assert max_added <= 400


def test_compress_c1_at_the_reference_constants(H, oracle):
    """BASELINE config 1 as the reference ships it: gp_compressor comp(cloud, 0.15f, 20); save_compressed; load_compressed
    (src/test_gp_compress.cpp:21-24) with sparse_gp(100, 1e-1f) / rbf_kernel(100, 1) / sparse_gp_field(100, 1e2f) -- no kernel
    override (src/sparse_gp.h:48, src/rbf_kernel.h:24, src/sparse_gp_field.h:43).  At these constants every K_ij lies within 2 % of
    sigma_f^2, the basis stays tiny and the branch `gamma < eps_tol` rides on rounding noise, so the statement is the statistical
    one of tests/sparse_parity.py: (1) the host class's cloud reconstructs the surface as well as the CPU oracle's reconstruction
    of the same patch batch does (other insertion orders: the host draws its own), (2) on that batch, with one explicit insertion
    order, GPU vs fp64 oracle vs binary128 arbiter: reconstruction RMSE, per-patch error percentiles, blow-ups."""
    import sparse_parity as SP
    from gp_compressor_amd import capi, synth
    res, sz = 0.15, 20
    xyz, rgb = H.synthetic_plane_cloud(10000, seed=1)
    g = H.GpCompressor(xyz, rgb, res=res, sz=sz, model="sparse", seed=7)           # set_sparse_kernel NOT called
    oxyz, orgb, mean_added, max_added = g.roundtrip()
    batch = g.project_cloud()
    off, x0, x1 = batch["off"], batch["x0"], batch["x1"]
    y = np.ascontiguousarray(batch["y"][None, :])
    P = len(off) - 1
    assert P == 64 and len(oxyz) == P * sz * sz and np.all(np.isfinite(oxyz))
    assert 1 <= mean_added <= 40 and max_added <= 64                               # "Mean added" / "Max added": the basis stays tiny
    err = oxyz[:, 2].astype(np.float64) - _surface(oxyz[:, 0].astype(np.float64), oxyz[:, 1].astype(np.float64))
    rms_host = float(np.sqrt(np.mean(err ** 2)))
    # the oracle's reconstruction of the same batch (depth GP at the defaults) under two other insertion orders
    op = oracle.sparse_params(1)
    xs0, xs1 = oracle.grid(res, sz)
    rms_orc = []
    for seed in (3, 4):
        perm = synth.sattolo_perms(off, seed=seed)
        fo, _, _ = oracle.sparse_fit_predict_batch(op, off, x0, x1, y, xs0, xs1, perm=perm)
        e = []
        for i in range(P):
            Rm, mu = batch["R"][i], batch["mean"][i]
            pts = mu[None, :] + np.outer(fo[i, 0], Rm[:, 0]) + np.outer(xs0, Rm[:, 1]) + np.outer(xs1, Rm[:, 2])
            e.append(pts[:, 2] - _surface(pts[:, 0], pts[:, 1]))
        rms_orc.append(float(np.sqrt(np.mean(np.concatenate(e) ** 2))))
    print(f"C1 at the reference constants: surface rms host class {rms_host:.3e}, oracle (two insertion orders) {rms_orc[0]:.3e} / {rms_orc[1]:.3e}")
    assert rms_host <= 1.25 * max(rms_orc) + 1e-4, (rms_host, rms_orc)
    # the same batch through the C-ABI with ONE explicit insertion order: the statistics of tests/sparse_parity.py
    ctx = capi.Context(0)
    perm = synth.sattolo_perms(off, seed=5)
    gs = capi.Sparse(ctx, capi.default_params_sparse(1), P, 1)
    assert np.all(gs.add(off, x0, x1, y, perm) == 0)
    f, _, _ = gs.predict(xs0, xs1, want_sigma=False)
    ft, _, _ = gs.predict_points(off, x0, x1)
    # (the statistics take the points in insertion order: re-lay the batch by perm)
    idx = np.concatenate([off[i] + perm[off[i]:off[i + 1]] for i in range(P)])
    st = SP.stats(op, off, x0[idx], x1[idx], np.ascontiguousarray(y[:, idx]), xs0, xs1, f, ft[:, idx], np.arange(P), full_oracle=True)
    print("C1 at the reference constants:", {k: st[k] for k in ("rmse_train", "err_vs_arbiter_abs", "blowups", "gate")})
    assert st["gate"]["ok"], st["gate"]["why"]
    gs.close()
    ctx.close()


def test_model_file_roundtrip_c1(H, tmp_path):
    """Row f3, the wire format the reference never wrote: save_model -> load_model -> load_compressed reconstructs the
    SAME cloud bit for bit from the file alone (frames + (BV, alpha) per patch), and the file is smaller than the cloud."""
    res, sz = 0.15, 20
    xyz, rgb = H.synthetic_plane_cloud(10000, seed=1)
    g = H.GpCompressor(xyz, rgb, res=res, sz=sz, model="sparse", seed=7)
    g.set_sparse_kernel(1.0, (res / 2) ** 2, 1e-2, 25.0, 40)
    oxyz, orgb, mean_added, max_added = g.roundtrip()
    path = str(tmp_path / "c1.gpcm")
    nbytes = g.save_model(path)
    import os
    assert nbytes == os.path.getsize(path) > 0
    xyz2, rgb2 = H.decompress_file(path, 64 * sz * sz)
    assert np.array_equal(xyz2.view(np.uint32), oxyz.view(np.uint32)) and np.array_equal(rgb2, orgb)
    raw = len(xyz) * 16                                      # x, y, z float + packed rgb, as PCL stores a PointXYZRGB compactly
    assert nbytes < raw, (nbytes, raw)
    # a truncated file is an error, not a crash
    with open(path, "rb") as f:
        blob = f.read()
    bad = str(tmp_path / "bad.gpcm")
    with open(bad, "wb") as f:
        f.write(blob[: len(blob) // 2])
    with pytest.raises(RuntimeError):
        H.decompress_file(bad, 64 * sz * sz)


def test_gpu_producer_equals_host_producer(H):
    """gp_compressor::project_cloud_device (gpc_project_cloud, row f2) hands train_processes the batch project_cloud()
    cuts on the host, bit for bit -- which is why save_compressed() may use either."""
    from gp_compressor_amd import synth
    for (xyz, rgb), res, sz in ((synth.plane_cloud(10000, seed=1), 0.15, 20), (synth.room_cloud(80000, seed=9), 0.15, 20)):
        g = H.GpCompressor(xyz, rgb, res=res, sz=sz)
        host = g.project_cloud()
        dev = g.project_cloud(device=True)
        assert len(host["off"]) > 10
        for k in host:
            assert np.array_equal(host[k], dev[k]), k


@pytest.mark.parametrize("model", ["dense", "sparse"])
def test_device_resident_round_trip_equals_host_buffers(H, model):
    """save_compressed() + load_compressed() with the batch resident in HBM from the producer to the reprojection
    (project_cloud_device: only `off`, the insertion orders and the final cloud cross PCIe) against the same flow through
    host buffers: the sparse kernels are deterministic -> identical clouds; the dense kernel's partial sums meet in LDS atomics
    in arrival order -> equal to a float ulp."""
    res, sz = 0.15, 20
    xyz, rgb = H.synthetic_plane_cloud(10000, seed=4)
    clouds = []
    for on_device in (True, False):
        g = H.GpCompressor(xyz, rgb, res=res, sz=sz, model=model, seed=11)
        g.set_gpu_producer(on_device)
        if model == "sparse":
            g.set_sparse_kernel(1.0, (res / 2) ** 2, 1e-2, 25.0, 40)
        oxyz, orgb, mean_added, max_added = g.roundtrip()
        clouds.append((oxyz, orgb, mean_added, max_added))
    (a_xyz, a_rgb, a_mean, a_max), (b_xyz, b_rgb, b_mean, b_max) = clouds
    assert a_xyz.shape == b_xyz.shape and (a_mean, a_max) == (b_mean, b_max)
    if model == "sparse":
        assert np.array_equal(a_xyz, b_xyz) and np.array_equal(a_rgb, b_rgb)
    else:
        assert np.max(np.abs(a_xyz - b_xyz)) <= 1e-6 and np.max(np.abs(a_rgb.astype(int) - b_rgb.astype(int))) <= 1


@pytest.mark.parametrize("model", ["dense", "sparse"])
def test_multi_gpu_mode_of_the_host_class(H, model):
    """gp_compressor::set_devices: the C++ surface -- not only the Python bench -- shards the flow: one gpc_ctx per device,
    gpc_partition_patches, per-device work, ONE RCCL all-gather (gpc_comm_create_all + gpc_group bracket), un-permute.  Dense
    model: fit + predict per device.  Sparse model (what the reference runs): the partition uses the sparse cost model, the depth
    and colour GPs of a patch are created on ITS device and stay there from train_processes() to load_compressed() (fixed
    affinity), which predicts per device and gathers once.
    The box has one GPU, so the device list is [0] (world 1: the exchange degenerates, every other step is the N-device
    code); the cloud must equal the single-device flow's -- bit for bit with the sparse model (deterministic kernels, the same
    insertion orders), to a float ulp with the dense one (its partial sums meet in LDS atomics)."""
    res, sz = 0.15, 20
    xyz, rgb = H.synthetic_plane_cloud(10000, seed=5)
    clouds = []
    for devices in (None, [0]):
        g = H.GpCompressor(xyz, rgb, res=res, sz=sz, model=model, seed=3)
        g.set_gpu_producer(False)
        if model == "sparse":
            g.set_sparse_kernel(1.0, (res / 2) ** 2, 1e-2, 25.0, 40)
        if devices is not None:
            g.set_devices(devices)
        clouds.append(g.roundtrip())
    (a_xyz, a_rgb, a_mean, a_max), (b_xyz, b_rgb, b_mean, b_max) = clouds
    assert a_xyz.shape == b_xyz.shape and len(a_xyz) == 64 * sz * sz and (a_mean, a_max) == (b_mean, b_max)
    if model == "sparse":
        assert np.array_equal(a_xyz, b_xyz) and np.array_equal(a_rgb, b_rgb)
    else:
        assert np.max(np.abs(a_xyz - b_xyz)) <= 1e-6 and np.max(np.abs(a_rgb.astype(int) - b_rgb.astype(int))) <= 1
