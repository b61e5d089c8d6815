"""CPU tests (-m "not gpu") of the host-side mirror of gp_compressor::project_cloud / compute_rotation / project_points
(src/gp_compressor.cpp:29-118, 177-249): the producer of the patch buffers the GPU path consumes."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def H():
    from gp_compressor_amd import host_api
    host_api.build()
    return host_api


def test_project_cloud_c1_contract(H):
    res, sz = 0.15, 20
    xyz, rgb = H.synthetic_plane_cloud(10000, seed=1)
    g = H.GpCompressor(xyz, rgb, res=res, sz=sz)
    b = g.project_cloud()
    P = len(b["off"]) - 1
    counts = np.diff(b["off"])
    # 8 x 8 occupied voxels; ownership is exclusive, and like in the reference a point that falls outside +-res/2 in the
    # tilted frame of every leaf whose search sphere reaches it is claimed by nobody (src/gp_compressor.cpp:85-87)
    assert 60 <= P <= 90 and 9900 <= counts.sum() <= 10000
    assert counts.max() <= 400 and np.median(counts[counts > 0]) > 100
    # value ranges the kernels are told to expect: X in [-res/2, res/2]^2, depth and colours mean-removed
    assert np.all(np.abs(b["x0"]) <= res / 2 + 1e-9) and np.all(np.abs(b["x1"]) <= res / 2 + 1e-9)
    for i in range(P):
        sl = slice(b["off"][i], b["off"][i + 1])
        if counts[i] == 0:
            continue
        assert abs(b["y"][sl].mean()) < 1e-9 and np.all(np.abs(b["rgb"][:, sl].mean(axis=1)) < 1e-9)
        assert np.abs(b["y"][sl]).max() < np.sqrt(3) / 2 * res
        R = b["R"][i]
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-9) and np.linalg.det(R) > 0.999      # a rotation
        assert abs(R[2, 0]) > 0.9                                                         # normal ~ +z for this cloud
    # re-projecting the patch-frame points gives back exactly the input points (each once)
    rec = []
    for i in range(P):
        sl = slice(b["off"][i], b["off"][i + 1])
        pts = np.stack([b["y"][sl], b["x0"][sl], b["x1"][sl]], 0)
        rec.append((b["R"][i] @ pts).T + b["mean"][i])
    rec = np.concatenate(rec)
    from scipy.spatial import cKDTree
    dist, idx = cKDTree(xyz.astype(np.float64)).query(rec)
    assert dist.max() < 1e-6                      # every patch-frame point is an input point ...
    assert len(np.unique(idx)) == len(idx)        # ... and no input point is used twice


def test_project_cloud_edge_cases(H):
    # empty cloud, fewer than 4 points in a leaf (identity rotation, src/gp_compressor.cpp:31-34)
    g = H.GpCompressor(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), res=0.1, sz=4)
    assert len(g.project_cloud()["off"]) == 1
    xyz = np.array([[0.01, 0.02, 0.03], [0.02, 0.01, 0.03], [0.9, 0.9, 0.9]], np.float32)
    rgb = np.array([[10, 20, 30], [30, 20, 10], [255, 0, 0]], np.uint8)
    b = H.GpCompressor(xyz, rgb, res=0.1, sz=4).project_cloud()
    assert np.diff(b["off"]).sum() == 3
    assert np.allclose(b["R"][0], np.eye(3))


def test_host_producer_equals_oracle_bit_for_bit(H, oracle):
    """The C++ host producer, the oracle (oracle/gpc_oracle_producer.c) and the GPU producer (tests/test_producer_gpu.py)
    evaluate the same expressions in the same order: identical batches, down to the last bit."""
    from gp_compressor_amd import synth
    for (xyz, rgb), res, sz in ((synth.plane_cloud(10000, seed=1), 0.15, 20), (synth.room_cloud(50000, seed=3), 0.15, 20),
                                (synth.room_cloud(20000, seed=5), 0.04, 6)):
        b = H.GpCompressor(xyz, rgb, res=res, sz=sz).project_cloud()
        o = oracle.project_cloud(xyz, rgb, res, sz)
        for k in b:
            assert np.array_equal(b[k], o[k]), k


def test_oracle_plane_fit_is_the_smallest_singular_vector(oracle):
    """compute_rotation (src/gp_compressor.cpp:35-36) takes JacobiSVD(points^T).matrixV().col(3); the oracle solves the 4x4
    moment matrix with cyclic Jacobi.  Pinned against LAPACK's SVD of the same k x 4 matrix, and the frame rules of
    :40-63 (dominant axis positive, right-handed, orthonormal)."""
    rng = np.random.default_rng(0)
    for trial in range(40):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        k = int(rng.integers(4, 400))
        basis = np.linalg.svd(n[None, :])[2][1:]                        # two in-plane directions
        pts = rng.uniform(-0.1, 0.1, (k, 2)) @ basis + rng.normal(0, 0.002, (k, 1)) * n + rng.uniform(-3, 3, 3)
        pts = pts.astype(np.float32).astype(np.float64)
        p4 = np.concatenate([pts, np.ones((k, 1))], 1)
        R = oracle.compute_rotation(p4.T @ p4, k)
        v = np.linalg.svd(p4, full_matrices=False)[2][3, :3]
        v /= np.linalg.norm(v)
        assert min(np.abs(R[:, 0] - v).max(), np.abs(R[:, 0] + v).max()) < 1e-7
        a = int(np.argmax(np.abs(R[:, 0])))
        assert R[a, 0] > 0
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        e = np.eye(3)[(a + 2) % 3]                                        # x dir: z cross n, y dir: x cross n, z dir: y cross n
        c1 = np.cross(e, R[:, 0])
        assert np.allclose(R[:, 1], c1 / np.linalg.norm(c1), atol=1e-12)
    assert np.array_equal(oracle.compute_rotation(np.eye(4), 3), np.eye(3))   # fewer than 4 points (:31-34)
