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
