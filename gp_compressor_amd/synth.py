"""Seeded synthetic patch batches of the shapes BASELINE.json names.

The buffers are generated directly in the patch frame that gp_compressor::project_points
(/root/reference/src/gp_compressor.cpp:66-118) hands to the GP: X in [-res/2, res/2]^2, depth y and colours
mean-removed per patch.  NumPy only; used by tests/, bench.py and __graft_entry__.smoke().
"""
import numpy as np


def make_patches(P, n, res=0.15, seed=2, ragged=False, ny=1, n_min=None, noise=0.003):
    """Returns off (P+1,) int32, x0, x1 (N,), y (ny, N) float64.

    ny == 1: depth plane only; ny == 3: mean-removed RGB planes (values within +-255); ny == 4: depth + RGB.
    ragged: point counts uniform in [n_min or n//2, n]."""
    rng = np.random.default_rng(seed)
    if ragged:
        lo = max(1, n // 2 if n_min is None else n_min)
        counts = rng.integers(lo, n + 1, size=P)
    else:
        counts = np.full(P, n, dtype=np.int64)
    off = np.zeros(P + 1, dtype=np.int32)
    off[1:] = np.cumsum(counts)
    N = int(off[-1])
    pid = np.repeat(np.arange(P), counts)
    x0 = rng.uniform(-res / 2, res / 2, size=N)
    x1 = rng.uniform(-res / 2, res / 2, size=N)
    # a smooth surface per patch (room wall / ground with gentle relief) + sensor noise sigma = 3 mm
    a = rng.uniform(0.005, 0.02, size=P)[pid]
    kx = rng.uniform(5.0, 30.0, size=P)[pid]
    ky = rng.uniform(5.0, 30.0, size=P)[pid]
    ph = rng.uniform(0, 2 * np.pi, size=P)[pid]
    depth = a * np.sin(kx * x0 + ph) * np.cos(ky * x1) + rng.normal(0.0, noise, size=N)
    planes = []
    if ny in (1, 4):
        planes.append(_demean(depth, off, counts))
    if ny in (3, 4):
        for c in range(3):
            tex = 127.0 + 100.0 * np.sin(10.0 * (x0 + 0.3 * c) + ph) * np.cos(7.0 * x1 - 0.2 * c)
            tex = np.clip(np.rint(tex + rng.normal(0.0, 2.0, size=N)), 0, 255)
            planes.append(_demean(tex, off, counts))
    y = np.ascontiguousarray(np.stack(planes, axis=0))
    return off, np.ascontiguousarray(x0), np.ascontiguousarray(x1), y


def _demean(v, off, counts):
    sums = np.add.reduceat(v, off[:-1].astype(np.int64)) if len(v) else np.zeros(0)
    means = sums / np.maximum(counts, 1)
    return v - np.repeat(means, counts)


def grid(res, sz):
    """The decompression grid of gp_compressor::load_compressed (/root/reference/src/gp_compressor.cpp:317-332):
    m = sz*sz, point p = y*sz + x, X*(p,0) = res*((x+.5)/sz-.5), X*(p,1) = res*((y+.5)/sz-.5)."""
    g = res * ((np.arange(sz, dtype=np.float64) + 0.5) / sz - 0.5)
    return np.ascontiguousarray(np.tile(g, sz)), np.ascontiguousarray(np.repeat(g, sz))


def sattolo_perms(off, seed=7):
    """One explicit insertion order per patch, the scheme of sparse_gp::shuffle
    (/root/reference/src/sparse_gp.hpp:43-56: for i=n-1..1: swap(ind[i], ind[rand() % i])) driven by a seeded
    generator instead of libc rand().  Returns perm (N,) int32 with patch-local indices."""
    rng = np.random.default_rng(seed)
    P = len(off) - 1
    perm = np.zeros(int(off[-1]), dtype=np.int32)
    for p in range(P):
        n = int(off[p + 1] - off[p])
        ind = np.arange(n, dtype=np.int32)
        rs = rng.integers(0, 2 ** 31 - 1, size=max(n - 1, 0))
        t = 0
        for i in range(n - 1, 0, -1):
            r = int(rs[t]) % i
            t += 1
            ind[i], ind[r] = ind[r], ind[i]
        perm[off[p]:off[p + 1]] = ind
    return perm


def plane_cloud(n=10000, seed=1, extent=1.2):
    """BASELINE config 1: z = 0.02 sin(3x) cos(2y) + N(0, 2 mm) over [0, extent)^2 with a smooth
    texture.  Returns xyz (n, 3) float32, rgb (n, 3) uint8."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, extent, n)
    y = rng.uniform(0, extent, n)
    z = 0.02 * np.sin(3 * x) * np.cos(2 * y) + rng.normal(0, 0.002, n)
    return np.stack([x, y, z], 1).astype(np.float32), _texture(x, y, x + y)


def room_cloud(n=200000, seed=3, size=(4.0, 3.0, 2.5), noise=0.003):
    """A scanned room: the six faces of a box (normals along +-x, +-y, +-z, so every branch of compute_rotation,
    src/gp_compressor.cpp:40-61, is taken) with gentle relief and sensor noise, plus a sphere standing in it (all the
    normals in between).  Returns xyz (n, 3) float32, rgb (n, 3) uint8."""
    rng = np.random.default_rng(seed)
    n_s = n // 8
    n_w = n - n_s
    face = rng.integers(0, 6, n_w)
    u, v = rng.uniform(0, 1, n_w), rng.uniform(0, 1, n_w)
    relief = 0.01 * np.sin(9 * u) * np.cos(7 * v) + rng.normal(0, noise, n_w)
    p = np.zeros((n_w, 3))
    for f in range(6):
        a = f // 2                       # axis of the normal
        b, c = (a + 1) % 3, (a + 2) % 3
        sel = face == f
        p[sel, a] = (size[a] if f % 2 else 0.0) + relief[sel]
        p[sel, b] = u[sel] * size[b]
        p[sel, c] = v[sel] * size[c]
    d = rng.normal(size=(n_s, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    s = np.array(size) / 2 + 0.6 * d * (1 + rng.normal(0, noise, (n_s, 1)))
    xyz = np.concatenate([p, s]).astype(np.float32)
    return xyz, _texture(xyz[:, 0] + xyz[:, 2], xyz[:, 1], xyz[:, 0] - xyz[:, 1])


def _texture(a, b, c):
    r = np.clip(127 + 100 * np.sin(10 * a), 0, 255)
    g = np.clip(127 + 100 * np.cos(7 * b), 0, 255)
    bl = np.clip(127 + 60 * np.sin(5 * c), 0, 255)
    return np.stack([r, g, bl], 1).astype(np.uint8)


def occupancy_labels(off, y):
    """BASELINE config 5: labels y = sign(depth - surface) in {+1, -1}.  The patch frame removes the mean
    depth, so the reference surface is depth 0: a sample above the fitted plane is +1, below -1 (ray samples either side of the
    surface; sensor noise makes the boundary ragged).  y: the depth plane of make_patches (N,)."""
    lab = np.where(np.asarray(y, dtype=np.float64).reshape(-1) >= 0.0, 1.0, -1.0)
    return np.ascontiguousarray(lab)
