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
