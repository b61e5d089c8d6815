"""CPU test (-m "not gpu") of the N > 1 path: two gloo ranks shard a ragged batch with gpc_partition_patches, each
computes its patches (the CPU oracle stands in for the GPU kernel here -- this test is about the exchange, not the
arithmetic), one all-gather reassembles f_star, and the result equals the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gp_compressor_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as O
    from gp_compressor_amd import dist as gdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the code path of bench.py at N > 1: rank-seeded shards -> global batch -> LPT slots -> fit + predict -> ONE
        # all-gather -> un-permute -> self-check.  Ragged patches, and a patch count that does not divide by the world
        # size, so that the last rank carries a padding slot (an empty patch whose f* row is zero).
        goff, gx0, gx1, gy = gdist.global_batch(world, 11, 48, seed=4, ragged=True)
        P = len(goff) - 2                                    # drop the last patch: world * 11 - 1 patches
        off = goff[:P + 1]
        x0, x1, y = gx0[:off[-1]], gx1[:off[-1]], gy[:, :off[-1]]
        xs0, xs1 = synth.grid(0.15, 6)
        slots, loff, lx0, lx1, ly = gdist.shard_batch(off, x0, x1, y, world, rank)
        assert slots.shape == (world, 11) and int((slots < 0).sum()) == 1
        f, _, st = O.dense_fit_predict_batch(O.dense_params(), loff, lx0, lx1, ly, xs0, xs1)
        local = torch.from_numpy(f)
        g = gdist.ShardedGather(slots, P, local, world)
        g.start(local, async_op=True)
        full = g.finish()
        ok = g.own_rows_match(local, rank, slots)
        full2 = gdist.gather_fstar(local, slots, P)
        ok = ok and bool(torch.equal(full, full2))
        if rank == 0:
            ref, _, _ = O.dense_fit_predict_batch(O.dense_params(), off, x0, x1, y, xs0, xs1)
            q.put((ok and bool(np.array_equal(full.numpy(), ref)), slots.tolist(), P))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_allgather_roundtrip_gloo(world):
    from gp_compressor_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, slots, P = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    flat = [s for row in slots for s in row if s >= 0]
    assert sorted(flat) == list(range(P)) and P == world * 11 - 1


def _sparse_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as O
    from gp_compressor_amd import dist as gdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # BASELINE configs[3] in small, the code path of bench.py's sharded C4 record: the partition is drawn ONCE with the sparse
        # cost model (n * capacity^2), every rank keeps the GP objects of its slots alive over the add calls (fixed patch -> rank
        # affinity: the state never moves), 4 online chunks, predict, ONE all-gather, un-permute.
        cap, chunks = 12, 4
        goff, gx0, gx1, gy = gdist.global_batch(world, 7, 40, seed=14, ragged=True, n_min=9)
        P = len(goff) - 2                                    # one patch short of a full last rank: a padding slot
        off = goff[:P + 1]
        x0, x1, y = gx0[:off[-1]], gx1[:off[-1]], gy[:, :off[-1]]
        xs0, xs1 = synth.grid(0.15, 5)
        slots, loff, lx0, lx1, ly = gdist.shard_batch(off, x0, x1, y, world, rank, sparse_capacity=cap)
        S = slots.shape[1]
        op = O.sparse_params(1, p0=1.0, p1=(0.15 / 4) ** 2, s20=1e-3, capacity=cap)
        gps = [O.Sparse(op, cap + 2) for _ in range(S)]
        cnt = np.diff(loff)
        for c in range(chunks):                              # the c-th quarter of every slot's points, one add call per chunk
            for i in range(S):
                lo = loff[i] + (cnt[i] * c) // chunks
                hi = loff[i] + (cnt[i] * (c + 1)) // chunks
                if hi > lo:
                    gps[i].add_measurements(lx0[lo:hi], lx1[lo:hi], ly[:, lo:hi])
        f = np.stack([g.predict(xs0, xs1)[0] for g in gps]) if S else np.zeros((0, 1, 25))
        local = torch.from_numpy(np.ascontiguousarray(f))
        g = gdist.ShardedGather(slots, P, local, world)
        g.start(local, async_op=True)
        full = g.finish()
        ok = g.own_rows_match(local, rank, slots)
        if rank == 0:
            ref, _, bv = O.sparse_fit_predict_batch(op, off, x0, x1, y, xs0, xs1)     # one process, the points in the same order
            q.put((ok and bool(np.array_equal(full.numpy(), ref)) and int(bv.max()) == cap, slots.tolist(), P))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sparse_shard_fixed_affinity_allgather_gloo():
    """The sparse online path over two ranks: states stay on their rank across the add calls, one all-gather at the end; the
    result equals the single-process result bit for bit (the exchange and the partition, not the arithmetic, are under test)."""
    from gp_compressor_amd import build
    build.build()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, slots, P = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    flat = [s for row in slots for s in row if s >= 0]
    assert sorted(flat) == list(range(P)) and P == world * 7 - 1
