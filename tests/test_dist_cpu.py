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
        P = 21
        off, x0, x1, y = synth.make_patches(P, 48, seed=4, ragged=True)
        xs0, xs1 = synth.grid(0.15, 6)
        slots, loff, lx0, lx1, ly = gdist.shard_batch(off, x0, x1, y, world, rank)
        f, _, st = O.dense_fit_predict_batch(O.dense_params(), loff, lx0, lx1, ly, xs0, xs1)
        full = gdist.gather_fstar(torch.from_numpy(f), slots, P).numpy()
        if rank == 0:
            ref, _, _ = O.dense_fit_predict_batch(O.dense_params(), off, x0, x1, y, xs0, xs1)
            q.put((bool(np.array_equal(full, ref)), slots.tolist()))
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
    ok, slots = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    flat = [s for row in slots for s in row if s >= 0]
    assert sorted(flat) == list(range(21))
