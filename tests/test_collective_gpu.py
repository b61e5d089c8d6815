"""GPU test of the C-ABI's multi-GPU entry points (include/gpc.h "multi-GPU"): a 1-rank RCCL communicator
on the one GPU of the box -- the all-gather degenerates to a copy, but the partition table, its inverse on the device, the
un-permutation kernel, the run-time binding of RCCL and both creation paths (gpc_comm_create with a unique id, the
single-process gpc_comm_create_all with a gpc_group bracket) are the code the N-rank run executes.  The N > 1 exchange
itself is rehearsed on CPU ranks with gloo (tests/test_dist_cpu.py) and runs in bench.py at N > 1."""
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


def _partition(capi, P, seed):
    off, _, _, _ = synth.make_patches(P, 64, seed=seed, ragged=True)
    slots = capi.partition_patches(off, 1)          # world 1: LPT order = patches sorted by cost, a non-trivial permutation
    assert sorted(slots.reshape(-1).tolist()) == list(range(P))
    return slots


def test_allgather_unpermute_one_rank(gp):
    import torch
    capi, ctx = gp
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    P, row = 37, 3 * 25
    slots = _partition(capi, P, 5)
    comm = capi.Comm(ctx, 1, 0)
    assert "rccl" in capi.load().gpc_comm_library().decode().lower()
    comm.set_partition(P, slots)
    dev = torch.device("cuda:0")
    local = torch.randn((P, row), dtype=torch.float64, device=dev)          # row s = the grid of the patch in slot s
    gathered = torch.empty_like(local)
    f_star = torch.empty_like(local)
    comm.allgather_fstar_dev(row, local, gathered, f_star)
    ctx.synchronize()
    assert torch.equal(gathered, local)
    want = np.empty((P, row))
    want[slots.reshape(-1)] = local.cpu().numpy()
    assert np.array_equal(f_star.cpu().numpy(), want)
    # odd row length (the 8-byte path of the un-permutation kernel)
    row2 = 7
    loc2 = torch.randn((P, row2), dtype=torch.float64, device=dev)
    g2, f2 = torch.empty_like(loc2), torch.empty_like(loc2)
    comm.allgather_fstar_dev(row2, loc2, g2, f2)
    ctx.synchronize()
    w2 = np.empty((P, row2))
    w2[slots.reshape(-1)] = loc2.cpu().numpy()
    assert np.array_equal(f2.cpu().numpy(), w2)
    # a table that is not a partition is refused
    bad = slots.copy().reshape(-1)
    bad[1] = bad[0]
    with pytest.raises(capi.GpcError):
        comm.set_partition(P, bad)
    comm.close()
    ctx.set_stream(None)


def test_single_process_group_form(gp):
    """gpc_comm_create_all + gpc_group_start / gpc_group_end: how one host process drives the GPUs of a node (here: one)."""
    import torch
    capi, ctx = gp
    lib = capi.load()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    P, row = 19, 400
    slots = _partition(capi, P, 9)
    (comm,) = capi.Comm.all([ctx])
    assert lib.gpc_comm_world(comm.h) == 1 and lib.gpc_comm_rank(comm.h) == 0
    comm.set_partition(P, slots)
    dev = torch.device("cuda:0")
    local = torch.randn((P, row), dtype=torch.float64, device=dev)
    gathered, f_star = torch.empty_like(local), torch.empty_like(local)
    assert lib.gpc_group_start() == 0
    comm.allgather_fstar_dev(row, local, gathered, None)        # inside the bracket the collective is only recorded
    assert lib.gpc_group_end() == 0
    comm.unpermute_fstar_dev(row, gathered, f_star)
    ctx.synchronize()
    want = np.empty((P, row))
    want[slots.reshape(-1)] = local.cpu().numpy()
    assert np.array_equal(f_star.cpu().numpy(), want)
    comm.close()
    ctx.set_stream(None)


def test_sharded_dense_flow_through_the_c_abi(gp, oracle):
    """the whole N-rank recipe with world = 1, C-ABI calls only: partition -> this rank's slot batch -> fit + predict ->
    all-gather + un-permute -> the grids of the unsharded call, bit for bit (same kernel, same inputs per patch)"""
    import torch
    from gp_compressor_amd import dist as gdist
    capi, ctx = gp
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    P, res, sz = 48, 0.15, 20
    off, x0, x1, y = synth.make_patches(P, 200, res=res, seed=21, ragged=True)
    slots, loff, lx0, lx1, ly = gdist.shard_batch(off, x0, x1, y, 1, 0)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    prm = capi.default_params_dense()
    S, m = slots.shape[1], sz * sz
    d = [t(a) for a in (loff, lx0, lx1, ly)]
    local = torch.empty((S, m), dtype=torch.float64, device=dev)
    ctx.dense_fit_predict_grid_dev(prm, S, d[0], int(np.max(np.diff(loff))), int(loff[-1]), d[1], d[2], d[3], 1, res, sz, local)
    comm = capi.Comm(ctx, 1, 0)
    comm.set_partition(P, slots)
    gathered, f_star = torch.empty_like(local), torch.empty((P, m), dtype=torch.float64, device=dev)
    comm.allgather_fstar_dev(m, local, gathered, f_star)
    ctx.synchronize()
    f_ref, st = ctx.dense_fit_predict_grid(prm, off, x0, x1, y, res, sz)
    got = f_star.cpu().numpy()
    assert np.all(st == 0) and np.max(np.abs(got - f_ref[:, 0, :])) <= 1e-12 * np.max(np.abs(f_ref))
    xs0, xs1 = oracle.grid(res, sz)
    fo, _, _ = oracle.dense_fit_predict_batch(oracle.dense_params(), off, x0, x1, y, xs0, xs1)
    assert np.max(np.abs(got - fo[:, 0, :])) <= 1e-9 * np.max(np.abs(fo))
    comm.close()
    ctx.set_stream(None)
