"""One process per GPU: patch sharding and the single all-gather that reassembles the decompressed grids.

Patches are independent in the reference (each gps[i] touches only its own X_i, y_i and state,
/root/reference/src/gp_compressor.cpp:146-163), so the only exchange of the multi-GPU path is one fixed-size
all-gather of f_star over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).  The partition itself is the
C-ABI's gpc_partition_patches (longest-processing-time on n^3 / n b^2, ranks padded to ceil(P/world) slots).
"""
import numpy as np

from . import capi


def shard_batch(off, x0, x1, y, world, rank, sparse_capacity=0):
    """Returns (slots, loc_off, loc_x0, loc_x1, loc_y): this rank's patches as a CSR batch of S = ceil(P/world) slots
    (padding slots are empty patches, n = 0) plus the global slot table (world, S) needed to undo the permutation."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    slots = capi.partition_patches(off, world, sparse_capacity)
    mine = slots[rank]
    counts = np.array([off[p + 1] - off[p] if p >= 0 else 0 for p in mine], dtype=np.int64)
    loc_off = np.zeros(len(mine) + 1, dtype=np.int32)
    loc_off[1:] = np.cumsum(counts)
    idx = np.concatenate([np.arange(off[p], off[p + 1]) for p in mine if p >= 0] or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
    y = np.atleast_2d(y)
    return slots, loc_off, np.ascontiguousarray(x0[idx]), np.ascontiguousarray(x1[idx]), np.ascontiguousarray(y[:, idx])


def gather_fstar(local_f, slots, P):
    """local_f: torch tensor (S, ny, m) of this rank's slots.  One all_gather_into_tensor, then un-permute to patch
    order: returns (P, ny, m) on the same device."""
    import torch
    import torch.distributed as dist
    world, S = slots.shape
    assert local_f.shape[0] == S
    flat = torch.empty((world * S,) + tuple(local_f.shape[1:]), dtype=local_f.dtype, device=local_f.device)
    if world > 1:
        dist.all_gather_into_tensor(flat, local_f.contiguous())   # rank r lands in rows [r*S, (r+1)*S)
    else:
        flat.copy_(local_f)
    slot_of_patch = np.empty(P, dtype=np.int64)
    sp = slots.reshape(-1)
    slot_of_patch[sp[sp >= 0]] = np.nonzero(sp >= 0)[0]
    return flat[torch.from_numpy(slot_of_patch).to(flat.device)]
