"""One process per GPU: patch sharding and the single all-gather that reassembles the decompressed grids.

Patches are independent in the reference (each gps[i] touches only its own X_i, y_i and state,
/root/reference/src/gp_compressor.cpp:146-163), so the only exchange of the multi-GPU path is one fixed-size
all-gather of f_star over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).  The partition itself is the
C-ABI's gpc_partition_patches (longest-processing-time on n^3 / n b^2, ranks padded to ceil(P/world) slots).

bench.py (N > 1) and tests/test_dist_cpu.py drive the SAME code: global_batch -> shard_batch -> [the rank's fit +
predict] -> ShardedGather (one all_gather_into_tensor, un-permute to patch order, self-check).
"""
import numpy as np

from . import capi, synth


def global_batch(world, patches_per_rank, n, res=0.15, seed=2, ragged=False, n_min=None):
    """The job's whole patch batch as the concatenation of `world` rank-seeded shards (shard r: seed + 1000 r, the same
    buffers a 1-rank run of shard r would see).  Deterministic, so every rank can build it without an exchange.
    Returns off (P+1,), x0, x1 (N,), y (1, N) with P = world * patches_per_rank."""
    offs, xs0, xs1, ys = [], [], [], []
    base = 0
    for r in range(world):
        off, x0, x1, y = synth.make_patches(patches_per_rank, n, res=res, seed=seed + 1000 * r, ragged=ragged, n_min=n_min)
        offs.append(off[:-1].astype(np.int64) + base)
        base += int(off[-1])
        xs0.append(x0)
        xs1.append(x1)
        ys.append(y)
    off = np.concatenate(offs + [np.array([base], dtype=np.int64)]).astype(np.int32)
    return off, np.concatenate(xs0), np.concatenate(xs1), np.ascontiguousarray(np.concatenate(ys, axis=1))


def shard_batch(off, x0, x1, y, world, rank, sparse_capacity=0):
    """Returns (slots, loc_off, loc_x0, loc_x1, loc_y): this rank's patches as a CSR batch of S = ceil(P/world) slots
    (padding slots are empty patches, n = 0) plus the global slot table (world, S) needed to undo the permutation."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    slots = capi.partition_patches(off, world, sparse_capacity)
    mine = slots[rank]
    off64 = off.astype(np.int64)
    counts = np.where(mine >= 0, off64[np.maximum(mine, 0) + 1] - off64[np.maximum(mine, 0)], 0)
    loc_off = np.zeros(len(mine) + 1, dtype=np.int32)
    loc_off[1:] = np.cumsum(counts)
    # rows of the rank's patches, slot by slot (vectorised: start of each slot repeated over its rows + a running index)
    starts = np.repeat(off64[np.maximum(mine, 0)], counts)
    within = np.arange(int(loc_off[-1]), dtype=np.int64) - np.repeat(loc_off[:-1].astype(np.int64), counts)
    idx = starts + within
    y = np.atleast_2d(y)
    return slots, loc_off, np.ascontiguousarray(x0[idx]), np.ascontiguousarray(x1[idx]), np.ascontiguousarray(y[:, idx])


def slot_index(slots, P):
    """slot_of_patch (P,): row of the gathered (world * S) buffer that holds patch p"""
    sp = np.asarray(slots).reshape(-1)
    slot_of_patch = np.full(P, -1, dtype=np.int64)
    slot_of_patch[sp[sp >= 0]] = np.nonzero(sp >= 0)[0]
    assert np.all(slot_of_patch >= 0), "a patch has no slot"
    return slot_of_patch


class ShardedGather:
    """The exchange step of the multi-GPU path: ONE all_gather_into_tensor of the ranks' (S, ny, m) slot buffers into a
    (world * S, ny, m) buffer, then the un-permutation to patch order (a device gather).  start() may run asynchronously
    (RCCL's own stream) so that the next step's kernel overlaps it; finish() waits and un-permutes."""

    def __init__(self, slots, P, like, world):
        import torch
        self.torch = torch
        self.world, self.S = slots.shape
        assert self.world == world
        self.P = P
        self.index = torch.from_numpy(slot_index(slots, P)).to(like.device)
        shape = tuple(like.shape[1:])
        self.flat = torch.empty((world * self.S,) + shape, dtype=like.dtype, device=like.device)
        self.out = torch.empty((P,) + shape, dtype=like.dtype, device=like.device)
        self.work = None

    def start(self, local_f, async_op=True):
        import torch.distributed as dist
        assert local_f.shape[0] == self.S and local_f.is_contiguous()
        if dist.is_initialized():
            w = dist.all_gather_into_tensor(self.flat, local_f, async_op=async_op)   # rank r lands in rows [r*S, (r+1)*S)
            self.work = w if async_op else None
        else:
            self.flat.copy_(local_f)
            self.work = None

    def finish(self):
        """returns f_star in patch order, (P, ny, m), on the device"""
        if self.work is not None:
            self.work.wait()
            self.work = None
        self.torch.index_select(self.flat, 0, self.index, out=self.out)
        return self.out

    def own_rows_match(self, local_f, rank, slots):
        """self-check: the gathered rows of this rank's own slots are the buffer it contributed, bit for bit, and the
        un-permuted result holds them at its patches' positions"""
        torch = self.torch
        mine = np.asarray(slots[rank])
        ok = bool(torch.equal(self.flat[rank * self.S:(rank + 1) * self.S], local_f))
        live = np.nonzero(mine >= 0)[0]
        if len(live):
            pid = torch.from_numpy(mine[live].astype(np.int64)).to(local_f.device)
            ok = ok and bool(torch.equal(self.out[pid], local_f[torch.from_numpy(live).to(local_f.device)]))
        return ok


class CabiGather:
    """The same exchange through the C-ABI a reference-side binding would call (include/gpc.h "multi-GPU"): gpc_comm_create
    (RCCL communicator from a unique id that rank 0 draws and torch.distributed merely carries to the other ranks),
    gpc_comm_set_partition, then per step ONE gpc_allgather_fstar_dev = ncclAllGather + the un-permutation kernel.
    The collective runs on a context of its own, bound to a side stream, so that the next step's kernel overlaps it
    (start() orders it behind the producing kernel with an event; finish() makes the caller's stream wait for it).
    Interface of ShardedGather.  Raises GpcError when RCCL cannot be bound (the caller falls back to ShardedGather)."""

    def __init__(self, slots, P, like, world, rank, device_index, share=None):
        """share: another CabiGather of the same partition whose communicator, context and side stream this one uses (a second
        set of buffers for double-buffered steps; one RCCL communicator per rank is enough)"""
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.world, self.S = slots.shape
        assert self.world == world
        self.P, self.rank = P, rank
        self.row = int(np.prod(like.shape[1:]))
        self.owner = share is None
        if share is None:
            # Agree BEFORE the collective: every step that can fail locally -- the side stream, the context, binding
            # RCCL (drawing a unique id does that on every rank; only rank 0's is used) -- runs first, then one all_reduce(MIN) of an
            # `able` flag decides for all ranks alike.  Only then does anybody enter ncclCommInitRank, which blocks until every rank is in.
            uid, able, err = [None], 1, None
            self.side = self.cctx = None
            try:
                self.side = torch.cuda.Stream(device=like.device)
                self.cctx = capi.Context(device_index)
                self.cctx.set_stream(self.side.cuda_stream)
                mine = capi.Comm.unique_id()
                if rank == 0:
                    uid[0] = mine
            except Exception as e:
                able, err = 0, e
            if dist.is_initialized() and world > 1:
                okt = torch.tensor([able], dtype=torch.int32, device=like.device)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                able_all = int(okt.item())
            else:
                able_all = able
            if not able_all:
                if self.cctx is not None:
                    self.cctx.close()
                raise RuntimeError(f"the C-ABI communicator cannot be set up on every rank ({err if err else 'another rank failed'})")
            if dist.is_initialized() and world > 1:
                dist.broadcast_object_list(uid, src=0)       # the channel "the host has" for the 128 bytes
            self.comm = capi.Comm(self.cctx, world, rank, uid[0])
            self.comm.set_partition(P, slots)
        else:
            self.side, self.cctx, self.comm = share.side, share.cctx, share.comm
        shape = tuple(like.shape[1:])
        self.flat = torch.empty((world * self.S,) + shape, dtype=like.dtype, device=like.device)
        self.out = torch.empty((P,) + shape, dtype=like.dtype, device=like.device)
        self.ev_in = torch.cuda.Event()
        self.ev_out = torch.cuda.Event()
        self.pending = False

    def start(self, local_f, async_op=True):
        assert local_f.shape[0] == self.S and local_f.is_contiguous()
        cur = self.torch.cuda.current_stream()
        self.ev_in.record(cur)                               # the kernel that produced local_f
        self.side.wait_event(self.ev_in)
        self.comm.allgather_fstar_dev(self.row, local_f, self.flat, self.out)
        self.ev_out.record(self.side)
        self.pending = True
        if not async_op:
            self.finish()

    def finish(self):
        if self.pending:
            self.torch.cuda.current_stream().wait_event(self.ev_out)
            self.pending = False
        return self.out

    own_rows_match = ShardedGather.own_rows_match

    def close(self):
        if self.owner:
            self.comm.close()
            self.cctx.close()


def make_gather(slots, P, like, world, rank, device_index, prefer_cabi=True):
    """The exchange object of the N-rank path and a word on which one it is: the C-ABI's communicator when RCCL binds on every
    rank, torch.distributed's all_gather_into_tensor otherwise (all ranks take the same branch)."""
    import torch
    import torch.distributed as dist
    g, why = None, ""
    if prefer_cabi and like.is_cuda:
        try:
            g = CabiGather(slots, P, like, world, rank, device_index)      # (raises on every rank or on none: it agrees before the collective)
        except Exception as e:          # RCCL not found / context refused
            why = f"{type(e).__name__}: {e}"
    if g is not None:
        return g, "C-ABI: gpc_comm_create + gpc_allgather_fstar_dev (ncclAllGather + un-permute kernel, " + capi.load().gpc_comm_library().decode() + ")"
    return ShardedGather(slots, P, like, world), "torch.distributed.all_gather_into_tensor + index_select" + (f" (C-ABI communicator unavailable: {why})" if why else "")


def gather_fstar(local_f, slots, P):
    """local_f: torch tensor (S, ny, m) of this rank's slots.  One all_gather_into_tensor, then un-permute to patch
    order: returns (P, ny, m) on the same device."""
    g = ShardedGather(np.asarray(slots), P, local_f, slots.shape[0])
    g.start(local_f.contiguous(), async_op=False)
    return g.finish().clone()
