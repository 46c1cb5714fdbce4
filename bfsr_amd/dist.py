"""Data-parallel sharding of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on
ROCm, over xGMI inside a node).  Every op on the path is per-sample (eval-mode BN/ActNorm are fixed
per-channel affines, eps standardisation is per pixel), so the batch dimension shards exactly: rank r
takes `lr[r*B/N:(r+1)*B/N]`, weights are replicated (each rank builds them itself) and the only exchange
is one all-gather of the `[B/N,3,H,W]` outputs (reference: nn.DataParallel scatter/gather,
SRFlow-LP/code/models/SRFlow_model.py:53)."""
import os

import torch
import torch.distributed as dist


def init(backend=None, force=None):
    """Initialise from the torchrun environment; returns (rank, world_size, local_rank).  A single process does not need a process
    group and gets none -- unless `force` (or BFSR_DIST_FORCE=1) asks for one: then even world_size 1 goes through
    `init_process_group` and every gather below runs the real collective (RCCL on a GPU box), which is how the nccl code path is
    exercised on the 1-GPU boxes of the test pool (tests/test_dist_rccl.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("BFSR_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("BFSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n, rank, world):
    """Contiguous near-equal split of n units (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(batch, rank, world):
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return batch[lo:hi]


def all_gather_batch(local_out, total=None):
    """All-gather per-rank outputs along dim 0.  Equal shards use one all_gather_into_tensor (a single
    RCCL collective); ragged shards are padded to the largest shard."""
    if not dist.is_initialized():
        return local_out
    world = dist.get_world_size()
    n_local = torch.tensor([local_out.shape[0]], device=local_out.device, dtype=torch.long)
    if total is not None and total % world == 0:
        out = local_out.new_empty((total,) + tuple(local_out.shape[1:]))
        dist.all_gather_into_tensor(out, local_out.contiguous())
        return out
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = local_out.new_zeros((m,) + tuple(local_out.shape[1:]))
    pad[:local_out.shape[0]] = local_out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


class AsyncGatherer(object):
    """Double-buffered all-gather of the per-step outputs.

    `submit(out_i)` starts the collective for step i without blocking the compute stream, so the gather of step i runs over xGMI
    while step i+1 computes.  It first completes the gather of step i-1 and RETURNS it (None on the first call); `finish()`
    completes whatever is in flight and returns the last gathered batch.  `.last` is always the most recently COMPLETED gather,
    i.e. after `submit(i)` it is step i-1 on every path (equal and ragged shards alike).

    Buffer lifetime: the tensor returned for step i is one of two internal buffers and is overwritten by `submit(i+2)`; clone it
    to keep it longer.  The local output is kept referenced until its gather has completed, so the caching allocator cannot
    recycle it early.  The collective is RCCL's all_gather_into_tensor on its own internal stream (`async_op=True`);
    `work.wait()` orders the CURRENT stream after it without a host sync.  Without a process group it degenerates to handing
    the local output back one step late (the same contract)."""

    def __init__(self, total):
        self.total, self.bufs, self.pending, self.n, self.last = total, [None, None], None, 0, None

    def submit(self, local_out):
        prev = self._complete()
        if not dist.is_initialized():
            self.pending = (None, local_out, None)
            return prev
        world = dist.get_world_size()
        if self.total % world != 0 or local_out.shape[0] * world != self.total:
            # ragged shards: synchronous padded gather, surfaced one step late like the asynchronous path
            self.pending = (None, all_gather_batch(local_out, total=self.total), None)
            return prev
        i = self.n & 1
        shape = (self.total,) + tuple(local_out.shape[1:])
        if self.bufs[i] is None or tuple(self.bufs[i].shape) != shape or self.bufs[i].device != local_out.device:
            self.bufs[i] = local_out.new_empty(shape)
        src = local_out.contiguous()
        work = dist.all_gather_into_tensor(self.bufs[i], src, async_op=True)
        self.pending = (work, self.bufs[i], src)
        self.n += 1
        return prev

    def _complete(self):
        if self.pending is not None:
            work, buf, _src = self.pending
            if work is not None:
                work.wait()                 # orders the current stream after the collective
            self.last, self.pending = buf, None
        return self.last

    def finish(self):
        return self._complete()


def barrier():
    if dist.is_initialized():
        dist.barrier()
