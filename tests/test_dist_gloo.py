"""not gpu: the N>1 path (shard -> per-rank pipeline -> all-gather) with world_size 2 on gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from bfsr_amd import dist as bdist
    r, w, _ = bdist.init(backend="gloo")
    full = torch.arange(total * 3 * 2 * 2, dtype=torch.float32).view(total, 3, 2, 2)
    mine = bdist.shard(full, r, w)
    out = mine * 2 + 1                                   # stands in for the per-sample pipeline
    gathered = bdist.all_gather_batch(out, total=total)
    ok = torch.equal(gathered, full * 2 + 1)
    bdist.barrier()
    q.put((rank, ok, tuple(mine.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
def test_shard_allgather_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert sum(s[0] for _, _, s in res) == total


def test_shard_bounds_cover_everything():
    from bfsr_amd.dist import shard_bounds
    for n in (1, 7, 8, 64, 129):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
