"""-m gpu: the RCCL code path of bfsr_amd.dist on the hardware that exists (one GPU): `init(force=True)` creates a real
`nccl` (= RCCL) process group of world size 1, and the double-buffered AsyncGatherer runs its `all_gather_into_tensor(async_op=True)`
/ `work.wait()` protocol through it while a compute kernel is in flight on the main stream.  (A 2-rank launch on one GPU is refused
by RCCL: "Duplicate GPU detected"; the world-2 logic is covered on gloo in tests/test_dist_gloo.py.)  Runs in a child process so
the process group never leaks into the other GPU tests."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time
import torch
sys.path.insert(0, os.environ["BFSR_ROOT"])
from bfsr_amd import dist as bdist
import torch.distributed as dist

rank, world, local = bdist.init(force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1, (dist.get_backend(), world)
dev = torch.device("cuda", 0)
total = 8
gat = bdist.AsyncGatherer(total)
outs = [torch.randn(total, 3, 640, 640, device=dev) for _ in range(3)]
a = torch.randn(4096, 4096, device=dev)
torch.cuda.synchronize()
prevs, ptrs = [], []
host_gap = []
for s, o in enumerate(outs):
    ev = torch.cuda.Event()
    for _ in range(8):
        b = a @ a                                   # compute of "step s+1" queued on the main stream
    ev.record()
    t0 = time.perf_counter()
    prev = gat.submit(o)                            # must not wait for the matmuls on the host
    host_gap.append((time.perf_counter() - t0, ev.query()))
    ptrs.append(None if prev is None else prev.data_ptr())
    prevs.append(None if prev is None else prev.clone())      # the returned buffer is only valid until the second following submit
last = gat.finish()
torch.cuda.synchronize()
assert prevs[0] is None
assert torch.equal(prevs[1], outs[0]) and torch.equal(prevs[2], outs[1]), "gathered tensor != local tensor"
assert torch.equal(last, outs[2])
assert ptrs[1] != ptrs[2], "double buffer"
# no full sync inside submit: at least one submit returned while the matmuls queued before it were still running
assert any(not done for _, done in host_gap), host_gap
g = bdist.all_gather_batch(outs[0][:5], total=5)    # synchronous path through RCCL as well
assert torch.equal(g, outs[0][:5])
bdist.barrier()
dist.destroy_process_group()
print("RCCL_OK", [round(t * 1e3, 3) for t, _ in host_gap])
'''


def test_async_gatherer_through_rccl_world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BFSR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
