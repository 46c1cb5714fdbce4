"""Host enqueue cost of the launch path: N conv_h2x calls on a small tensor without synchronisation (Python + ctypes time per launch), and the split-batch RRDB
chain's host time against its GPU time.  GPU box: python tools/exp/host_rate.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
x = ops.h2_pack(torch.randn(1, 64, 16, 32, device="cuda"), ops.h2_empty(1, 64, 16, 32))
y = ops.h2_empty(1, 32, 16, 32)
pw = ops.pack_conv_x3(torch.randn(32, 64, 3, 3, generator=g) * 0.03, 1, lazy=True)
epi = ops.pack_epilogue(32, bias=torch.zeros(32))
for _ in range(10): ops.conv_h2x(x, pw, y, epi=epi, act=2)
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N): ops.conv_h2x(x, pw, y, epi=epi, act=2)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("conv_h2x: %.1f us of host time per launch (%d launches; GPU drained %.1f ms later)" % ((t1 - t0) / N * 1e6, N, (t2 - t1) * 1e3))
z = torch.randn(1, 12, 16, 32, device="cuda")
t0 = time.perf_counter()
for _ in range(N): ops.flow_pointwise(z, z, False)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("flow_pointwise: %.1f us of host time per launch" % ((t1 - t0) / N * 1e6))
