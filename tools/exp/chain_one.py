"""Chain-of-one timing of the five dense-block convs on conv_chain_kernel (and conv_h2x beside it): us, TFLOP/s-equivalent, us per LDS stage and workgroup.
GPU box: python tools/exp/chain_one.py B H [reps]    (BFSR_HIP_LIB selects an ablation build: results are then wrong by construction)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
B, H = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D = ops.h2_pack(torch.randn(B, 192, H, H, device="cuda") * 0.5, ops.h2_empty(B, 192, H, H))
O = ops.h2_empty(B, 64, H, H)
def timed(f, n):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tiles = B * ((H + 31) // 32) ** 2
abl = "BFSR_HIP_LIB" in os.environ
for ci, co in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
    pw = ops.pack_conv_x3(torch.randn(co, ci, 3, 3, generator=g) * 0.01, 1, lazy=True)
    epi = ops.pack_epilogue(co, bias=torch.zeros(co))
    one = dict(x=D[:, :ci // 8], pw=pw, out=O[:, :co // 8], epi=epi, act=2, slope=0.2)
    ch = ops.conv_chain([one])
    t2 = timed(lambda: ch.run(), reps)
    f = 2 * 9 * ci * co * B * H * H
    items = tiles * (co // 32)
    stages_per_wg = -(-items // 256) * (ci // 8)
    line = "  %3d->%2d @ %dx%d^2: chain-of-one %7.1f us (%4.0f TFLOP/s-eq, %.2f of 833; %d items, %.2f us per stage at the busiest workgroup)" % (
        ci, co, B, H, t2 * 1e3, f / t2 / 1e9, f / t2 / 1e9 / 833, items, t2 * 1e3 / stages_per_wg)
    if not abl:
        t1 = timed(lambda: ops.conv_h2x(one["x"], one["pw"], one["out"], epi=epi, act=2, slope=0.2), reps)
        line += "   conv_h2x %7.1f us (%4.0f)" % (t1 * 1e3, f / t1 / 1e9)
    print(line, flush=True)
ops.range_flag.zero_()
