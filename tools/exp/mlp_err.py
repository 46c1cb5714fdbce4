import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bfsr_amd.ops import HipOps
from cpu_ops import CpuOps
import oracle.linf_ref as O
hip, CPU = HipOps("cuda:0"), CpuOps()
def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed)); return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))
for (h, w), (qh, qw) in (((8, 12), (21, 33)), ((16, 16), (22, 22)), ((5, 7), (30, 9)), ((24, 20), (65, 55))):
    B, HD, Cout = 2, 256, 540
    cf = rnd(1, B, 2 * HD, h, w); phase = rnd(2, HD // 2, 2, scale=0.5)
    ws = [rnd(11, HD, 4 * HD, scale=1.0 / 32), rnd(12, HD, HD, scale=1.0 / 16), rnd(13, HD, HD, scale=1.0 / 16), rnd(14, Cout, HD, scale=1.0 / 16)]
    bs = [rnd(15, HD, scale=0.1), rnd(16, HD, scale=0.1), rnd(17, HD, scale=0.1), rnd(18, Cout, scale=0.1)]
    torch.manual_seed(0)
    prep = O.batch_prep(torch.rand(B, 3, h, w), (qh * 3 - 1, qw * 3 - 2))
    coord, cell = prep["coord"], prep["cell"]
    for x3 in (True, False):
        ref = CPU.linf_mlp(cf, coord, cell, phase.reshape(-1), CPU.pack_linf_mlp(ws, bs, x3=x3), torch.empty(B, Cout, qh, qw), HD, x3=x3)
        outs = [hip.linf_mlp(hip.to_device(cf), hip.to_device(coord), hip.to_device(cell), hip.vec(phase), hip.pack_linf_mlp(ws, bs, x3=x3), hip.empty(B, Cout, qh, qw), HD, x3=x3).cpu().clone() for _ in range(3)]
        print((h, w), (qh, qw), "x3=%s" % x3, "err vs CPU %.3e" % float((outs[0] - ref).abs().max()), "run-to-run equal:", all(torch.equal(outs[0], o) for o in outs[1:]), flush=True)
