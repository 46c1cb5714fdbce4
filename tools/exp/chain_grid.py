"""Does a power-limited persistent kernel need every CU?  The fused dense-block chain with its grid capped at 256, 224, 192, 160, 128 workgroups
(interleaved rounds, sustained).  GPU box: python tools/exp/chain_grid.py B H NB"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, NB = (int(v) for v in sys.argv[1:4])
shapes = ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64))
ring = [ops.h2_pack(torch.randn(B, 192, H, H, device="cuda") * 0.5 if i == 0 else torch.zeros(B, 192, H, H, device="cuda"), ops.h2_empty(B, 192, H, H)) for i in range(4)]
allw = [[ops.pack_conv_x3(torch.randn(co, ci, 3, 3, generator=g) * (0.05 / (ci * 9) ** 0.5), 1, lazy=True) for ci, co in shapes] for _ in range(NB)]
epis = [ops.pack_epilogue(co, bias=torch.zeros(co)) for ci, co in shapes]
sp, cur = [], 0
for r in range(NB):
    D, Dn = ring[cur], ring[(cur + 1) % 4]
    for i, (ci, co) in enumerate(shapes[:4]):
        sp.append(dict(x=D[:, :ci // 8], pw=allw[r][i], out=D[:, ci // 8: ci // 8 + 4], epi=epis[i], act=2, slope=0.2))
    sp.append(dict(x=D, pw=allw[r][4], out=Dn[:, :8], epi=epis[4], res1=D[:, :8], alpha1=0.2))
    cur = (cur + 1) % 4
ch = ops.conv_chain(sp)
grids = (256, 240, 224, 208, 192, 160, 128)
times = {gd: [] for gd in grids}
for gd in grids: ch.run(tune=gd)
torch.cuda.synchronize()
for rnd in range(5):
    for gd in grids:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ch.run(tune=gd); e1.record(); torch.cuda.synchronize()
        times[gd].append(e0.elapsed_time(e1))
ops.check_range()
print("B=%d %dx%d, %d dense blocks: us per dense block by grid size (median of 5 interleaved rounds)" % (B, H, H, NB))
for gd in grids:
    m = statistics.median(times[gd])
    print("  grid %3d: %8.1f us   (x%.3f of 256)" % (gd, m / NB * 1e3, m / statistics.median(times[256])))
