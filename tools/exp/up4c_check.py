"""compact + up4 path vs the pre_add path of the x4 level at the config-4 shape (bit-identical by construction). GPU box: python tools/exp/up4c_check.py [B h Cout]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
B, h, Cout = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 96, 1024)
g = torch.Generator().manual_seed(0)
taps = torch.randn(B, 256, h, h, generator=g).cuda()
key = torch.randn(B, 64, 4 * h, 4 * h, generator=g).cuda()
wt = torch.randn(Cout, 256, 3, 3, generator=g) * 0.02
wk = torch.randn(Cout, 64, 3, 3, generator=g) * 0.04
th = ops.h2_pack(taps, ops.h2_empty(B, 256, h, h))
kh = ops.h2_pack(key, ops.h2_empty(B, 64, 4 * h, 4 * h))
pt, pk = ops.pack_conv_up4_h2t(wt), ops.pack_conv_x3(wk, 1, lazy=True)
a = ops.conv_h2x(kh, pk, ops.empty(B, Cout, 4 * h, 4 * h), y_fmt=1)
ops.conv_up4_h2t(th, pt, a, pre_add=a)
comp = ops.conv_up4_h2t(th, pt, ops.empty(B, 9 * Cout, h, h), compact=True)
b = ops.conv_h2x(kh, pk, ops.empty(B, Cout, 4 * h, 4 * h), y_fmt=1, up4=comp)
torch.cuda.synchronize()
d = (a - b).abs()
print("max diff %.3e, differing elements %d of %d" % (float(d.max()), int((d > 0).sum()), d.numel()))
if float(d.max()) > 0:
    q = d.view(B, Cout // 4, 4 * h, 4 * h, 4)
    idx = (q > 0).nonzero()
    print("first differing (b, quad, y, x, c):", idx[:5].tolist(), " last:", idx[-3:].tolist())
    print("quads with differences:", sorted(set(idx[:, 1].tolist()))[:20], "...", "rows mod 4:", sorted(set((idx[:, 2] % 4).tolist())), "cols mod 4:", sorted(set((idx[:, 3] % 4).tolist())))
for n in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv_h2x(kh, pk, a, y_fmt=1); ops.conv_up4_h2t(th, pt, a, pre_add=a); e1.record(); torch.cuda.synchronize(); t0 = e0.elapsed_time(e1)
    e0.record(); ops.conv_up4_h2t(th, pt, comp, compact=True); e1.record(); torch.cuda.synchronize(); t1 = e0.elapsed_time(e1)
    e0.record(); ops.conv_h2x(kh, pk, b, y_fmt=1, up4=comp); e1.record(); torch.cuda.synchronize(); t2 = e0.elapsed_time(e1)
    e0.record(); ops.conv_h2x(kh, pk, b, y_fmt=1); e1.record(); torch.cuda.synchronize(); t3 = e0.elapsed_time(e1)
    print("pre_add path %.2f ms | compact taps %.2f + key conv with up4 %.2f = %.2f ms | key conv alone %.2f" % (t0, t1, t2, t1 + t2, t3))
