import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools", "exp", "libdbg.so")
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
torch.manual_seed(0)
B, Cin, Cout, H, W = 1, 16, 32, 8, 32
x = torch.randn(B, Cin, H, W)
w = torch.randn(Cout, Cin, 3, 3) * 0.05
pw = ops.pack_conv_x3(w, 1)
x3 = ops.x3_pack(x.cuda(), ops.x3_empty(B, Cin, H, W))
out = torch.zeros(B, Cout, 64, 64, device="cuda")     # big enough for the dump (64512 B)
a = _lib.BfsrConvX3Args()
a.x, a.x_bs, a.Cin = x3.data_ptr(), x3.stride(0), Cin
a.w = pw.data.data_ptr()
a.y, a.y_bs, a.Cout, a.y_fmt = out.data_ptr(), Cout * H * W, Cout, 0
a.B, a.H, a.W, a.tune = B, H, W, -7
err = ops.lib.bfsr_conv3x3_x3s(C.byref(a), None)
torch.cuda.synchronize()
print("rc", err)
dump = out.view(-1).view(torch.int16)[: 64512 // 2].cpu()
wl = dump[36864 // 2:]
wg = pw.data.cpu().view(torch.int16)[: 27648 // 2]
print("weights equal:", torch.equal(wl, wg), "n diff", int((wl != wg).sum()), "of", wl.numel())
if not torch.equal(wl, wg):
    d = (wl != wg).view(27, 512)      # per 1024-B piece
    print("pieces with diffs:", d.any(dim=1).nonzero().view(-1).tolist())
    print("lds piece0 first 32:", wl[:32].tolist()); print("glb piece0 first 32:", wg[:32].tolist())
    print("lds piece1 first 16:", wl[512:528].tolist()); print("glb piece1 first 16:", wg[512:528].tolist())
# input check: sub-image s (plane pl, khalf kh): position p -> 8 bf16
xin = dump[: 36864 // 2].view(6, 384, 8)
x3c = x3.cpu().view(torch.int16)     # [1, 2, 3, 8, 32, 8]
ok = True
for s_ in range(6):
    pl, kh = s_ >> 1, s_ & 1
    for r in range(10):
        for c in range(34):
            gy, gx = r - 1, c - 1
            exp = x3c[0, kh, pl, gy, gx] if (0 <= gy < H and 0 <= gx < W) else torch.zeros(8, dtype=torch.int16)
            if not torch.equal(xin[s_, r * 34 + c], exp):
                ok = False
print("input tile equal:", ok, "padding zero:", bool((xin[:, 340:] == 0).all()))

a.tune = -8
out.zero_()
err = ops.lib.bfsr_conv3x3_x3s(C.byref(a), None)
torch.cuda.synchronize()
fr = out.view(-1).view(torch.int16)[: 512 * 16].cpu().view(512, 2, 8)
wg3 = wg.view(3, 9, 2, 32, 8)
bad = 0
for t in range(512):
    lane = t & 63
    exp = wg3[0, 0, lane >> 5, lane & 31]
    if not torch.equal(fr[t, 0], exp):
        bad += 1
        if bad < 6:
            print("A frag tid", t, "got", fr[t, 0].tolist(), "exp", exp.tolist())
print("A frags bad:", bad)
bad = 0
for t in range(512):
    lane, wave = t & 63, t >> 6
    exp = xin[0 * 2 + (lane >> 5), wave * 34 + (lane & 31)]
    if not torch.equal(fr[t, 1], exp):
        bad += 1
        if bad < 4:
            print("B frag tid", t, "got", fr[t, 1].tolist(), "exp", exp.tolist())
print("B frags bad:", bad)

import torch.nn.functional as F
a.tune = -9
out.zero_()
err = ops.lib.bfsr_conv3x3_x3s(C.byref(a), None)
torch.cuda.synchronize()
acc = out.view(-1)[: 512 * 16].cpu().view(8, 2, 32, 16)     # [wave][lhi][l31][r]
ref = F.conv2d(x, w, None, padding=1)[0]                     # [32, 8, 32]
bad = 0
for r in range(16):
    for lhi in range(2):
        co = (r & 3) + 8 * (r >> 2) + 4 * lhi
        d = (acc[:, lhi, :, r] - ref[co]).abs().max().item()
        if d > 1e-4:
            bad += 1
            print("acc r", r, "lhi", lhi, "co", co, "err", d, "vs ch0:", (acc[:, lhi, :, r] - ref[0]).abs().max().item())
print("acc mismatches:", bad)
