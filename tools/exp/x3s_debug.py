import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
torch.manual_seed(0)
B, Cin, Cout, H, W = 1, 16, 32, 8, 32
x = torch.randn(B, Cin, H, W)
w = torch.randn(Cout, Cin, 3, 3) * 0.05
ref = F.conv2d(x, w, None, padding=1)
pw = ops.pack_conv_x3(w, 1)
x3 = ops.x3_pack(x.cuda(), ops.x3_empty(B, Cin, H, W))
y = ops.conv_x3s(x3, pw, ops.empty(B, Cout, H, W)).cpu()
err = (y - ref).abs()
print("err per channel:", [round(v, 3) for v in err.amax(dim=(0, 2, 3)).tolist()])
print("err per row:", [round(v, 3) for v in err.amax(dim=(0, 1, 3)).tolist()])
print("err per col:", [round(v, 3) for v in err.amax(dim=(0, 1, 2)).tolist()])
# does y match ref under a channel permutation?
for c in range(Cout):
    d = (y[0, c][None] - ref[0]).abs().amax(dim=(1, 2))
    j = int(d.argmin())
    print("out ch %2d best matches ref ch %2d (err %.3e)" % (c, j, d[j]))
# interior only
print("interior err:", err[..., 2:-2, 2:-2].max().item())
# identity-ish test: only centre tap, w = I
w2 = torch.zeros(Cout, Cin, 3, 3)
for c in range(16):
    w2[c, c, 1, 1] = 1.0
    w2[c + 16, c, 1, 1] = 2.0
pw2 = ops.pack_conv_x3(w2, 1)
y2 = ops.conv_x3s(x3, pw2, ops.empty(B, Cout, H, W)).cpu()
ref2 = F.conv2d(x, w2, None, padding=1)
print("centre-tap identity err:", (y2 - ref2).abs().max().item())
if (y2 - ref2).abs().max() > 1e-3:
    print("y2[0,0,:2,:8]", y2[0, 0, :2, :8]); print("x[0,0,:2,:8]", x[0, 0, :2, :8])
    print("y2[0,1,:2,:8]", y2[0, 1, :2, :8]); print("x[0,1,:2,:8]", x[0, 1, :2, :8])
