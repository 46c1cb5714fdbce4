"""Is the soffset of a raw buffer store part of the range check?  conv_h2x with fp32 NCHW output, Cout = 40 (the second 32-channel group is
partial) into a channel slice of a wider zero buffer: anything written behind channel 40 shows.  GPU box: python tools/exp/soffset_bounds.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
for name in ("conv_h2x", "conv_h2s"):
    B, Cin, Cout, H, W = 2, 64, 40, 20, 36
    x = ops.h2_pack(torch.randn(B, Cin, H, W, device="cuda"), ops.h2_empty(B, Cin, H, W))
    wide = ops.zeros(B, Cout + 64, H, W)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    if name == "conv_h2x":
        ops.conv_h2x(x, ops.pack_conv_x3(w, 1, lazy=True), wide[:, 8:8 + Cout], epi=ops.pack_epilogue(Cout, bias=torch.ones(Cout)))
    else:
        ops.conv_h2s(x, ops.pack_conv_h2s(w), wide[:, 8:8 + Cout], epi=ops.pack_epilogue(Cout, bias=torch.ones(Cout)))
    torch.cuda.synchronize()
    print(name, "written inside:", int((wide[:, 8:8 + Cout] != 0).sum()), "of", wide[:, 8:8 + Cout].numel(),
          "| before the slice:", int((wide[:, :8] != 0).sum()), "| behind the slice:", int((wide[:, 8 + Cout:] != 0).sum()))
