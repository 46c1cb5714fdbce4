"""conv_h2x against the native fp32 kernel at engine shapes (debug aid).  Usage (GPU box): python tools/exp/h2x_check.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(3)
r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
for (B, H, W) in ((8, 96, 96), (8, 48, 48), (3, 96, 96), (8, 160, 160), (2, 192, 192)):
    D = ops.to_device(r(B, 192, H, W)); xr = ops.to_device(r(B, 64, H, W))
    Dh = ops.h2_pack(D, ops.h2_empty(B, 192, H, W)); xrh = ops.h2_pack(xr, ops.h2_empty(B, 64, H, W))
    D22 = ops.h2_unpack(Dh, ops.empty(B, 192, H, W)); x22 = ops.h2_unpack(xrh, ops.empty(B, 64, H, W))
    for Cin in (64, 96, 128, 160, 192):
        Cout = 64 if Cin == 192 else 32
        w, b = r(Cout, Cin, 3, 3, scale=0.03), r(Cout, scale=0.1)
        epi = ops.pack_epilogue(Cout, bias=b)
        kw = dict(act=2, slope=0.2) if Cin < 192 else dict(res1=None, alpha1=0.2, res2=None, alpha2=0.2)
        ref = ops.empty(B, Cout, H, W)
        if Cin < 192:
            ops.conv(D22[:, :Cin], ops.pack_conv(w, 1), ref, epi=epi, **kw)
        else:
            ops.conv(D22, ops.pack_conv(w, 1), ref, epi=epi, res1=D22[:, :64], alpha1=0.2, res2=x22, alpha2=0.2)
        for fmt in ("h2", "f32", "q4"):
            pw = ops.pack_conv_x3(w, 1)
            if fmt == "h2":
                nxt = ops.h2_empty(B, 192, H, W); nxt.fill_(float("nan"))
                out = nxt[:, Cin // 8: Cin // 8 + Cout // 8] if Cin < 192 else nxt[:, :8]
            else:
                out = ops.empty(B, Cout, H, W); out.fill_(float("nan"))
            if Cin < 192:
                ops.conv_h2x(Dh[:, :Cin // 8], pw, out, epi=epi, y_fmt=int(fmt == "q4"), **kw)
            else:
                ops.conv_h2x(Dh, pw, out, epi=epi, res1=Dh[:, :8], alpha1=0.2, res2=xrh, alpha2=0.2, y_fmt=int(fmt == "q4"))
            if fmt == "h2":
                got = ops.h2_unpack(out, ops.empty(B, Cout, H, W))
            elif fmt == "q4":
                got = out.view(B, Cout // 4, H, W, 4).permute(0, 1, 4, 2, 3).reshape(B, Cout, H, W)
            else:
                got = out
            d = (got - ref).abs()
            nn = int(torch.isnan(got).sum())
            print("B=%d %dx%d Cin=%d %s: max err %.3e nan %d" % (B, H, W, Cin, fmt, float(torch.nan_to_num(d).max()), nn), flush=True)
            if nn:
                idx = torch.isnan(got).nonzero()
                print("   first nan at", idx[0].tolist(), "last", idx[-1].tolist(), "channels", sorted(set(idx[:, 1].tolist()))[:10], "rows", sorted(set(idx[:, 2].tolist()))[:12])
