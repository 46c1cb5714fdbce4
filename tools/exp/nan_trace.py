"""Find the first op whose output holds a NaN in an 8x encode at config-4 size (debug aid).  GPU box: python tools/exp/nan_trace.py"""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from bfsr_amd import synth
from test_srflow_gpu import build
from bfsr_amd.ops import HipOps, MODE_BILINEAR
hip = HipOps("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 96
m, prior, opt, sd, psd = build(hip, S)
eng = m.netG.module.engine()
seen = [0]
def wrap(name):
    f = getattr(hip, name)
    def g(*a, **k):
        out = f(*a, **k)
        o = out if torch.is_tensor(out) else (a[2] if len(a) > 2 and torch.is_tensor(a[2]) else None)
        if o is not None and seen[0] < 6:
            bad = int(torch.isnan(o.float()).sum())
            if bad:
                seen[0] += 1
                ins = [tuple(t.shape) for t in a if torch.is_tensor(t)]
                print("NaN after", name, "out", tuple(o.shape), o.dtype, "count", bad, "inputs", ins, {k_: (tuple(v.shape) if torch.is_tensor(v) else v) for k_, v in k.items()}, flush=True)
                idx = torch.isnan(o.float()).nonzero()
                print("   first", idx[0].tolist(), "last", idx[-1].tolist(), flush=True)
                b_, _, y_, x_ = idx[0].tolist()[:4] if o.dim() == 4 else (0, 0, 0, 0)
                for k_, v in list(k.items()) + [("arg%d" % i_, t_) for i_, t_ in enumerate(a)]:
                    if torch.is_tensor(v) and v.dim() >= 4 and v is not o:
                        if v.dim() == 6:
                            print("   ", k_, "h2 at pixel hi", v[b_, :, 0, y_, x_].float().flatten().tolist()[:64], "\n       lo", v[b_, :, 1, y_, x_].float().flatten().tolist()[:16])
                        elif k.get("h_ft_fmt") and k_ == "h_ft":
                            q = v.reshape(v.shape[0], v.shape[1] // 4, v.shape[2], v.shape[3], 4)
                            print("   ", k_, "q4 at pixel", q[b_, :, y_, x_].flatten().tolist(), "nan total", int(torch.isnan(v).sum()), "absmax", float(torch.nan_to_num(v).abs().max()))
                        else:
                            print("   ", k_, "at pixel", v[b_, :, y_, x_].flatten().tolist()[:32], "absmax", float(torch.nan_to_num(v.float()).abs().max()))
                for t in a:
                    if torch.is_tensor(t) and t is not o and t.dtype in (torch.float32, torch.float16):
                        print("   input nan count", int(torch.isnan(t.float()).sum()), tuple(t.shape))
        return out
    setattr(hip, name, g)
for n in ("conv", "conv_f16", "conv_x3", "conv_x3s", "conv_h2x", "conv_h2r", "conv_up2", "conv_up2_x3", "h2_pack", "h2_unpack", "coupling_head", "coupling_tail", "conv1x1", "resize"):
    if hasattr(hip, n):
        wrap(n)
lr = hip.to_device(synth.smooth_lr_batch(61, 8, L, L))
lr_up = hip.resize(lr, hip.empty(8, 3, S * L, S * L), MODE_BILINEAR, 1.0 / S, 1.0 / S)
ep = eng.encode(lr_up, lr)
print("encode nan:", [int(torch.isnan(e).sum()) for e in ep])
rt = eng.decode(lr, epses=[e.clone() for e in ep])
print("decode nan:", int(torch.isnan(rt).sum()), "err", float((rt - lr_up).abs().max()))
