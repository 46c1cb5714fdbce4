import sys, torch
sys.path.insert(0, "/root/repo")
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
for C, hw in ((12, 320), (24, 160), (96, 80)):
    B = 8
    z = torch.randn(B, C, hw, hw, device="cuda")
    ha = torch.randn(B, 2 * (C - C // 2), hw, hw, device="cuda") * 0.1
    hf = torch.randn(B, 2 * C, hw, hw, device="cuda") * 0.1
    w = torch.linalg.qr(torch.randn(C, C))[0].contiguous().cuda()
    wt = w.t().contiguous()
    ab, ae = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    for rev in (True, False):
        f = lambda: ops.flow_pointwise(z, z, rev, h_aff=ha, h_ft=hf, w=w.reshape(-1), wt=wt.reshape(-1), an_bias=ab, an_escale=ae)
        f(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print("C=%d %dx%d rev=%d: %.1f us  %.2f TB/s" % (C, hw, hw, rev, us, 20.0 * C * B * hw * hw / us / 1e6))
