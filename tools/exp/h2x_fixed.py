"""conv_h2x: launch time against the number of 16-channel chunks and of item rounds (fit: rounds * (fixed + nchunk * per_chunk) + launch).
GPU box: python tools/exp/h2x_fixed.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
def timed(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, note in ((5, "250 items: 1 round"), (8, "400 items: 2 rounds"), (10, "500 items: 2 rounds"), (15, "750 items: 3 rounds")):
    row = []
    for cin in (16, 32, 64, 128, 192, 256):
        x = ops.h2_pack(torch.randn(B, cin, 160, 160, device="cuda"), ops.h2_empty(B, cin, 160, 160))
        y = ops.h2_empty(B, 32, 160, 160)
        pw = ops.pack_conv_x3(torch.randn(32, cin, 3, 3, generator=g) * 0.03, 1)
        epi = ops.pack_epilogue(32, bias=torch.zeros(32))
        row.append("%d ch %.1f us" % (cin, timed(lambda: ops.conv_h2x(x, pw, y, epi=epi, act=2))))
    print("B=%d (%s): %s" % (B, note, ", ".join(row)), flush=True)
