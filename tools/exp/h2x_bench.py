"""conv_h2x (fp16 two-term split, 3 products) vs conv_x3s (3xBF16, 6 products) at the RDB shapes: per-shape time and fp32-equivalent TFLOP/s.
Usage: python tools/exp/h2x_bench.py [B H W [split]]"""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps

shapes = [tuple(int(v) for v in sys.argv[1:4])] if len(sys.argv) >= 4 else [(8, 160, 160), (16, 256, 256)]
ops = HipOps()
g = torch.Generator().manual_seed(0)
for B, H, W in shapes:
    tot = {}
    for split in ((sys.argv[4],) if len(sys.argv) > 4 else ("bf16x3", "f16x2")):
        ops.split = split
        D = ops.x3_pack(torch.randn(B, 192, H, W, device="cuda"), ops.x3_empty(B, 192, H, W))
        N = ops.x3_empty(B, 192, H, W)
        tot[split] = 0.0
        for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
            pw = ops.pack_conv_x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.03, 1)
            epi = ops.pack_epilogue(cout, bias=torch.zeros(cout))
            if cout == 32:
                run = lambda: ops.conv_x3s(D[:, :cin // 8], pw, D[:, cin // 8:cin // 8 + 4], epi=epi, act=2)
            else:
                run = lambda: ops.conv_x3s(D, pw, N[:, :8], epi=epi, res1=D[:, :8], alpha1=0.2)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tot[split] += ms
            print("%s B%d %dx%d %3d->%2d: %.3f ms  %.0f TFLOP/s fp32-equivalent" % (split, B, H, W, cin, cout, ms, 2.0 * 9 * cin * cout * B * H * W / ms * 1e-9))
    print("one RDB at B%d %dx%d: %s" % (B, H, W, ", ".join("%s %.3f ms" % kv for kv in tot.items())))
