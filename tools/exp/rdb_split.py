"""Does splitting the batch of a dense-block chain over two streams fill the tile-quantisation holes?  20 dense blocks (5 dependent convs each) at
8 x 160^2 on one stream, against the same work as two independent half-batch chains on two streams.  GPU box: python tools/exp/rdb_split.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, NB = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 160, 20)
D = [ops.h2_pack(torch.randn(B, 192, H, H, device="cuda"), ops.h2_empty(B, 192, H, H)) for _ in range(2)]
pws, epis = [], []
for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
    pws.append(ops.pack_conv_x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.01, 1, lazy=True))
    epis.append(ops.pack_epilogue(cout, bias=torch.zeros(cout)))
def chain(b0, b1):
    cur = 0
    for _ in range(NB):
        Dc, Dn = D[cur][b0:b1], D[cur ^ 1][b0:b1]
        for i, cin in enumerate((64, 96, 128, 160)):
            ops.conv_h2x(Dc[:, :cin // 8], pws[i], Dc[:, cin // 8: cin // 8 + 4], epi=epis[i], act=2)
        ops.conv_h2x(Dc, pws[4], Dn[:, :8], epi=epis[4], res1=Dc[:, :8], alpha1=0.2)
        cur ^= 1
main = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def split(parts):
    streams = [s1, s2, torch.cuda.Stream(), torch.cuda.Stream()][:parts]
    step = B // parts
    for i, s in enumerate(streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            chain(i * step, (i + 1) * step)
    for s in streams:
        main.wait_stream(s)
def timed(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t1 = timed(lambda: chain(0, B))
print("B=%d %dx%d, %d dense blocks: one stream %.3f ms (%.1f us per block)" % (B, H, H, NB, t1, t1 / NB * 1e3), flush=True)
for parts in (2, 4):
    if B % parts == 0:
        t = timed(lambda: split(parts))
        print("   %d streams x B=%d: %.3f ms (%.1f us per block)" % (parts, B // parts, t, t / NB * 1e3), flush=True)
