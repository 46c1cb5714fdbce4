"""Where an item of the fused dense-block launch spends its time: per-item phase timestamps of 16 workgroups (conv_chain.hip built with
-DBFSR_CHAIN_TRACE=1, tools/exp/chain_trace.sh).  GPU box: BFSR_HIP_LIB=$PWD/tools/exp/libchain_trace.so python tools/exp/chain_trace.py [B H NB]
Fields per item (wall_clock64 = 100 MHz): 0 item start (compute wave 0), 1 first chunk landed, 2 K loop done, 3 epilogue issued, 4 publish done (0: deferred),
5 / 9 clock64 at start / end, 6 conv | nchunk << 16 | group << 32, 7 loader 0: first dependency poll, 8 loader 0: dependencies satisfied."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, NB = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 160, 69)
ring = [ops.h2_pack(torch.randn(B, 192, H, H, device="cuda") * 0.5 if _ == 0 else torch.zeros(B, 192, H, H, device="cuda"), ops.h2_empty(B, 192, H, H)) for _ in range(4)]
shapes = ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64))
allw = [[ops.pack_conv_x3(torch.randn(co, ci, 3, 3, generator=g) * (0.05 / (ci * 9) ** 0.5), 1, lazy=True) for ci, co in shapes] for _ in range(NB)]
epis = [ops.pack_epilogue(co, bias=torch.zeros(co)) for ci, co in shapes]
sp, cur = [], 0
for r in range(NB):
    D, Dn = ring[cur], ring[(cur + 1) % 4]
    for i, (ci, co) in enumerate(shapes[:4]):
        sp.append(dict(x=D[:, :ci // 8], pw=allw[r][i], out=D[:, ci // 8: ci // 8 + 4], epi=epis[i], act=2, slope=0.2))
    sp.append(dict(x=D, pw=allw[r][4], out=Dn[:, :8], epi=epis[4], res1=D[:, :8], alpha1=0.2))
    cur = (cur + 1) % 4
ch = ops.conv_chain(sp)
for _ in range(3):
    ch.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ch.run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flops = 2 * 9 * sum(ci * co for ci, co in shapes) * B * H * H * NB
print("B=%d %dx%d, %d dense blocks in one launch: %.3f ms, %.1f us per block, %.0f TFLOP/s-eq (%.3f of 833)" % (B, H, H, NB, ms, ms / NB * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 833))
if not hasattr(ops.lib, "bfsr_chain_trace_read"):
    sys.exit(0)
WG, NI, NF = 16, 1024, 12
buf = np.zeros(WG * NI * NF, dtype=np.uint64)
rc = ops.lib.bfsr_chain_trace_read(C.c_void_p(buf.ctypes.data), C.c_longlong(buf.nbytes))
assert rc == 0, rc
T = buf.reshape(WG, NI, NF).astype(np.int64)
tiles = B * ((H + 15) // 16) * ((H + 31) // 32)
nitems = NB * tiles * 6
per_wg = nitems // 256
rows = []
for w in range(WG):
    n = min(NI, per_wg)
    t = T[w, :n]
    ok = t[:, 0] > 0
    t = t[ok]
    if len(t) < 3:
        continue
    start, land, kend, epi, pub = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4]
    nxt = np.concatenate([start[1:], [0]])
    conv, nch = t[:, 6] & 0xffff, (t[:, 6] >> 16) & 0xffff
    poll, ready = t[:, 7], t[:, 8]
    for i in range(1, len(t) - 1):
        rows.append((w, int(conv[i] % 5), int(nch[i]), (land[i] - start[i]) / 100.0, (kend[i] - land[i]) / 100.0, (epi[i] - kend[i]) / 100.0,
                     ((pub[i] - epi[i]) / 100.0 if pub[i] else 0.0), (nxt[i] - start[i]) / 100.0,
                     (ready[i] - poll[i]) / 100.0 if ready[i] and poll[i] else 0.0, (start[i] - ready[i]) / 100.0 if ready[i] else 0.0, int(pub[i] != 0),
                     (t[i, 9] - t[i, 5]) / max((max(pub[i], epi[i]) - start[i]) / 100.0, 1e-9)))
R = np.array(rows)
print("items traced: %d (16 workgroups); clock64 ticks per us over an item: median %.1f" % (len(R), np.median(R[:, 11])))
print("conv nchunk |   n | wait 1st chunk | K loop | us/chunk | epilogue | publish (share not deferred) | item total | deps: poll->ready | ready->start | MFMA-only floor")
for cv in range(5):
    m = R[:, 1] == cv
    if not m.any():
        continue
    r = R[m]
    nchunk = r[0, 2]
    floor = nchunk * 54 * 32 * 2  # pipe cycles per SIMD and item (two waves per SIMD)
    print("conv%d  %4d  | %4d | %6.2f (p90 %5.2f) | %6.2f | %6.3f | %6.2f | %5.2f (%.2f) | %6.2f | %5.2f (p90 %5.2f) | %6.2f | %d cycles" % (
        cv + 1, nchunk, len(r), r[:, 3].mean(), np.percentile(r[:, 3], 90), r[:, 4].mean(), r[:, 4].mean() / nchunk, r[:, 5].mean(), r[:, 6].mean(), r[:, 10].mean(),
        r[:, 7].mean(), r[:, 8].mean(), np.percentile(r[:, 8], 90), r[:, 9].mean(), floor))
tot = R[:, 7].sum()
print("share of traced item time: wait for first chunk %.3f, K loop %.3f, epilogue %.3f, publish %.3f" % (R[:, 3].sum() / tot, R[:, 4].sum() / tot, R[:, 5].sum() / tot, R[:, 6].sum() / tot))
print("per traced workgroup (slot = 16 * index; XCD = slot // 32): mean wait for the first chunk | K loop us per chunk | epilogue | items")
for w in range(WG):
    r = R[R[:, 0] == w]
    if len(r):
        print("  wg %3d xcd %d | %5.2f | %5.3f | %5.2f | %d" % (16 * w, (16 * w) // 32, r[:, 3].mean(), (r[:, 4] / r[:, 2]).mean(), r[:, 5].mean(), len(r)))
