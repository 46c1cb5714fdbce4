"""Dump tile 0's hid tile from the trace build (experiment bit 2048) and compare it with the torch semantics."""
import ctypes, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bfsr_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STEPLIB", "libstep_2048.so"))
from bfsr_amd.ops import HipOps
from cpu_ops import CpuOps
hip, CPU = HipOps("cuda:0"), CpuOps()
hip.lib.bfsr_debug_step_trace.restype = ctypes.c_int
hip.lib.bfsr_debug_step_trace.argtypes = [ctypes.c_void_p]
def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))
C, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, cn, cc2 = 1, C // 2, 2 * (C - C // 2)
z, pre = rnd(161, B, C, H, W), rnd(162, B, 64, H, W, scale=0.5)
w0, w2 = rnd(163, 64, cn, 3, 3, scale=0.1), rnd(164, 64, 64, 1, 1, scale=0.1)
s0, c0, s2, c2 = rnd(165, 64, scale=0.1), torch.exp(rnd(166, 64, scale=0.1)), rnd(167, 64, scale=0.1), torch.exp(rnd(168, 64, scale=0.1))
w4, b4, ps = rnd(173, cc2, 64, 3, 3, scale=0.02), rnd(174, cc2, scale=0.2), torch.exp(rnd(175, cc2, scale=0.2))
hpk = hip.pack_coupling_step(w0, w2, s0, c0, s2, c2, w4, b4, ps)
hid_ref = CPU.coupling_head(z, CPU.pack_coupling_head(w0, w2, s0, c0, s2, c2), pre, torch.empty(B, 64, H, W))[0]
ref = F.pad(hid_ref, (1, 40, 1, 10))[:, :8, :32]                # tile 0's halo: rows -1..6, cols -1..30, zero outside the image
zd, pd = hip.to_device(z), hip.to_device(pre)
zp = F.pad(z[:, :cn], (2, 2, 2, 2))
t1_full = F.conv2d(zp, w0)[0]                                   # [64, H+2, W+2]: positions -1..H, -1..W
t1_ref = F.pad(t1_full, (0, 40, 0, 10))[:, :8, :32]
pre_ref = F.pad(pre[0], (1, 40, 1, 10))[:, :8, :32]
t1a = torch.relu((t1_ref + pre_ref + s0.view(-1, 1, 1)) * c0.view(-1, 1, 1))
h2_ref = F.conv2d(t1a.unsqueeze(0), w2)[0]
SZ = 4096 + 32768 + 4 * 65536 + 98304
def acc_view(buf, off):
    a = torch.from_numpy(buf[off:off + 65536].view(np.float32).copy()).reshape(8, 2, 16, 2, 32)   # [wave][m][r][lhi][l31]
    out = torch.zeros(64, 8, 32)
    for m in range(2):
        for r in range(16):
            for lh in range(2):
                out[32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh] = a[:, m, r, lh, :]
    return out
def show(name, got, ref, tol):
    err = (got - ref).abs()
    print("  %s: max err %.3e" % (name, err.max().item()))
    if err.max().item() > tol:
        m = err.amax(0)
        for y in range(m.shape[0]):
            print("     row %d: " % y + "".join("#" if v > tol else "." for v in m[y].tolist()))
for rep in range(int(os.environ.get("REPS", "3"))):
    buf = torch.zeros(SZ // 8 + 8, dtype=torch.int64, device="cuda")
    assert hip.lib.bfsr_debug_step_trace(buf.data_ptr()) == 0
    hip.coupling_step(zd, hip.empty(B, C, H, W), hpk, pd, 1)
    torch.cuda.synchronize()
    hip.lib.bfsr_debug_step_trace(None)
    raw = buf.cpu().numpy().view(np.uint8)[4096:]
    print("rep", rep)
    zt = (raw[:3 * 340 * 16].view(np.uint16).reshape(3, 340, 8).astype(np.uint32) << 16).view(np.float32)
    zgot = torch.from_numpy(zt.copy()).sum(0).reshape(10, 34, 8).permute(2, 0, 1)
    zref = F.pad(z[0, :cn], (2, 40, 2, 12))[:, :10, :34]
    show("z1 tile", zgot[:cn], zref, 1e-6)
    show("S1 acc", acc_view(raw, 32768), t1_ref, 2e-5)
    show("pre", acc_view(raw, 32768 + 65536), pre_ref, 1e-7)
    bb = torch.from_numpy(raw[32768 + 3 * 65536 + 98304:32768 + 4 * 65536 + 98304].view(np.float32).copy()).reshape(8, 4, 8, 2, 32)   # [wave][c][e][lhi][l31]
    t1got = torch.zeros(64, 8, 32)
    for c in range(4):
        for e in range(8):
            for lh in range(2):
                r = (c & 1) * 8 + e
                t1got[(c >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = bb[:, c, e, lh, :]
    show("t1 (b2)", t1got, t1a, 2e-5)
    show("S2 acc", acc_view(raw, 32768 + 2 * 65536), h2_ref, 5e-5)
    hr = (raw[32768 + 3 * 65536:32768 + 3 * 65536 + 98304].view(np.uint16).reshape(3, 8, 256, 8).astype(np.uint32) << 16).view(np.float32)
    hgot = torch.from_numpy(hr.copy()).sum(0).permute(0, 2, 1).reshape(64, 8, 32)
    show("hid", hgot, ref, 5e-5)
