"""tools/exp/sgpr_war_probe.hip against the 1x1-only coupling_head (MFMA-heavy, leaves room for other waves on its SIMDs).  Build the victim library first (see the .hip header).  The aggressor is the product library's kernel as built (NOPK or not: it only has to run MFMAs).
GPU box: python tools/exp/sgpr_war_probe.py"""
import ctypes as C, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from bfsr_amd.ops import HipOps
ops = HipOps("cuda:0")
lib = C.CDLL(os.path.join(HERE, "libsgprwar.so"))
g = np.random.Generator(np.random.PCG64(3))
r = lambda *s, scale=1.0: torch.from_numpy((g.standard_normal(s) * scale).astype(np.float32))
B = 32
hp1 = ops.pack_coupling_head(None, r(64, 64, 1, 1, scale=0.1), r(64, scale=0.1), torch.exp(r(64, scale=0.1)), r(64, scale=0.1), torch.exp(r(64, scale=0.1)))
raw, h2b = torch.randn(B, 64, 96, 96, device="cuda"), ops.h2_empty(B, 64, 96, 96)
aggr = lambda: ops.coupling_head(None, hp1, raw, h2b, pre_fmt=0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
N = 1 << 22
xin, out = torch.randn(2 * N, device="cuda"), torch.empty(2 * N, device="cuda")
names = {0: "v_pk_mul_f32 reads s[12:13]; next: s_mov_b64 s[12:13], -1   (the compiled pattern)", 1: "the same with s_nop 7 in between",
         2: "v_mul_f32 reads s12; next: s_mov_b32 s12, -1", 3: "v_pk_mul_f32 reads s[12:13], pair not overwritten (control)",
         4: "dependent packed chain pk_mul -> pk_add(op_sel) -> pk_mul -> add with s_nop 0 between (as compiled)", 5: "the same chain with s_nop 4 between",
         6: "two independent VGPR x VGPR v_pk_mul_f32", 7: "v_pk_mul_f32 -> dependent v_pk_mul_f32 (s_nop 4)", 8: "v_pk_add_f32 with op_sel:[0,1] op_sel_hi:[1,0], independent inputs",
         9: "v_pk_mul_f32 -> dependent v_pk_add_f32 without op_sel (s_nop 4)"}
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
aggr(); torch.cuda.synchronize()
for mode in (4, 6, 7, 8, 9):
    run = lambda: (lib.run_war(mode, C.c_void_p(xin.data_ptr()), C.c_void_p(out.data_ptr()), N, C.c_float(0.4947), C.c_float(1.25), st()), out)[1]
    ref = run().clone(); torch.cuda.synchronize()
    assert torch.equal(ref, run()), "alone not deterministic"
    bad = nel = 0
    for rep in range(5):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(4):
                o = run()
            ev = side.record_event()
        while not ev.query():
            for _ in range(8):
                aggr()
        torch.cuda.synchronize()
        ne = (o != ref) | (o != o)
        n = int(ne.sum())
        bad += n > 0; nel = max(nel, n)
    print("mode %d  %-88s wrong in %d of 5 overlapped runs (up to %d of %d values)" % (mode, names[mode], bad, nel, 2 * N), flush=True)
