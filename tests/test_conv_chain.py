"""-m gpu: the fused conv chain (bfsr_conv_chain_*, conv_chain.hip) -- the dense blocks of the RRDB encoder in ONE persistent launch with per-tile
dependency counters (RRDBNet_arch.py:25-65, LINF-LP/models/rrdb.py:38-74).  Its arithmetic is conv3x3_h2x_kernel's, so the bar is BIT-identity
with the same convs launched one by one through bfsr_conv3x3_h2x (which tests/test_hip_ops.py holds to an fp64 conv) -- run after run, on ragged
sizes, on sizes whose rows do not align with cache lines (false sharing between tiles), with every output format, and under a shrunken grid that
forces workgroups to wait for each other and for themselves."""
import numpy as np
import pytest
import torch

from cpu_ops import CpuOps

pytestmark = pytest.mark.gpu
CPU = CpuOps()


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


@pytest.fixture(scope="module")
def hip():
    from bfsr_amd.ops import HipOps
    return HipOps("cuda:0")


CASES = [(1, 32, 32, 16, 32), (2, 64, 32, 19, 45), (1, 192, 64, 33, 65), (3, 96, 32, 128, 128), (2, 64, 24, 9, 33), (1, 16, 40, 70, 70),
         (5, 48, 32, 40, 40), (2, 64, 64, 50, 40), (1, 160, 104, 17, 31), (1, 16, 8, 5, 3), (2, 32, 16, 64, 64)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["f32_out", "h2_out", "quad_out"])
def test_chain_of_one_is_fp32_accurate(hip, case, mode):
    """One conv through the chain kernel against an fp64 conv of the SAME 22-bit inputs with the unsplit fp32 weights (the bar of
    test_conv_h2x_is_fp32_accurate), weights spanning five orders of magnitude, all three output formats + the second fp32 copy."""
    B, Cin, Cout, H, W = case
    x, w, b = rnd(71, B, Cin, H, W), rnd(72, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9)), rnd(73, Cout, scale=0.3)
    w = w * torch.logspace(-4, 0, Cout).view(-1, 1, 1, 1) * 3.0
    xd = hip.to_device(x)
    xh = hip.h2_pack(xd, hip.h2_empty(B, Cin, H, W))
    x22 = hip.h2_unpack(xh, hip.empty(B, Cin, H, W)).cpu()
    ref64 = torch.nn.functional.conv2d(x22.double(), w.double(), b.double(), 1, 1)
    ref64 = torch.where(ref64 > 0, ref64, ref64 * 0.2)
    pw, epi = hip.pack_conv_x3(w, 1, lazy=True), hip.pack_epilogue(Cout, bias=b)
    tol = 4e-6 * float(ref64.abs().max())
    f32 = hip.conv(xd.new_tensor(x22), hip.pack_conv(w, 1), hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
    err32 = float((f32.cpu().double() - ref64).abs().max())
    if mode == "h2_out":
        if Cout % 8:
            pytest.skip("h2 outputs need Cout % 8 == 0")
        yh = hip.h2_empty(B, Cout, H, W)
        yh.fill_(float("nan"))
        y2 = hip.empty(B, Cout, H, W).fill_(float("nan"))
        hip.conv_chain([dict(x=xh, pw=pw, out=yh, epi=epi, act=2, slope=0.2, out2=y2)]).run()
        got = hip.h2_unpack(yh, hip.empty(B, Cout, H, W)).cpu().double()
        assert float((got - ref64).abs().max()) <= max(tol, 4 * err32) + 2.0 ** -21 * float(ref64.abs().max()), "chain h2 out %s" % (case,)
        assert torch.equal(yh, hip.conv_h2x(xh, pw, hip.h2_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)), "chain-of-one != conv_h2x (h2 out)"
        assert torch.equal(y2, hip.h2_unpack(yh, hip.empty(B, Cout, H, W))), "the second fp32 copy is not h2_unpack(y) %s" % (case,)
    else:
        if mode == "quad_out" and Cout % 8:
            pytest.skip("quad-major outputs need Cout % 8 == 0")
        out = hip.empty(B, Cout, H, W).fill_(float("nan"))
        hip.conv_chain([dict(x=xh, pw=pw, out=out, epi=epi, act=2, slope=0.2, y_fmt=1 if mode == "quad_out" else 0)]).run()
        if mode == "quad_out":
            out = out.view(B, Cout // 4, H, W, 4).permute(0, 1, 4, 2, 3).reshape(B, Cout, H, W)
        same = hip.conv_h2x(xh, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
        assert torch.equal(out, same), "chain-of-one != conv_h2x (%s)" % mode
        err = float((out.cpu().double() - ref64).abs().max())
        assert err <= max(tol, 4 * err32), "chain %s %s: max-abs %g (native fp32 kernel %g, |ref|max %g)" % (mode, case, err, err32, float(ref64.abs().max()))
    hip.check_range()


def _rrdb_weights(seed, nrdb):
    ws = []
    for r in range(nrdb):
        for i, (cin, cout) in enumerate(((64, 32), (96, 32), (128, 32), (160, 32), (192, 64))):
            ws.append((rnd(seed + 10 * r + i, cout, cin, 3, 3, scale=0.7 / np.sqrt(cin * 9)), rnd(seed + 500 + 10 * r + i, cout, scale=0.1)))
    return ws


def _rrdb_specs(hip, ring, packed, nrdb, tap=None):
    """Descriptor list of `nrdb` dense blocks walking the 4-buffer ring exactly as RRDBEncoder does (RRDB-level residual every third block)."""
    specs, cur, x_rrdb = [], 0, None
    for r in range(nrdb):
        D = ring[cur]
        if r % 3 == 0:
            x_rrdb = D[:, :8]
        for i in range(4):
            pw, epi = packed[5 * r + i]
            specs.append(dict(x=D[:, :8 + 4 * i], pw=pw, out=D[:, 8 + 4 * i: 12 + 4 * i], epi=epi, act=2, slope=0.2))
        pw, epi = packed[5 * r + 4]
        nxt = (cur + 1) % 4
        sp = dict(x=D, pw=pw, out=ring[nxt][:, :8], epi=epi, res1=D[:, :8], alpha1=0.2)
        if r % 3 == 2:
            sp.update(res2=x_rrdb, alpha2=0.2)
            if tap is not None:
                sp["out2"] = tap
        specs.append(sp)
        cur = nxt
    return specs, cur


def _run_unfused(hip, specs, one_by_one_chain):
    for sp in specs:
        if one_by_one_chain:
            hip.conv_chain([sp]).run()
        else:
            kw = {k: v for k, v in sp.items() if k not in ("x", "pw", "out", "out2")}
            hip.conv_h2x(sp["x"], sp["pw"], sp["out"], **kw)


@pytest.mark.parametrize("shape", [(2, 40, 72), (1, 33, 47), (3, 64, 64), (2, 21, 37), (1, 100, 70)])
@pytest.mark.parametrize("nrdb", [1, 3, 7])
@pytest.mark.parametrize("tune", [0, 5, 0x10000 + 37])
def test_dense_block_chain_is_bit_identical_to_single_launches(hip, shape, nrdb, tune):
    """`nrdb` dense blocks (5 convs each, ring of four 192-channel h2 buffers, RRDB residual every third block: 7 blocks reuse every ring buffer)
    in ONE launch == the same convs as one chain launch each == the same convs on bfsr_conv3x3_h2x, bit for bit.  Widths 47, 37, 70: rows are not
    multiples of 128 bytes, neighbouring tiles share cache lines.  tune = 5: five persistent workgroups walk the whole list, so most items wait
    for tiles another workgroup (or the workgroup itself) has not finished yet; bit 16 of tune publishes every item eagerly (no deferral)."""
    B, H, W = shape
    ws = _rrdb_weights(900, nrdb)
    packed = [(hip.pack_conv_x3(w, 1, lazy=True), hip.pack_epilogue(w.shape[0], bias=b)) for w, b in ws]
    x0 = rnd(5, B, 64, H, W)

    def fresh():
        ring = [hip.h2_empty(B, 192, H, W) for _ in range(4)]
        for t in ring:
            t.fill_(float("nan"))
        hip.h2_pack(hip.to_device(x0), ring[0][:, :8])
        return ring

    ringA = fresh()
    tapA = hip.empty(B, 64, H, W).fill_(float("nan"))
    specsA, curA = _rrdb_specs(hip, ringA, packed, nrdb, tap=tapA if nrdb >= 3 else None)
    chain = hip.conv_chain(specsA)
    chain.run(tune=tune)
    ringB = fresh()
    specsB, curB = _rrdb_specs(hip, ringB, packed, nrdb)
    _run_unfused(hip, specsB, one_by_one_chain=True)
    hip.check_range()
    a, b = ringA[curA][:, :8], ringB[curB][:, :8]
    assert not torch.isnan(a.float()).any()
    assert torch.equal(a, b), "fused chain differs from one-launch-per-conv: %d elements" % int((a != b).sum())
    if nrdb == 1:
        for k in range(4):                                           # every intermediate slice as well
            assert torch.equal(ringA[0][:, 8 + 4 * k: 12 + 4 * k], ringB[0][:, 8 + 4 * k: 12 + 4 * k])
    ringC = fresh()
    specsC, curC = _rrdb_specs(hip, ringC, packed, nrdb)
    _run_unfused(hip, specsC, one_by_one_chain=False)
    assert torch.equal(a, ringC[curC][:, :8]), "fused chain differs from conv_h2x launches: %d elements" % int((a != ringC[curC][:, :8]).sum())
    va = hip.h2_unpack(a, hip.empty(B, 64, H, W)).cpu()
    if nrdb >= 3:                                                     # the fp32 copy of the last tapped block output
        last_tap_block = (nrdb // 3) * 3 - 1
        if last_tap_block == nrdb - 1:
            assert torch.equal(tapA.cpu(), va), "tapped copy != h2_unpack of the block output"
    # run-to-run: the same chain again on fresh buffers
    for rep in range(3):
        ringD = fresh()
        specsD, curD = _rrdb_specs(hip, ringD, packed, nrdb)
        hip.conv_chain(specsD).run(tune=(0, 3, 64)[rep])
        assert torch.equal(ringD[curD][:, :8], a), "repetition %d differs" % rep


def test_chain_overflow_is_loud(hip):
    B, Cin, Cout, H, W = 1, 64, 32, 16, 32
    x = rnd(303, B, Cin, H, W)
    w = rnd(304, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9))
    hip.check_range()
    xh = hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Cin, H, W))
    hip.conv_chain([dict(x=xh, pw=hip.pack_conv_x3(w * 1.0e5, 1, lazy=True), out=hip.h2_empty(B, Cout, H, W))]).run()
    with pytest.raises(RuntimeError, match="range of the two-term fp16 split"):
        hip.check_range()
    hip.check_range()
