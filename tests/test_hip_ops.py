"""-m gpu: every HIP op (through the C ABI) against the CPU semantics (tests/cpu_ops.py, which the oracle and
the golden vectors pin) on seeded inputs, including ragged sizes, channel-slice views and all epilogue stages."""
import numpy as np
import pytest
import torch

from cpu_ops import CpuOps

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


@pytest.fixture(scope="module")
def hip():
    from bfsr_amd.ops import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="module")
def hipb():
    """The same library with the 3xBF16 split (exact three-term bf16 tensors, six products): BFSR_SPLIT=bf16x3."""
    from bfsr_amd.ops import HipOps
    o = HipOps("cuda:0")
    o.split = "bf16x3"
    return o


@pytest.fixture(scope="module", params=["f16x2", "bf16x3"])
def hips(request):
    """The library under each split of its fp32-accurate 16-bit contraction mode (BFSR_SPLIT)."""
    from bfsr_amd.ops import HipOps
    o = HipOps("cuda:0")
    o.split = request.param
    return o


CPU = CpuOps()


def close(a, b, tol, what=""):
    a = a.detach().cpu()
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert not torch.isnan(a).any(), what + ": NaN in output"
    assert err <= tol * ref, "%s: max-abs %.3e > %.1e * %.2f" % (what, err, tol, ref)
    return err


CONV_CASES = [
    # (B, Cin, Cout, H, W, KS, mtile)
    (1, 3, 64, 16, 16, 3, 2), (2, 64, 32, 40, 72, 3, 1), (1, 96, 32, 33, 47, 3, 1), (1, 192, 64, 20, 36, 3, 2),
    (1, 320, 128, 24, 40, 3, 2), (2, 6, 12, 18, 34, 3, 1), (1, 64, 96, 17, 65, 3, 3), (1, 48, 64, 10, 10, 3, 2),
    (1, 64, 64, 37, 50, 1, 2), (2, 70, 27, 9, 31, 1, 1), (1, 64, 96, 16, 32, 1, 3), (1, 262, 64, 12, 33, 3, 2),
    (1, 64, 24, 70, 130, 3, 1), (3, 12, 64, 64, 64, 3, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_plain(hip, case):
    B, Cin, Cout, H, W, KS, mt = case
    x = rnd(1, B, Cin, H, W)
    w = rnd(2, Cout, Cin, KS, KS, scale=1.0 / np.sqrt(Cin * KS * KS))
    ref = CPU.conv(x, CPU.pack_conv(w, mt), torch.empty(B, Cout, H, W))
    out = hip.conv(hip.to_device(x), hip.pack_conv(w, mt), hip.empty(B, Cout, H, W))
    close(out, ref, 2e-5, "conv%s" % (case,))


def test_conv_large_tile_path(hip):
    # enough pixels to take the NR=4 (16x32 tile) path, ragged edges
    B, Cin, Cout, H, W = 2, 64, 64, 370, 390
    x = rnd(3, B, Cin, H, W)
    w = rnd(4, Cout, Cin, 3, 3, scale=0.05)
    ref = CPU.conv(x, CPU.pack_conv(w, 2), torch.empty(B, Cout, H, W))
    out = hip.conv(hip.to_device(x), hip.pack_conv(w, 2), hip.empty(B, Cout, H, W))
    close(out, ref, 2e-5, "conv NR=4")


X3S_CASES = [
    # (B, Cin, Cout, H, W): single tile; ragged edges + two cout groups; many items per persistent workgroup; Cout not a multiple of 32
    (1, 16, 32, 8, 32), (2, 64, 32, 19, 45), (1, 192, 64, 33, 65), (3, 96, 32, 160, 160), (2, 48, 24, 9, 33), (1, 32, 40, 70, 70),
]


@pytest.mark.parametrize("case", X3S_CASES)
@pytest.mark.parametrize("fp32_out", [False, True])
def test_conv_x3s_and_x3_tensors(hipb, case, fp32_out):
    """conv_x3s (LDS-DMA staged 3x3 conv over x3 tensors, conv_x3s.hip) vs the fp32 conv semantics; x3 pack/unpack lossless."""
    hip = hipb
    B, Cin, Cout, H, W = case
    x, w, b = rnd(31, B, Cin, H, W), rnd(32, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9)), rnd(33, Cout, scale=0.3)
    xd = hip.to_device(x)
    x3 = hip.x3_pack(xd, hip.x3_empty(B, Cin, H, W))
    assert torch.equal(hip.x3_unpack(x3, hip.empty(B, Cin, H, W)), xd), "x3 encoding is not lossless"
    planes = x3.float()
    assert torch.equal((planes[:, :, 0] + planes[:, :, 1]) + planes[:, :, 2], xd.view(B, Cin // 8, 8, H, W).permute(0, 1, 3, 4, 2))
    ref = CPU.conv(x, CPU.pack_conv(w, 1), torch.empty(B, Cout, H, W), bias=b, act=2, slope=0.2)
    pw, epi = hip.pack_conv_x3(w, 1), hip.pack_epilogue(Cout, bias=b)
    if fp32_out:
        out = hip.conv_x3s(x3, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
    else:
        out = hip.x3_unpack(hip.conv_x3s(x3, pw, hip.x3_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2), hip.empty(B, Cout, H, W))
    close(out, ref, 2e-5, "conv_x3s%s" % (case,))
    # bit-identical to the register-staged 3xBF16 kernel (same six products, same order)
    same = hip.conv_x3(xd, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
    assert torch.equal(out, same), "conv_x3s differs from conv_x3"


@pytest.mark.parametrize("cus", [1, 7, 40, 100, 255])
def test_persistent_convs_are_independent_of_the_workgroup_count(hip, hipb, cus):
    """conv_x3s / conv_h2s walk their tiles with one persistent workgroup per CU (`tune` overrides the count, as a partitioned
    GPU would): every split of the tile list -- incl. the half-height tiles of a last partial round and a ring that wraps
    across many tiles -- must give the same bits as the default launch."""
    B, Cin, Cout, H, W = 2, 64, 40, 75, 70
    x, w, b = rnd(61, B, Cin, H, W), rnd(62, Cout, Cin, 3, 3, scale=0.05), rnd(63, Cout, scale=0.3)
    xd = hip.to_device(x)
    x3 = hipb.x3_pack(xd, hipb.x3_empty(B, Cin, H, W))
    pw, epi = hip.pack_conv_x3(w, 1), hip.pack_epilogue(Cout, bias=b)
    ref = hipb.conv_x3s(x3, pw, hipb.x3_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2).clone()
    got = hipb.conv_x3s(x3, pw, hipb.x3_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2, tune=cus)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), "conv_x3s depends on the number of workgroups"
    xh = hip.h2_pack(xd, hip.h2_empty(B, Cin, H, W))
    refx = hip.conv_h2x(xh, pw, hip.h2_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2).clone()
    gotx = hip.conv_h2x(xh, pw, hip.h2_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2, tune=cus)
    assert torch.equal(gotx.view(torch.int16), refx.view(torch.int16)), "conv_h2x depends on the number of workgroups"
    ph = hip.pack_conv_h2s(w)
    refh = hip.conv_h2s(xh, ph, hip.h2_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2).clone()
    goth = hip.conv_h2s(xh, ph, hip.h2_empty(B, Cout, H, W), epi=epi, act=2, slope=0.2, tune=cus)
    assert torch.equal(goth.view(torch.int16), refh.view(torch.int16)), "conv_h2s depends on the number of workgroups"


def test_conv_x3s_dense_block_views_and_residuals(hipb):
    """The RDB pattern (RRDBNet_arch.py:39-45): octet-sliced views of one x3 block buffer, conv5 with `x5*0.2 + x` and the
    RRDB-level `*0.2 + x_rrdb`, all residuals x3."""
    hip = hipb
    B, H, W = 2, 21, 37
    D = rnd(41, B, 192, H, W)
    xr = rnd(42, B, 64, H, W)
    D3 = hip.x3_pack(hip.to_device(D), hip.x3_empty(B, 192, H, W))
    xr3 = hip.x3_pack(hip.to_device(xr), hip.x3_empty(B, 64, H, W))
    ref = D.clone()
    w2, b2 = rnd(43, 32, 96, 3, 3, scale=0.04), rnd(44, 32, scale=0.1)
    CPU.conv(ref[:, :96].clone(), CPU.pack_conv(w2, 1), ref[:, 96:128], bias=b2, act=2, slope=0.2)
    hip.conv_x3s(D3[:, :12], hip.pack_conv_x3(w2, 1), D3[:, 12:16], epi=hip.pack_epilogue(32, bias=b2), act=2, slope=0.2)
    close(hip.x3_unpack(D3, hip.empty(B, 192, H, W)), ref, 2e-5, "x3 slice views")
    w5, b5 = rnd(45, 64, 192, 3, 3, scale=0.03), rnd(46, 64, scale=0.1)
    out_ref = CPU.conv(ref.clone(), CPU.pack_conv(w5, 1), torch.empty(B, 64, H, W), bias=b5, res1=ref[:, :64].clone(), alpha1=0.2,
                       res2=xr, alpha2=0.2)
    nxt = hip.x3_empty(B, 192, H, W)
    hip.conv_x3s(D3, hip.pack_conv_x3(w5, 1), nxt[:, :8], epi=hip.pack_epilogue(64, bias=b5), res1=D3[:, :8], alpha1=0.2, res2=xr3, alpha2=0.2)
    close(hip.x3_unpack(nxt[:, :8], hip.empty(B, 64, H, W)), out_ref, 2e-5, "x3 conv5 residuals")


H2S_CASES = [
    # (B, Cin, Cout, H, W): one tile; ragged edges + two cout groups + many chunks (the loaders run 3 stages ahead across items);
    # many items per persistent workgroup; Cout not a multiple of 32; more tiles than the ring is deep with ONE chunk each
    (1, 32, 32, 16, 32), (2, 64, 32, 19, 45), (1, 192, 64, 33, 65), (3, 96, 32, 128, 128), (2, 64, 24, 9, 33), (1, 32, 40, 70, 70),
    (5, 32, 32, 40, 40), (2, 64, 64, 50, 40), (1, 160, 104, 17, 31),
]


@pytest.mark.parametrize("case", H2S_CASES)
@pytest.mark.parametrize("mode", ["f32_out", "h2_out", "hi_only"])
def test_conv_h2s_and_h2_tensors(hip, case, mode):
    """conv_h2s (LDS-DMA staged fp16 3x3 conv over h2 tensors, conv_h2s.hip) vs the fp16-operand conv semantics (operands rounded
    to fp16, fp32 accumulation) and vs conv_f16 on the fp32 tensor (the hi plane is that kernel's staging-time rounding)."""
    B, Cin, Cout, H, W = case
    x, w, b = rnd(31, B, Cin, H, W), rnd(32, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9)), rnd(33, Cout, scale=0.3)
    xd = hip.to_device(x)
    xh = hip.h2_pack(xd, hip.h2_empty(B, Cin, H, W))
    planes = xh.float()
    want_hi = xd.view(B, Cin // 8, 8, H, W).permute(0, 1, 3, 4, 2).half()
    assert torch.equal(xh[:, :, 0], want_hi), "hi plane != fp16(x)"
    close(hip.h2_unpack(xh, hip.empty(B, Cin, H, W)), x, 1e-6, "h2 round trip (22 bits)")
    ref = CPU.conv(x.half().float(), CPU.pack_conv(w.half().float(), 1), torch.empty(B, Cout, H, W), bias=b, act=2, slope=0.2)
    pw, epi = hip.pack_conv_h2s(w), hip.pack_epilogue(Cout, bias=b)
    if mode == "f32_out":
        out = hip.conv_h2s(xh, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
        close(out, ref, 2e-5, "conv_h2s%s" % (case,))
        same = hip.conv_f16(xd, hip.pack_conv_f16(w, 1), hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
        close(out, same.cpu(), 2e-6, "conv_h2s vs conv_f16 (same operands, different summation order)")
    elif Cout % 8 == 0:
        yh = hip.h2_empty(B, Cout, H, W)
        yh.fill_(float("nan"))
        hip.conv_h2s(xh, pw, yh, epi=epi, act=2, slope=0.2, hi_only=mode == "hi_only")
        if mode == "hi_only":
            assert torch.isnan(yh[:, :, 1]).all(), "hi_only wrote the lo plane"
            o = yh[:, :, 0].float().permute(0, 1, 4, 2, 3).reshape(B, Cout, H, W)
            close(o, ref.half().float(), 2e-3, "conv_h2s hi plane")           # 1 fp16 ulp where the fp32 sums straddle a rounding boundary
        else:
            close(hip.h2_unpack(yh, hip.empty(B, Cout, H, W)), ref, 2e-5, "conv_h2s h2 out")


def test_conv_h2s_dense_block_views_and_residuals(hip):
    """The RDB pattern (LINF-LP/models/rrdb.py:52-58) on h2 tensors: octet-sliced views of one block buffer, conv5 with
    `x5*0.2 + x` and the RRDB-level `*0.2 + x_rrdb`; the residual operands are read as hi + lo (22 bits), not hi."""
    B, H, W = 2, 21, 37
    D = rnd(41, B, 192, H, W)
    xr = rnd(42, B, 64, H, W)
    Dh = hip.h2_pack(hip.to_device(D), hip.h2_empty(B, 192, H, W))
    xrh = hip.h2_pack(hip.to_device(xr), hip.h2_empty(B, 64, H, W))
    Dq = D.half().float()                                              # what the convs see
    w2, b2 = rnd(43, 32, 96, 3, 3, scale=0.04), rnd(44, 32, scale=0.1)
    x3ref = CPU.conv(Dq[:, :96].clone(), CPU.pack_conv(w2.half().float(), 1), torch.empty(B, 32, H, W), bias=b2, act=2, slope=0.2)
    hip.conv_h2s(Dh[:, :12], hip.pack_conv_h2s(w2), Dh[:, 12:16], epi=hip.pack_epilogue(32, bias=b2), act=2, slope=0.2, hi_only=True)
    got = Dh[:, 12:16, 0].float().permute(0, 1, 4, 2, 3).reshape(B, 32, H, W)
    close(got, x3ref.half().float(), 2e-3, "h2 slice view, hi plane")
    Dq[:, 96:128] = got.cpu()
    w5, b5 = rnd(45, 64, 192, 3, 3, scale=0.03), rnd(46, 64, scale=0.1)
    out_ref = CPU.conv(Dq.clone(), CPU.pack_conv(w5.half().float(), 1), torch.empty(B, 64, H, W), bias=b5, res1=D[:, :64].clone(), alpha1=0.2,
                       res2=xr, alpha2=0.2)
    nxt = hip.h2_empty(B, 192, H, W)
    hip.conv_h2s(Dh, hip.pack_conv_h2s(w5), nxt[:, :8], epi=hip.pack_epilogue(64, bias=b5), res1=Dh[:, :8], alpha1=0.2, res2=xrh, alpha2=0.2)
    close(hip.h2_unpack(nxt[:, :8], hip.empty(B, 64, H, W)), out_ref, 2e-5, "h2 conv5 residuals")


@pytest.mark.parametrize("case", [(2, 192, 64, 21, 37), (1, 64, 64, 16, 32), (3, 64, 128, 40, 70), (2, 96, 64, 5, 3), (1, 128, 192, 33, 33)])
def test_conv_h2s_64_cout_tiles_give_the_bits_of_32_cout_tiles(hip, case):
    """conv_h2s with 64 output channels per workgroup tile (mtile 2, round 6: the input tile staged once for 64 channels, per-M-tile epilogue) against
    the 32-channel form: same summation order per output element -> identical bits, with every epilogue stage and both residuals, all three outputs."""
    B, Cin, Cout, H, W = case
    x, w = rnd(131, B, Cin, H, W), rnd(132, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9))
    r1, r2 = rnd(133, B, Cout, H, W), rnd(134, B, Cout, H, W)
    xh = hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Cin, H, W))
    r1h = hip.h2_pack(hip.to_device(r1), hip.h2_empty(B, Cout, H, W))
    r2h = hip.h2_pack(hip.to_device(r2), hip.h2_empty(B, Cout, H, W))
    p1, p2 = hip.pack_conv_h2s(w, mtile=1), hip.pack_conv_h2s(w, mtile=2)
    assert (p1.mtile, p2.mtile) == (1, 2)
    epis = [hip.pack_epilogue(Cout, bias=rnd(135, Cout, scale=0.3)),
            hip.pack_epilogue(Cout, bias=rnd(135, Cout, scale=0.3), aff_shift=rnd(136, Cout, scale=0.1), aff_scale=torch.exp(rnd(137, Cout, scale=0.1)),
                              post_scale=torch.exp(rnd(138, Cout, scale=0.1)))]
    for epi in epis:
        for kw in (dict(), dict(res1=r1h, alpha1=0.2), dict(res1=r1h, alpha1=0.2, res2=r2h, alpha2=0.2)):
            a = hip.conv_h2s(xh, p1, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2, **kw)
            b = hip.conv_h2s(xh, p2, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2, **kw)
            assert torch.equal(a.cpu(), b.cpu()), "fp32 output differs (%s)" % sorted(kw)
            for hi_only in (False, True):
                ya, yb = hip.h2_empty(B, Cout, H, W).zero_(), hip.h2_empty(B, Cout, H, W).zero_()
                hip.conv_h2s(xh, p1, ya, epi=epi, act=2, slope=0.2, hi_only=hi_only, **kw)
                hip.conv_h2s(xh, p2, yb, epi=epi, act=2, slope=0.2, hi_only=hi_only, **kw)
                assert torch.equal(ya.cpu(), yb.cpu()), "h2 output differs (hi_only=%s, %s)" % (hi_only, sorted(kw))


@pytest.mark.parametrize("case", [(2, 64, 32, 21, 37), (3, 96, 32, 40, 70), (1, 128, 32, 16, 32), (2, 160, 32, 33, 33), (2, 64, 64, 50, 40), (5, 128, 24, 20, 36)])
def test_conv_h2s_resident_weights_give_the_bits_of_streamed_weights(hip, case, monkeypatch):
    """conv_h2s with the conv's whole weight tensor resident in LDS (round 6: one output-channel group, <= 8 chunks; the ring then holds input only, 4-6
    stages) against the streamed form (BFSR_H2S_RES=0): same fragments, same order -> identical bits.  B > 1 and several tiles per workgroup make the
    persistent workgroups walk items with the weights staged once; (2, 160, 32, ..) is a shape that stays streamed (10 chunks) -- the switch is a no-op there."""
    B, Cin, Cout, H, W = case
    x, w = rnd(141, B, Cin, H, W), rnd(142, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9))
    xh = hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Cin, H, W))
    pw, epi = hip.pack_conv_h2s(w), hip.pack_epilogue(Cout, bias=rnd(143, Cout, scale=0.3))
    ref = CPU.conv(x.half().float(), CPU.pack_conv(w.half().float(), 1), torch.empty(B, Cout, H, W), bias=rnd(143, Cout, scale=0.3), act=2, slope=0.2)
    outs = {}
    for res in ("1", "0"):
        monkeypatch.setenv("BFSR_H2S_RES", res)
        y = hip.h2_empty(B, (Cout + 7) // 8 * 8, H, W).zero_() if Cout % 8 == 0 else None
        o32 = hip.conv_h2s(xh, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
        if y is not None:
            hip.conv_h2s(xh, pw, y, epi=epi, act=2, slope=0.2, hi_only=True)
        outs[res] = (o32.clone(), None if y is None else y.clone())
    close(outs["1"][0], ref, 2e-5, "conv_h2s resident weights %s" % (case,))
    assert torch.equal(outs["1"][0].cpu(), outs["0"][0].cpu()), "fp32 output differs between resident and streamed weights"
    if outs["1"][1] is not None:
        assert torch.equal(outs["1"][1].cpu(), outs["0"][1].cpu()), "h2 output differs between resident and streamed weights"


H2X_CASES = [(1, 32, 32, 16, 32), (2, 64, 32, 19, 45), (1, 192, 64, 33, 65), (3, 96, 32, 128, 128), (2, 64, 24, 9, 33), (1, 16, 40, 70, 70),
             (5, 48, 32, 40, 40), (2, 64, 64, 50, 40), (1, 160, 104, 17, 31)]


@pytest.mark.parametrize("case", H2X_CASES)
@pytest.mark.parametrize("mode", ["f32_out", "h2_out"])
def test_conv_h2x_is_fp32_accurate(hip, case, mode):
    """conv_h2x (conv3x3_h2x_kernel: both planes of the h2 input x two-term fp16 weights, three products, power-of-two weight
    scale) against an fp64 conv of the SAME 22-bit inputs with the UNSPLIT fp32 weights: the error must be at the level of the
    native fp32 kernel's, also for weights spanning five orders of magnitude (the scale keeps their lo terms normal)."""
    B, Cin, Cout, H, W = case
    x, w, b = rnd(71, B, Cin, H, W), rnd(72, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9)), rnd(73, Cout, scale=0.3)
    w = w * torch.logspace(-4, 0, Cout).view(-1, 1, 1, 1) * 3.0           # per-channel magnitudes 3e-4 .. 3
    xd = hip.to_device(x)
    xh = hip.h2_pack(xd, hip.h2_empty(B, Cin, H, W))
    x22 = hip.h2_unpack(xh, hip.empty(B, Cin, H, W)).cpu()
    ref64 = torch.nn.functional.conv2d(x22.double(), w.double(), b.double(), 1, 1)
    ref64 = torch.where(ref64 > 0, ref64, ref64 * 0.2)
    pw, epi = hip.pack_conv_x3(w, 1), hip.pack_epilogue(Cout, bias=b)
    tol = 4e-6 * float(ref64.abs().max())
    if mode == "f32_out":
        out = hip.conv_h2x(xh, pw, hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
        err = float((out.cpu().double() - ref64).abs().max())
        f32 = hip.conv(xd.new_tensor(x22), hip.pack_conv(w, 1), hip.empty(B, Cout, H, W), epi=epi, act=2, slope=0.2)
        err32 = float((f32.cpu().double() - ref64).abs().max())
        assert err <= max(tol, 4 * err32), "conv_h2x %s: max-abs %g (native fp32 kernel %g, |ref|max %g)" % (case, err, err32, float(ref64.abs().max()))
    elif Cout % 8 == 0:
        yh = hip.h2_empty(B, Cout, H, W)
        yh.fill_(float("nan"))
        hip.conv_h2x(xh, pw, yh, epi=epi, act=2, slope=0.2)
        got = hip.h2_unpack(yh, hip.empty(B, Cout, H, W)).cpu().double()
        assert float((got - ref64).abs().max()) <= tol + 2.0 ** -21 * float(ref64.abs().max()), "conv_h2x h2 out %s" % (case,)


@pytest.mark.parametrize("wscale", [0.01, 1.0, 100.0])
@pytest.mark.parametrize("ascale", [1e-4, 1e-2, 1.0, 1e2, 3e3])
@pytest.mark.parametrize("kernel", ["conv_h2x", "conv_x3"])
def test_f16x2_split_vs_activation_and_weight_scale(hip, kernel, ascale, wscale):
    """Range behaviour of the two-term fp16 split, against an fp64 conv of the UNQUANTISED input (not of its 22-bit image): activations
    scaled 1e-4 ... 3e3, weights x0.01 ... x100.  The weights carry a power-of-two scale, the activations do not: x = hi + lo holds 22
    significant bits while lo = fp16(x - hi) is a normal number (|x| >= 2^-3) and degrades to an ABSOLUTE error of 2^-25 per element
    below that (fp16 subnormal spacing 2^-24).  Asserted model:  |err| <= 4 x (native fp32 kernel's error) + 2^-25 x max_co sum|w|.
    At |x| ~ 1 this is fp32-class; a tensor that is uniformly tiny keeps fewer RELATIVE bits (the printed figure) -- DESIGN.md section 3.7
    states the regime, BFSR_SPLIT=bf16x3 has fp32's full exponent range."""
    B, Cin, Cout, H, W = 1, 64, 32, 24, 40
    x = rnd(301, B, Cin, H, W) * ascale
    w = rnd(302, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9)) * wscale
    truth = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1)
    xd = hip.to_device(x)
    if kernel == "conv_h2x":
        out = hip.conv_h2x(hip.h2_pack(xd, hip.h2_empty(B, Cin, H, W)), hip.pack_conv_x3(w, 1), hip.empty(B, Cout, H, W))
    else:
        out = hip.conv_x3(xd, hip.pack_conv_x3(w, 1), hip.empty(B, Cout, H, W))
    f32 = hip.conv(xd, hip.pack_conv(w, 1), hip.empty(B, Cout, H, W))
    hip.check_range()
    err, err32 = float((out.cpu().double() - truth).abs().max()), float((f32.cpu().double() - truth).abs().max())
    floor = 2.0 ** -25 * float(w.abs().sum(dim=(1, 2, 3)).max())
    rel = err / float(truth.abs().max())
    assert err <= 4 * err32 + floor, "%s x%g w%g: max-abs %.3e (fp32 kernel %.3e, floor %.3e), relative %.2e" % (kernel, ascale, wscale, err, err32, floor, rel)
    if ascale >= 1.0:
        assert rel <= 2e-6, "%s x%g w%g: relative error %.2e is not fp32-class" % (kernel, ascale, wscale, rel)


@pytest.mark.parametrize("kernel", ["h2_pack", "conv_x3", "conv_h2x_out"])
def test_f16x2_split_overflow_is_loud(hip, kernel):
    """|x| >= 65504 cannot enter the fp16 split: the kernel that would split it raises the device flag, check_range() raises."""
    B, Cin, Cout, H, W = 1, 64, 32, 16, 32
    x = rnd(303, B, Cin, H, W)
    w = rnd(304, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9))
    hip.check_range()
    if kernel == "conv_h2x_out":                        # the overflow is produced BY the conv: its h2 output cannot hold 1e5
        xh = hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Cin, H, W))
        hip.conv_h2x(xh, hip.pack_conv_x3(w * 1.0e5, 1), hip.h2_empty(B, Cout, H, W))
    else:
        x[0, 3, 5, 7] = 1.0e5
        if kernel == "h2_pack":
            hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Cin, H, W))
        else:
            hip.conv_x3(hip.to_device(x), hip.pack_conv_x3(w, 1), hip.empty(B, Cout, H, W))
    with pytest.raises(RuntimeError, match="range of the two-term fp16 split"):
        hip.check_range()
    hip.check_range()


def test_conv_h2x_dense_block_views_and_residuals(hip):
    """The RDB pattern on h2 tensors with the fp32-class arithmetic: octet-sliced views of one block buffer, conv5 with `x5*0.2 + x`
    and the RRDB-level `*0.2 + x_rrdb` (RRDBNet_arch.py:39-45, :59-65)."""
    B, H, W = 2, 21, 37
    D = rnd(41, B, 192, H, W)
    xr = rnd(42, B, 64, H, W)
    Dh = hip.h2_pack(hip.to_device(D), hip.h2_empty(B, 192, H, W))
    xrh = hip.h2_pack(hip.to_device(xr), hip.h2_empty(B, 64, H, W))
    ref = hip.h2_unpack(Dh, hip.empty(B, 192, H, W)).cpu()              # what the convs see: 22-bit values
    xr22 = hip.h2_unpack(xrh, hip.empty(B, 64, H, W)).cpu()
    w2, b2 = rnd(43, 32, 96, 3, 3, scale=0.04), rnd(44, 32, scale=0.1)
    CPU.conv(ref[:, :96].clone(), CPU.pack_conv(w2, 1), ref[:, 96:128], bias=b2, act=2, slope=0.2)
    hip.conv_x3s(Dh[:, :12], hip.pack_conv_x3(w2, 1), Dh[:, 12:16], epi=hip.pack_epilogue(32, bias=b2), act=2, slope=0.2)
    close(hip.h2_unpack(Dh, hip.empty(B, 192, H, W)), ref, 2e-5, "h2 slice views (conv_x3s dispatch on the tensor type)")
    w5, b5 = rnd(45, 64, 192, 3, 3, scale=0.03), rnd(46, 64, scale=0.1)
    out_ref = CPU.conv(ref.clone(), CPU.pack_conv(w5, 1), torch.empty(B, 64, H, W), bias=b5, res1=ref[:, :64].clone(), alpha1=0.2,
                       res2=xr22, alpha2=0.2)
    nxt = hip.h2_empty(B, 192, H, W)
    hip.conv_h2x(Dh, hip.pack_conv_x3(w5, 1), nxt[:, :8], epi=hip.pack_epilogue(64, bias=b5), res1=Dh[:, :8], alpha1=0.2, res2=xrh, alpha2=0.2)
    close(hip.h2_unpack(nxt[:, :8], hip.empty(B, 64, H, W)), out_ref, 2e-5, "h2x conv5 residuals")


def test_conv_epilogue_all_stages(hip):
    B, Cin, Cout, H, W = 2, 40, 48, 21, 35
    x, w = rnd(5, B, Cin, H, W), rnd(6, Cout, Cin, 3, 3, scale=0.08)
    v = lambda s: rnd(s, Cout, scale=0.5)
    kw = dict(bias=v(7), aff_shift=v(8), aff_scale=torch.exp(v(9)), aff_post=v(10), post_scale=torch.exp(v(11)))
    pre, r1, r2 = rnd(12, B, Cout, H, W), rnd(13, B, Cout, H, W), rnd(14, B, Cout, H, W)
    for act in (0, 1, 2):
        ref = CPU.conv(x, CPU.pack_conv(w, 2), torch.empty(B, Cout, H, W), pre_add=pre, act=act, slope=0.2,
                       res1=r1, alpha1=0.2, res2=r2, alpha2=0.3, **kw)
        dkw = dict(kw)
        out = hip.conv(hip.to_device(x), hip.pack_conv(w, 2), hip.empty(B, Cout, H, W), pre_add=hip.to_device(pre),
                       act=act, slope=0.2, res1=hip.to_device(r1), alpha1=0.2, res2=hip.to_device(r2), alpha2=0.3, **dkw)
        close(out, ref, 2e-5, "conv epilogue act=%d" % act)


def test_conv_channel_slice_views_and_inplace_1x1(hip):
    # dense-block pattern: read channels [:96] of a 192-ch buffer, write channels [96:128] of the same buffer
    B, H, W = 2, 19, 45
    D = rnd(15, B, 192, H, W)
    w = rnd(16, 32, 96, 3, 3, scale=0.05)
    Dd = hip.to_device(D)
    ref = D.clone()
    CPU.conv(ref[:, :96].clone(), CPU.pack_conv(w, 1), ref[:, 96:128], act=2)
    hip.conv(Dd[:, :96], hip.pack_conv(w, 1), Dd[:, 96:128], act=2)
    close(Dd, ref, 2e-5, "conv slice views")
    # 1x1 in place on a slice
    w1 = rnd(17, 64, 64, 1, 1, scale=0.1)
    ref2 = ref.clone()
    CPU.conv(ref2[:, 64:128].clone(), CPU.pack_conv(w1, 2), ref2[:, 64:128], act=1)
    hip.conv(Dd[:, 64:128], hip.pack_conv(w1, 2), Dd[:, 64:128], act=1)
    close(Dd, ref2, 2e-5, "conv 1x1 in place")


def test_conv_nearest_upsample_on_read(hip):
    B, C, h, w_ = 2, 64, 13, 21
    x, w = rnd(18, B, C, h, w_), rnd(19, 64, C, 3, 3, scale=0.05)
    b = rnd(20, 64, scale=0.1)
    ref = CPU.conv(x, CPU.pack_conv(w, 2), torch.empty(B, 64, 2 * h, 2 * w_), in_shift=1, bias=b, act=2)
    out = hip.conv(hip.to_device(x), hip.pack_conv(w, 2), hip.empty(B, 64, 2 * h, 2 * w_), in_shift=1, bias=b, act=2)
    close(out, ref, 2e-5, "conv in_shift")


def _h2_values(t):
    """fp32 [B,C,H,W] values (hi + lo) of an h2 tensor, on the CPU."""
    return sum(CPU._h2_planes(t.cpu()))


@pytest.mark.parametrize("Cz", [6, 12])
@pytest.mark.parametrize("hw", [(16, 40), (9, 33), (70, 70), (4, 32), (3, 5)])
def test_coupling_head(hip, Cz, hw):
    """coupling.hip head: 3x3 on z1 (+ hoisted partial, ActNorm, ReLU) -> register-chained 1x1 (+ ActNorm, ReLU) on the two-term fp16
    split, hid as an h2 tensor.  Ragged last tiles, images smaller than a tile; B = 3 exercises the persistent tile walk."""
    H, W = hw
    B = 3
    z, pre = rnd(61, B, 2 * Cz, H, W), rnd(62, B, 64, H, W, scale=0.5)
    w0, w2 = rnd(63, 64, Cz, 3, 3, scale=0.1), rnd(64, 64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = rnd(65, 64, scale=0.1), torch.exp(rnd(66, 64, scale=0.1)), rnd(67, 64, scale=0.1), torch.exp(rnd(68, 64, scale=0.1))
    ref = CPU.coupling_head(z, CPU.pack_coupling_head(w0, w2, s0, c0, s2, c2), pre, torch.empty(B, 64, H, W))
    pk = hip.pack_coupling_head(w0, w2, s0, c0, s2, c2)
    out = hip.coupling_head(hip.to_device(z), pk, hip.to_device(pre), hip.h2_empty(B, 64, H, W))
    close(_h2_values(out), ref, 2e-5, "coupling_head Cz=%d" % Cz)
    # pre_aff handed over quad-major ([B][16][H][W][4], 16-byte loads): bit-identical result
    o1 = hip.coupling_head(hip.to_device(z), pk, hip.to_device(CPU.quads(pre).contiguous()), hip.h2_empty(B, 64, H, W), pre_fmt=1)
    assert torch.equal(o1.cpu(), out.cpu()), "pre_fmt=1 must give the values of pre_fmt=0"
    hip.check_range()


def test_coupling_head_on_channel_slices(hip):
    """z1 and pre_aff as channel slices of wider buffers (batch stride != C*H*W), as the engine passes them."""
    B, Cz, H, W = 2, 6, 20, 44
    zbuf, prebuf = rnd(181, B, 20, H, W), rnd(182, B, 3 * 64, H, W, scale=0.5)
    w0, w2 = rnd(163, 64, Cz, 3, 3, scale=0.1), rnd(164, 64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = rnd(165, 64, scale=0.1), torch.exp(rnd(166, 64, scale=0.1)), rnd(167, 64, scale=0.1), torch.exp(rnd(168, 64, scale=0.1))
    ref = CPU.coupling_head(zbuf[:, :12], CPU.pack_coupling_head(w0, w2, s0, c0, s2, c2), prebuf[:, 64:128], torch.empty(B, 64, H, W))
    out = hip.coupling_head(hip.to_device(zbuf)[:, :12], hip.pack_coupling_head(w0, w2, s0, c0, s2, c2), hip.to_device(prebuf)[:, 64:128],
                            hip.h2_empty(B, 64, H, W))
    close(_h2_values(out), ref, 2e-5, "coupling_head on slices")


def test_coupling_head_range_guard_is_loud(hip):
    """The fp16 split cannot represent |x| >= 2^15: the kernel raises the device flag and check_range() turns it into an error."""
    B, Cz, H, W = 1, 6, 8, 32
    z, pre = rnd(61, B, 2 * Cz, H, W), rnd(62, B, 64, H, W, scale=0.5)
    w0, w2 = rnd(63, 64, Cz, 3, 3, scale=0.1), rnd(64, 64, 64, 1, 1, scale=0.1)
    one = torch.ones(64)
    pk = hip.pack_coupling_head(w0, w2, 0 * one, one, 0 * one, one)
    hip.check_range()                                  # clean so far
    zb = z.clone()
    zb[0, 2, 3, 7] = 7.0e4
    hip.coupling_head(hip.to_device(zb), pk, hip.to_device(pre), hip.h2_empty(B, 64, H, W))
    with pytest.raises(RuntimeError, match="range of the two-term fp16 split"):
        hip.check_range()
    hip.check_range()                                  # the flag was cleared
    zb[0, 2, 3, 7] = float("nan")
    hip.coupling_head(hip.to_device(zb), pk, hip.to_device(pre), hip.h2_empty(B, 64, H, W))
    with pytest.raises(RuntimeError):
        hip.check_range()


@pytest.mark.parametrize("C", [12, 24])
@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("hw", [(16, 40), (9, 33), (70, 70), (5, 3), (37, 91)])
def test_coupling_tail(hip, C, reverse, hw):
    """Coupling tail (conv3x3_h2x_kernel, coupling epilogue): Conv2dZeros 64 -> 2*(C - C/2) over the h2 tensor hid + the FlowStep
    pointwise chain, in place.  The reference gets the SAME hid values (hi + lo of the h2 tensor)."""
    H, W = hw
    B, cc2 = 3, 2 * (C - C // 2)
    hid, z = rnd(71, B, 64, H, W).abs(), rnd(72, B, C, H, W)
    w4, b4, ps = rnd(73, cc2, 64, 3, 3, scale=0.02), rnd(74, cc2, scale=0.2), torch.exp(rnd(75, cc2, scale=0.2))
    h_ft = rnd(76, B, 2 * C, H, W, scale=0.5)
    Wm = torch.from_numpy(np.linalg.qr(np.random.Generator(np.random.PCG64(7)).standard_normal((C, C)))[0].astype(np.float32))
    bias, es = rnd(77, C, scale=0.1), torch.exp(rnd(78, C, scale=0.1))
    hid_h2 = CPU.h2_pack(hid, CPU.h2_empty(B, 64, H, W))
    hid22 = sum(CPU._h2_planes(hid_h2))
    tpk = hip.pack_coupling_tail(w4, b4, ps)
    for kw in (dict(h_ft=h_ft, w=Wm, an_bias=bias, an_escale=es), dict()) if not reverse else (dict(h_ft=h_ft, w=Wm, an_bias=bias, an_escale=es),):
        ref = CPU.coupling_tail(hid22, CPU.pack_coupling_tail(w4, b4, ps), z, torch.empty_like(z), reverse,
                                **{k: (v.reshape(-1) if k == "w" else v) for k, v in kw.items()})
        zd = hip.to_device(z).clone()
        dkw = {k: (hip.vec(v) if v.dim() <= 2 else hip.to_device(v)) for k, v in kw.items()}
        hip.coupling_tail(hid_h2.to(hip.device), tpk, zd, zd, reverse, **dkw)
        close(zd, ref, 2e-5, "coupling_tail C=%d rev=%d" % (C, reverse))
        if "h_ft" in kw:                          # h_ft handed over quad-major: bit-identical result
            z1 = hip.to_device(z).clone()
            dk2 = dict(dkw, h_ft=hip.to_device(CPU.quads(h_ft).contiguous()))
            hip.coupling_tail(hid_h2.to(hip.device), tpk, z1, z1, reverse, h_ft_fmt=1, **dk2)
            assert torch.equal(z1.cpu(), zd.cpu()), "coupling_tail h_ft_fmt=1 differs from h_ft_fmt=0"
    hip.check_range()


@pytest.mark.parametrize("hw", [(16, 40), (9, 33), (70, 70)])
@pytest.mark.parametrize("pre_fmt", [0, 1])
def test_coupling_head_without_3x3_stage(hip, hw, pre_fmt):
    """The 1x1-only form of coupling_head (Cz = 0): hid = relu(AN2(W2 . relu(AN0(pre)))) -- fFeatures.0's ActNorm + ReLU on the hoisted conv
    result, then fFeatures.2 (FlowAffineCouplingsAblation.py:127-135), h2 output."""
    H, W = hw
    B = 2
    pre = rnd(62, B, 64, H, W, scale=0.8)
    w2 = rnd(64, 64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = rnd(65, 64, scale=0.1), torch.exp(rnd(66, 64, scale=0.1)), rnd(67, 64, scale=0.1), torch.exp(rnd(68, 64, scale=0.1))
    ref = CPU.coupling_head(None, CPU.pack_coupling_head(None, w2, s0, c0, s2, c2), pre, torch.empty(B, 64, H, W))
    pd = hip.to_device(CPU.quads(pre).contiguous() if pre_fmt else pre)
    out = hip.coupling_head(None, hip.pack_coupling_head(None, w2, s0, c0, s2, c2), pd, hip.h2_empty(B, 64, H, W), pre_fmt=pre_fmt)
    close(_h2_values(out), ref, 2e-5, "coupling_head Cz=0")


@pytest.mark.parametrize("Cout", [12, 24, 32])
@pytest.mark.parametrize("hw", [(16, 40), (9, 33), (70, 70), (5, 3)])
def test_conv_h2r(hip, Cout, hw):
    """bfsr_conv3x3_h2r (the coupling tail's conv kernel with a plain epilogue): 3x3 conv 64 -> Cout <= 32 over an h2 tensor with bias, post-scale
    and activation, fp32 NCHW and quad-major output (bit-identical values), channel-slice output views."""
    H, W = hw
    B = 3
    x = rnd(71, B, 64, H, W)
    w, b, ps = rnd(73, Cout, 64, 3, 3, scale=0.03), rnd(74, Cout, scale=0.2), torch.exp(rnd(75, Cout, scale=0.2))
    xh = CPU.h2_pack(x, CPU.h2_empty(B, 64, H, W))
    x22 = sum(CPU._h2_planes(xh))
    ref = CPU.conv(x22, CPU.pack_conv(w, 1), torch.empty(B, Cout, H, W), bias=b, post_scale=ps, act=2, slope=0.2)
    pk, epi = hip.pack_coupling_tail(w, b, ps), hip.pack_epilogue(Cout, bias=b, post_scale=ps)
    wide = hip.zeros(B, Cout + 8, H, W)
    out = hip.conv_h2r(xh.to(hip.device), pk, wide[:, 4:4 + Cout], epi=epi, act=2, slope=0.2)
    close(out, ref, 2e-5, "conv_h2r Cout=%d" % Cout)
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0, "wrote outside its channel slice"
    wq = hip.zeros(B, Cout + 8, H, W)
    hip.conv_h2r(xh.to(hip.device), pk, wq[:, 4:4 + Cout], epi=epi, act=2, slope=0.2, y_fmt=1)
    assert torch.equal(CPU.quads(wq[:, 4:4 + Cout].cpu().contiguous(), inverse=True), out.cpu()), "conv_h2r y_fmt=1"
    assert float(wq[:, :4].abs().max()) == 0.0 and float(wq[:, 4 + Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(2, 32, 64, 8, 32), (1, 64, 32, 21, 37), (3, 16, 96, 5, 70), (2, 256, 64, 24, 40), (1, 16, 32, 9, 33), (1, 16, 32, 1, 1)])
def test_conv_up4_h2t(hip, case):
    """bfsr_conv2d_up4_h2t: conv3x3(nearest_up4(taps)) + pre_add evaluated at source resolution (25 pre-summed weight blocks, nine phase classes),
    two-term fp16 split (three products), quad-major fp32 output with and without pre_add (also in place): against an fp64 conv of the SAME
    22-bit inputs with the unsplit fp32 weights (SRFlowNet_arch.py:122-137 with RRDBNet_arch.py:105-112 fea_up4: F.interpolate(..., mode='nearest')
    + Conv2d).  Ragged tiles in both directions, several output-channel groups, per-channel weight magnitudes over four decades, a 1 x 1 source."""
    B, Ct, Cout, h, w = case
    x = rnd(311, B, Ct, h, w)
    wt = rnd(312, Cout, Ct, 3, 3, scale=1.0 / np.sqrt(Ct * 9)) * torch.logspace(-3, 0, Cout).view(-1, 1, 1, 1) * 2.0
    pre = rnd(313, B, Cout, 4 * h, 4 * w)
    xh = hip.h2_pack(hip.to_device(x), hip.h2_empty(B, Ct, h, w))
    x22 = hip.h2_unpack(xh, hip.empty(B, Ct, h, w)).cpu()
    ref64 = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x22.double(), scale_factor=4, mode="nearest"), wt.double(), None, 1, 1)
    tol = 4e-6 * float(ref64.abs().max())
    pk = hip.pack_conv_up4_h2t(wt)
    out = hip.empty(B, Cout, 4 * h, 4 * w)
    out.fill_(float("nan"))
    hip.conv_up4_h2t(xh, pk, out)
    got = CPU.quads(out.cpu(), inverse=True)
    err = float((got.double() - ref64).abs().max())
    assert err <= tol, "conv_up4_h2t %s: max-abs %g > %g" % (case, err, tol)
    wide = hip.zeros(B, Cout + 8, 4 * h, 4 * w)                        # pre_add from a second buffer, output into a channel slice (quads 1 ..)
    pq = hip.to_device(CPU.quads(pre))
    hip.conv_up4_h2t(xh, pk, wide[:, 4:4 + Cout], pre_add=pq)
    got2 = CPU.quads(wide[:, 4:4 + Cout].cpu().contiguous(), inverse=True)
    assert float((got2.double() - (ref64 + pre.double())).abs().max()) <= tol + 1e-6, "conv_up4_h2t with pre_add %s" % (case,)
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0, "wrote outside its channel slice"
    hip.conv_up4_h2t(xh, pk, pq, pre_add=pq)                           # in place
    assert torch.equal(CPU.quads(pq.cpu(), inverse=True), got2), "conv_up4_h2t in place %s" % (case,)
    # compact form (nine class values per source pixel) + the conv over the channels at output resolution that adds it (conv_h2x `up4`): bit-identical
    # to that conv's quad-major result going through pre_add
    if Cout % 32 == 0:
        key = rnd(314, B, 32, 4 * h, 4 * w)
        kh = hip.h2_pack(hip.to_device(key), hip.h2_empty(B, 32, 4 * h, 4 * w))
        pk_key, epi = hip.pack_conv_x3(rnd(315, Cout, 32, 3, 3, scale=0.06), 1, lazy=True), hip.pack_epilogue(Cout, bias=rnd(316, Cout, scale=0.1))
        via_pre = hip.conv_h2x(kh, pk_key, hip.empty(B, Cout, 4 * h, 4 * w), epi=epi, y_fmt=1)
        hip.conv_up4_h2t(xh, pk, via_pre, pre_add=via_pre)
        comp = hip.conv_up4_h2t(xh, pk, hip.empty(B, 9 * Cout, h, w).fill_(float("nan")), compact=True)
        wide2 = hip.zeros(B, Cout + 8, 4 * h, 4 * w)
        hip.conv_h2x(kh, pk_key, wide2[:, 4:4 + Cout], epi=epi, y_fmt=1, up4=comp)
        assert torch.equal(wide2[:, 4:4 + Cout], via_pre), "compact + up4 differs from the pre_add path %s" % (case,)
        assert float(wide2[:, :4].abs().max()) == 0.0 and float(wide2[:, 4 + Cout:].abs().max()) == 0.0
    # the register-staged x4 kernel it replaces (NCHW): same arithmetic class
    old = hip.conv_up4_x3(hip.to_device(x), hip.pack_conv_up4_x3(wt), hip.empty(B, Cout, 4 * h, 4 * w))
    assert float((old.cpu().double() - ref64).abs().max()) <= 2.5 * tol + 1e-7


@pytest.mark.parametrize("case", [(2, 32, 0, 64, 16, 32), (1, 64, 16, 32, 21, 37), (3, 16, 32, 96, 5, 70), (2, 256, 64, 64, 40, 40), (1, 16, 16, 32, 9, 33)])
def test_conv_up2_h2t(hip, case):
    """bfsr_conv2d_up2_h2t: conv3x3(cat([key, nearest_up2(taps)])) evaluated at source resolution -- the taps parity-decomposed with pre-summed
    weights, the key channels (output resolution) as space-to-depth planes (bfsr_h2_pack_s2d) --, two-term fp16 split (three products), quad-major
    fp32 output with and without pre_add (also in place): against an fp64 conv of the SAME 22-bit inputs with the unsplit fp32 weights
    (SRFlowNet_arch.py:122-137: F.interpolate(..., mode='nearest') + cat + Conv2d).  Ragged tiles, several output-channel groups, with and without key channels, per-channel weight magnitudes over four decades."""
    B, Ct, Ck, Cout, h, w = case
    x, key = rnd(301, B, max(Ct, 1), h, w)[:, :Ct], rnd(304, B, max(Ck, 1), 2 * h, 2 * w)[:, :Ck]
    wt = rnd(302, Cout, Ct + Ck, 3, 3, scale=1.0 / np.sqrt((Ct + Ck) * 9)) * torch.logspace(-3, 0, Cout).view(-1, 1, 1, 1) * 2.0
    pre = rnd(303, B, Cout, 2 * h, 2 * w)
    xh = hip.h2_empty(B, Ct + 4 * Ck, h, w)
    if Ct:
        hip.h2_pack(hip.to_device(x), xh[:, :Ct // 8])
    if Ck:
        hip.h2_pack_s2d(hip.to_device(key), xh[:, Ct // 8:])
    x22 = hip.h2_unpack(xh, hip.empty(B, Ct + 4 * Ck, h, w)).cpu()
    k22 = torch.nn.functional.pixel_shuffle(x22[:, Ct:].reshape(B, 4, Ck, h, w).transpose(1, 2).reshape(B, 4 * Ck, h, w), 2) if Ck else key
    if Ck:
        assert float((k22 - key).abs().max()) <= 2.0 ** -21 * float(key.abs().max()), "h2_pack_s2d"
    cat = torch.cat([k22.double(), torch.nn.functional.interpolate(x22[:, :Ct].double(), scale_factor=2, mode="nearest")], 1) if Ct else k22.double()
    ref64 = torch.nn.functional.conv2d(cat, wt.double(), None, 1, 1)             # the reference's channel order: key first, then the taps
    tol = 4e-6 * float(ref64.abs().max())
    pk = hip.pack_conv_up2_h2t(wt[:, Ck:].contiguous(), wt[:, :Ck].contiguous() if Ck else None)
    out = hip.empty(B, Cout, 2 * h, 2 * w)
    out.fill_(float("nan"))
    hip.conv_up2_h2t(xh, pk, out)
    got = CPU.quads(out.cpu(), inverse=True)
    err = float((got.double() - ref64).abs().max())
    assert err <= tol, "conv_up2_h2t %s: max-abs %g > %g" % (case, err, tol)
    wide = hip.zeros(B, Cout + 8, 2 * h, 2 * w)                        # pre_add from a second buffer, output into a channel slice (quads 1 ..)
    pq = hip.to_device(CPU.quads(pre))
    hip.conv_up2_h2t(xh, pk, wide[:, 4:4 + Cout], pre_add=pq)
    got2 = CPU.quads(wide[:, 4:4 + Cout].cpu().contiguous(), inverse=True)
    assert float((got2.double() - (ref64 + pre.double())).abs().max()) <= tol + 1e-6, "conv_up2_h2t with pre_add %s" % (case,)
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + Cout:].abs().max()) == 0.0, "wrote outside its channel slice"
    hip.conv_up2_h2t(xh, pk, pq, pre_add=pq)                           # in place
    assert torch.equal(pq.cpu(), wide[:, 4:4 + Cout].cpu().contiguous()), "in-place pre_add differs from the two-buffer call"
    xc = CPU.h2_empty(B, Ct + 4 * Ck, h, w)
    if Ct:
        CPU.h2_pack(x, xc[:, :Ct // 8])
    if Ck:
        CPU.h2_pack_s2d(key, xc[:, Ct // 8:])
    cpu = CPU.conv_up2_h2t(xc, CPU.pack_conv_up2_h2t(wt[:, Ck:].contiguous(), wt[:, :Ck].contiguous() if Ck else None), torch.empty(B, Cout, 2 * h, 2 * w),
                           pre_add=CPU.quads(pre))
    close(pq, cpu, 2e-5, "conv_up2_h2t vs the CPU test double")


def test_coupling_pair_on_channel_slices_and_views(hip):
    """head -> tail with z / pre_aff / h_ft as channel slices of wider buffers, out of place into a slice (the engine runs in place)."""
    B, C, H, W = 2, 12, 20, 44
    zbuf, obuf = rnd(181, B, 20, H, W), torch.zeros(B, 16, H, W)
    prebuf, hfbuf = rnd(182, B, 3 * 64, H, W, scale=0.5), rnd(183, B, 3 * 24, H, W, scale=0.5)
    w0, w2 = rnd(163, 64, 6, 3, 3, scale=0.1), rnd(164, 64, 64, 1, 1, scale=0.1)
    s0, c0, s2, c2 = rnd(165, 64, scale=0.1), torch.exp(rnd(166, 64, scale=0.1)), rnd(167, 64, scale=0.1), torch.exp(rnd(168, 64, scale=0.1))
    w4, b4, ps = rnd(173, 12, 64, 3, 3, scale=0.02), rnd(174, 12, scale=0.2), torch.exp(rnd(175, 12, scale=0.2))
    hid = CPU.coupling_head(zbuf[:, :C], CPU.pack_coupling_head(w0, w2, s0, c0, s2, c2), prebuf[:, 64:128], torch.empty(B, 64, H, W))
    ref = CPU.coupling_tail(hid, CPU.pack_coupling_tail(w4, b4, ps), zbuf[:, :C], torch.empty(B, C, H, W), 1, h_ft=hfbuf[:, 24:48])
    od, zd = hip.to_device(obuf), hip.to_device(zbuf)
    hd = hip.coupling_head(zd[:, :C], hip.pack_coupling_head(w0, w2, s0, c0, s2, c2), hip.to_device(prebuf)[:, 64:128], hip.h2_empty(B, 64, H, W))
    hip.coupling_tail(hd, hip.pack_coupling_tail(w4, b4, ps), zd[:, :C], od[:, 2:2 + C], 1, h_ft=hip.to_device(hfbuf)[:, 24:48])
    close(od[:, 2:2 + C], ref, 3e-5, "coupling pair on slices")
    assert float(od[:, :2].abs().max()) == 0.0 and float(od[:, 2 + C:].abs().max()) == 0.0, "wrote outside its channel slice"
    assert torch.equal(zd.cpu(), zbuf), "the input must be left alone"


@pytest.mark.parametrize("C", [12, 24, 96])
@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("hw", [(16, 24), (7, 9)])
def test_flow_pointwise(hip, C, reverse, hw):
    H, W = hw
    B = 2
    z = rnd(30 + C, B, C, H, W)
    ha = rnd(31 + C, B, 2 * (C - C // 2), H, W)
    hf = rnd(32 + C, B, 2 * C, H, W)
    q, _ = np.linalg.qr(np.random.Generator(np.random.PCG64(C)).standard_normal((C, C)))
    Wm = torch.from_numpy(q.astype(np.float32)) + rnd(33, C, C, scale=0.02)
    bias, es = rnd(34, C, scale=0.2), torch.exp(rnd(35, C, scale=0.2))
    combos = [dict(h_aff=ha, h_ft=hf, w=Wm, an_bias=bias, an_escale=es), dict(w=Wm, an_bias=bias, an_escale=es),
              dict(h_aff=ha), dict(h_ft=hf, an_bias=bias, an_escale=es)]
    for kw in combos:
        ref = CPU.flow_pointwise(z, torch.empty_like(z), reverse, **{k: (v.reshape(-1) if k == "w" else v) for k, v in kw.items()})
        dkw = {k: (hip.to_device(v) if v.dim() == 4 else hip.vec(v)) for k, v in kw.items()}
        zd = hip.to_device(z)
        out = hip.flow_pointwise(zd, zd, reverse, **dkw)          # in place
        close(out, ref, 1e-5, "flow_pointwise C=%d rev=%d %s" % (C, reverse, sorted(kw)))
        if "w" in kw:                                             # MFMA path (taken for C=96 when wt is given)
            zd = hip.to_device(z)
            out = hip.flow_pointwise(zd, zd, reverse, wt=hip.vec(Wm.t().contiguous()), **dkw)
            close(out, ref, 1e-5, "flow_pointwise+wt C=%d rev=%d %s" % (C, reverse, sorted(kw)))


def test_flow_pointwise_on_channel_slice(hip):
    # z is the first 12 channels of a wider buffer (batch stride != C*H*W)
    B, C, H, W = 2, 12, 8, 12
    buf = rnd(40, B, 20, H, W)
    hf = rnd(41, B, 24, H, W)
    ref = buf.clone()
    CPU.flow_pointwise(buf[:, :C].clone(), ref[:, :C], 1, h_ft=hf)
    bd = hip.to_device(buf)
    hip.flow_pointwise(bd[:, :C], bd[:, :C], 1, h_ft=hip.to_device(hf))
    close(bd, ref, 1e-5, "flow slice")


def test_squeeze_unsqueeze(hip):
    x = rnd(50, 2, 6, 10, 14)
    y = hip.squeeze2d(hip.to_device(x), hip.empty(2, 24, 5, 7))
    assert torch.equal(y.cpu(), CPU.squeeze2d(x, torch.empty(2, 24, 5, 7)))
    back = hip.unsqueeze2d(y, hip.empty(2, 6, 10, 14))
    assert torch.equal(back.cpu(), x)
    # strided source (first 6 of 12 channels) as after a Split2d
    big = rnd(51, 2, 12, 8, 8)
    y2 = hip.squeeze2d(hip.to_device(big)[:, :6], hip.empty(2, 24, 4, 4))
    assert torch.equal(y2.cpu(), CPU.squeeze2d(big[:, :6], torch.empty(2, 24, 4, 4)))


def test_split2d_standardize(hip):
    h, s = rnd(60, 2, 12, 9, 11, scale=0.3), rnd(61, 2, 6, 9, 11)
    for rev in (0, 1):
        ref = CPU.split2d(h, s, torch.empty_like(s), rev)
        out = hip.split2d(hip.to_device(h), hip.to_device(s), hip.empty(2, 6, 9, 11), rev)
        close(out, ref, 1e-6, "split2d rev=%d" % rev)
    for C in (6, 96):
        x = rnd(62 + C, 2, C, 5, 7, scale=2.0) + 0.3
        ref = CPU.standardize(x, torch.empty_like(x))
        out = hip.standardize(hip.to_device(x), hip.empty(*x.shape))
        close(out, ref, 2e-6, "standardize C=%d" % C)


RESIZE = [
    (0, (8, 10), (16, 20), None), (0, (16, 20), (8, 10), None), (0, (7, 9), (14, 18), None),
    (1, (8, 10), (32, 40), 0.25), (1, (16, 20), (8, 10), None), (1, (9, 7), (20, 15), None),
    (2, (5, 6), (10, 12), None), (2, (1, 3), (2, 6), None),
]


@pytest.mark.parametrize("case", RESIZE)
def test_resize_matches_torch_interpolate(hip, case):
    import torch.nn.functional as F
    mode, (ih, iw), (oh, ow), r = case
    x = rnd(70, 2, 5, ih, iw)
    if mode == 0:
        ref = F.interpolate(x, (oh, ow))
        rh, rw = ih / oh, iw / ow
    elif mode == 1:
        if r is not None:
            ref = F.interpolate(x, scale_factor=1 / r, mode="bilinear", align_corners=False)
            rh = rw = r
        else:
            ref = F.interpolate(x, (oh, ow), mode="bilinear", align_corners=False)
            rh, rw = ih / oh, iw / ow
    else:
        ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        rh = (ih - 1) / (oh - 1) if oh > 1 else 0.0
        rw = (iw - 1) / (ow - 1) if ow > 1 else 0.0
    out = hip.resize(hip.to_device(x), hip.empty(2, 5, oh, ow), mode, rh, rw)
    close(out, ref, 2e-6, "resize %s" % (case,))


def test_resize_padded_window_maxpool_clamp(hip):
    import torch.nn.functional as F
    x = rnd(80, 1, 4, 5, 6)
    ref = F.pad(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), [0, 1, 1, 0])
    out = hip.resize(hip.to_device(x), hip.empty(1, 4, 11, 13), 2, 4 / 9, 5 / 11, window=(1, 0, 10, 12))
    close(out, ref, 2e-6, "resize window")
    y = rnd(81, 2, 3, 9, 11)
    assert torch.equal(hip.maxpool2(hip.to_device(y), hip.empty(2, 3, 4, 5)).cpu(), F.max_pool2d(y, 2))
    r = rnd(82, 2, 3, 9, 11)
    ref = torch.clamp(0.5 * y + 0.5 + r, 0, 1)
    close(hip.axpb_clamp(hip.to_device(y), hip.empty(2, 3, 9, 11), 0.5, 0.5, 0.0, 1.0, r=hip.to_device(r)), ref, 1e-7, "axpb")


@pytest.mark.parametrize("mode,window", [(0, None), (1, None), (2, None), (2, (1, 2, 38, 76))])
def test_resize_and_clamp_vector_paths_equal_scalar_paths(hip, mode, window):
    """resize4_kernel / axpb_clamp4_kernel (four outputs per thread, 16-byte stores: row length a multiple of four, aligned views) against the scalar
    kernels (forced by an output view that starts one float into a wider buffer): bit-identical."""
    x = hip.to_device(rnd(85, 2, 3, 20, 39))
    oh, ow = 40, 80
    rh, rw = (20 / oh, 39 / ow) if mode < 2 else (19 / (oh - 1), 38 / (ow - 1))
    kw = dict(window=window) if window else {}
    fast = hip.resize(x, hip.empty(2, 3, oh, ow), mode, rh, rw, **kw)
    slab = hip.zeros(2, 3 * oh * ow + 4)
    slow = hip.resize(x, slab[:, 1:1 + 3 * oh * ow].view(2, 3, oh, ow), mode, rh, rw, **kw)
    assert torch.equal(fast, slow), "resize: vector and scalar kernels differ"
    a, r = hip.to_device(rnd(86, 2, 3, oh, ow)), hip.to_device(rnd(87, 2, 3, oh, ow))
    fast = hip.axpb_clamp(a, hip.empty(2, 3, oh, ow), 0.5, 0.5, 0.0, 1.0, r=r)
    slab2 = hip.zeros(2, 3 * oh * ow + 4)
    slow = hip.axpb_clamp(a, slab2[:, 1:1 + 3 * oh * ow].view(2, 3, oh, ow), 0.5, 0.5, 0.0, 1.0, r=r)
    assert torch.equal(fast, slow), "axpb_clamp: vector and scalar kernels differ"


@pytest.mark.parametrize("case", [(2, 6, 64, 64, 40, 70), (1, 48, 64, 64, 19, 33), (1, 12, 64, 64, 64, 64), (1, 70, 50, 40, 9, 31)])
def test_conv_fused_1x1_second_stage(hip, case):
    """3x3 conv (+pre_add, ActNorm affine, ReLU) with the following 1x1 conv (+affine, ReLU) fused in-kernel."""
    B, Cin, Cmid, C2, H, W = case
    x, w = rnd(90, B, Cin, H, W), rnd(91, Cmid, Cin, 3, 3, scale=0.1)
    w2 = rnd(92, C2, Cmid, 1, 1, scale=0.15)
    pre = rnd(93, B, Cmid, H, W)
    sh1, sc1 = rnd(94, Cmid, scale=0.2), torch.exp(rnd(95, Cmid, scale=0.2))
    sh2, sc2 = rnd(96, C2, scale=0.2), torch.exp(rnd(97, C2, scale=0.2))
    ref = CPU.conv(x, CPU.pack_conv(w, 2), torch.empty(B, C2, H, W), pre_add=pre, aff_shift=sh1, aff_scale=sc1, act=1,
                   stage2=(CPU.pack_conv(w2, 2), CPU.pack_epilogue(C2, aff_shift=sh2, aff_scale=sc2), 1))
    out = hip.conv(hip.to_device(x), hip.pack_conv(w, 2), hip.empty(B, C2, H, W), pre_add=hip.to_device(pre),
                   aff_shift=sh1, aff_scale=sc1, act=1,
                   stage2=(hip.pack_conv(w2, 2), hip.pack_epilogue(C2, aff_shift=sh2, aff_scale=sc2), 1))
    close(out, ref, 2e-5, "fused 3x3+1x1 %s" % (case,))


@pytest.mark.parametrize("case", [(2, 70, 64, 9, 21, 2), (1, 256, 128, 12, 40, 2), (1, 16, 24, 5, 33, 1), (2, 64, 64, 16, 32, 2)])
def test_conv_up2_parity_decomposition(hip, case):
    """conv3x3(nearest_up2(x)) computed on the source grid with pre-summed weights == the materialised computation."""
    B, Cin, Cout, h, w_, mt = case
    x, w = rnd(110, B, Cin, h, w_), rnd(111, Cout, Cin, 3, 3, scale=1.0 / np.sqrt(Cin * 9))
    pre = rnd(112, B, Cout, 2 * h, 2 * w_)
    sh, sc = rnd(113, Cout, scale=0.2), torch.exp(rnd(114, Cout, scale=0.2))
    ref = CPU.conv_up2(x, CPU.pack_conv_up2(w, mt), torch.empty(B, Cout, 2 * h, 2 * w_))
    out = hip.conv_up2(hip.to_device(x), hip.pack_conv_up2(w, mt), hip.empty(B, Cout, 2 * h, 2 * w_))
    close(out, ref, 2e-5, "conv_up2 plain %s" % (case,))
    epi_c = CPU.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc)
    ref = CPU.conv_up2(x, CPU.pack_conv_up2(w, mt), torch.empty(B, Cout, 2 * h, 2 * w_), epi=epi_c, pre_add=pre, act=1)
    out = hip.conv_up2(hip.to_device(x), hip.pack_conv_up2(w, mt), hip.empty(B, Cout, 2 * h, 2 * w_),
                       epi=hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc), pre_add=hip.to_device(pre), act=1)
    close(out, ref, 2e-5, "conv_up2 epilogue %s" % (case,))


@pytest.mark.parametrize("case", [(2, 70, 64, 64, 9, 21), (1, 256, 64, 128, 12, 40), (1, 24, 10, 40, 7, 35)])
def test_conv_up2_with_key_channels(hip, case):
    """conv3x3(cat[key @ output res, nearest_up2(taps)]) in one kernel."""
    B, Ct, Ck, Cout, h, w_ = case
    taps, key = rnd(120, B, Ct, h, w_), rnd(121, B, Ck, 2 * h, 2 * w_)
    w = rnd(122, Cout, Ck + Ct, 3, 3, scale=1.0 / np.sqrt((Ck + Ct) * 9))
    sh, sc = rnd(123, Cout, scale=0.2), torch.exp(rnd(124, Cout, scale=0.2))
    wk, wt = w[:, :Ck].contiguous(), w[:, Ck:].contiguous()
    ref = CPU.conv_up2(taps, CPU.pack_conv_up2(wt, 2), torch.empty(B, Cout, 2 * h, 2 * w_),
                       epi=CPU.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc), act=1, key=(key, CPU.pack_conv(wk, 2)))
    out = hip.conv_up2(hip.to_device(taps), hip.pack_conv_up2(wt, 2), hip.empty(B, Cout, 2 * h, 2 * w_),
                       epi=hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc), act=1,
                       key=(hip.to_device(key), hip.pack_conv(wk, 2)))
    close(out, ref, 2e-5, "conv_up2+key %s" % (case,))


def test_bad_arguments_fail_loudly(hip):
    """Error behaviour of the boundary: shape / layout mistakes raise, unsupported configurations return an error code."""
    x = hip.to_device(rnd(130, 1, 8, 8, 8))
    pw = hip.pack_conv(rnd(131, 16, 8, 3, 3))
    with pytest.raises(ValueError):
        hip.conv(x, pw, hip.empty(1, 16, 9, 8))                              # spatial mismatch
    with pytest.raises(ValueError):
        hip.conv(hip.to_device(rnd(132, 1, 7, 8, 8)), pw, hip.empty(1, 16, 8, 8))   # channel mismatch
    with pytest.raises(ValueError):
        hip.conv(x.permute(0, 1, 3, 2), pw, hip.empty(1, 16, 8, 8))          # not plane-contiguous
    with pytest.raises(RuntimeError):
        hip.flow_pointwise(hip.to_device(rnd(133, 1, 7, 4, 4)), hip.empty(1, 7, 4, 4), 1)   # unsupported channel count
    with pytest.raises(RuntimeError):
        hip.squeeze2d(hip.to_device(rnd(134, 1, 3, 5, 6)), hip.empty(1, 12, 2, 3))     # odd height


def test_conv_output_larger_than_2gib(hip):
    """Outputs are written with 64-bit addressing: a > 2 GiB result (full-image hoisted tensors) is fine."""
    B, Cin, Cout, H, W = 1, 8, 1024, 736, 736            # 1024*736*736*4 B = 2.22 GB
    x, w = rnd(140, B, Cin, H, W), rnd(141, Cout, Cin, 3, 3, scale=0.1)
    out = hip.conv(hip.to_device(x), hip.pack_conv(w, 2), hip.empty(B, Cout, H, W))
    for sl in (slice(0, 4), slice(1020, 1024)):
        ref = torch.nn.functional.conv2d(x, w[sl], None, 1, 1)
        close(out[:, sl], ref, 2e-5, "conv >2GiB out, channels %s" % (sl,))
    del out
    torch.cuda.empty_cache()


F16_CASES = [(2, 64, 32, 40, 72, 3, 1), (1, 192, 64, 20, 36, 3, 2), (1, 70, 50, 17, 65, 3, 2), (1, 3, 64, 16, 16, 3, 2),
             (1, 1024, 256, 9, 40, 1, 2), (2, 100, 27, 9, 31, 1, 1), (2, 64, 64, 130, 130, 3, 2)]


@pytest.mark.parametrize("case", F16_CASES)
def test_conv_f16_path(hip, case):
    """fp16-MFMA conv == fp32 conv of the fp16-rounded operands (fp32 accumulation), incl. epilogue + residuals."""
    B, Cin, Cout, H, W, KS, mt = case
    x = rnd(160, B, Cin, H, W)
    w = rnd(161, Cout, Cin, KS, KS, scale=1.0 / np.sqrt(Cin * KS * KS))
    b, r1 = rnd(162, Cout, scale=0.2), rnd(163, B, Cout, H, W)
    ref = CPU.conv_f16(x, CPU.pack_conv_f16(w, mt), torch.empty(B, Cout, H, W), bias=b, act=2, res1=r1, alpha1=0.2)
    out = hip.conv_f16(hip.to_device(x), hip.pack_conv_f16(w, mt), hip.empty(B, Cout, H, W),
                       epi=hip.pack_epilogue(Cout, bias=b), act=2, res1=hip.to_device(r1), alpha1=0.2)
    close(out, ref, 2e-5, "conv_f16 %s" % (case,))
    full = CPU.conv(x, CPU.pack_conv(w, mt), torch.empty(B, Cout, H, W), bias=b, act=2, res1=r1, alpha1=0.2)
    dev = (out.cpu() - full).abs().max().item()
    assert dev < 2e-2, "fp16 path deviates %.3e from fp32" % dev          # reported, not a 1e-4 claim


X3_CASES = [(2, 64, 32, 40, 72, 3, 1), (1, 192, 64, 20, 36, 3, 2), (1, 70, 50, 17, 65, 3, 2), (1, 3, 64, 16, 16, 3, 2),
            (1, 1024, 256, 9, 40, 1, 2), (2, 100, 27, 9, 31, 1, 1), (1, 320, 128, 48, 64, 3, 2)]


@pytest.mark.parametrize("case", X3_CASES)
def test_conv_bf16x3_is_fp32_accurate(hips, case):
    """The split convs (three-term bf16 / two-term fp16): error against an fp64 conv is at the level of the native fp32 MFMA kernel
    (and of the CPU fp32 conv the oracle uses) -- i.e. they are fp32 convs, not reduced-precision ones.  The fp16 pair carries 22
    instead of 24 significant bits per operand: its bound is 4x instead of 2x the fp32 kernels' error."""
    hip = hips
    slack = 2.0 if hip.split == "bf16x3" else 4.0
    B, Cin, Cout, H, W, KS, mt = case
    x = rnd(170, B, Cin, H, W) * 3.0
    x[:, :, ::3] *= 1e-3                                        # wide dynamic range
    w = rnd(171, Cout, Cin, KS, KS, scale=1.0 / np.sqrt(Cin * KS * KS))
    b, r1 = rnd(172, Cout, scale=0.2), rnd(173, B, Cout, H, W)
    truth = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=KS // 2)
    truth = torch.nn.functional.leaky_relu(truth, 0.2) * 0.2 + r1.double()
    kw = dict(act=2, res1=hip.to_device(r1), alpha1=0.2)
    xd = hip.to_device(x)
    o32 = hip.conv(xd, hip.pack_conv(w, mt), hip.empty(B, Cout, H, W), epi=hip.pack_epilogue(Cout, bias=b), **kw)
    for tune in ([0] if KS == 1 else [200, 400]):
        ox3 = hip.conv_x3(xd, hip.pack_conv_x3(w, mt), hip.empty(B, Cout, H, W), epi=hip.pack_epilogue(Cout, bias=b), tune=tune, **kw)
        e32 = (o32.cpu().double() - truth).abs().max().item()
        ex3 = (ox3.cpu().double() - truth).abs().max().item()
        cpu = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, w, b, padding=KS // 2), 0.2) * 0.2 + r1
        ecpu = (cpu.double() - truth).abs().max().item()
        print("case %s tune %d: max-abs err vs fp64: %s %.2e, fp32 MFMA %.2e, torch CPU fp32 %.2e" % (case, tune, hip.split, ex3, e32, ecpu))
        assert ex3 <= slack * max(e32, ecpu) + 1e-7, (ex3, e32, ecpu)
        close(ox3, o32.cpu(), 1e-5, "bf16x3 vs fp32 %s" % (case,))


@pytest.mark.parametrize("case", [(2, 70, 64, 64, 9, 21), (1, 256, 64, 128, 12, 40), (1, 24, 10, 40, 7, 35), (1, 48, 16, 96, 33, 50)])
@pytest.mark.parametrize("tune", [402, 401, 801])
def test_conv_up2_bf16x3_with_key_channels(hips, case, tune):
    """The 3xBF16 pair (plain conv over the key channels -> taps kernel with pre_add aliasing the output) == the conv over
    cat[key, nearest_up2(taps)], at fp32 accuracy."""
    hip = hips
    B, Ct, Ck, Cout, h, w_ = case
    taps, key = rnd(180, B, Ct, h, w_), rnd(181, B, Ck, 2 * h, 2 * w_)
    w = rnd(182, Cout, Ck + Ct, 3, 3, scale=1.0 / np.sqrt((Ck + Ct) * 9))
    sh, sc = rnd(183, Cout, scale=0.2), torch.exp(rnd(184, Cout, scale=0.2))
    wk, wt = w[:, :Ck].contiguous(), w[:, Ck:].contiguous()
    xin = torch.cat([key, torch.nn.functional.interpolate(taps, scale_factor=2, mode="nearest")], 1)
    truth = torch.relu((torch.nn.functional.conv2d(xin.double(), w.double(), padding=1) + sh.double().view(1, -1, 1, 1))
                       * sc.double().view(1, -1, 1, 1))
    out = hip.empty(B, Cout, 2 * h, 2 * w_)
    hip.conv_x3(hip.to_device(key), hip.pack_conv_x3(wk, 2), out)
    hip.conv_up2_x3(hip.to_device(taps), hip.pack_conv_up2_x3(wt), out, epi=hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc),
                    act=1, pre_add=out, tune=tune)
    o32 = hip.conv_up2(hip.to_device(taps), hip.pack_conv_up2(wt, 2), hip.empty(B, Cout, 2 * h, 2 * w_),
                       epi=hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc), act=1, key=(hip.to_device(key), hip.pack_conv(wk, 2)))
    ex3 = (out.cpu().double() - truth).abs().max().item()
    e32 = (o32.cpu().double() - truth).abs().max().item()
    assert ex3 <= (2.0 if hip.split == "bf16x3" else 4.0) * e32 + 1e-7, (ex3, e32)
    close(out, o32.cpu(), 1e-5, "conv_up2_x3 %s" % (case,))


@pytest.mark.parametrize("case", [(2, 64, 64, 64, 9, 21), (1, 256, 64, 128, 12, 40), (1, 48, 16, 96, 33, 50)])
def test_quad_major_outputs_are_the_same_values(hip, case):
    """y_fmt=1 of the split-conv kernels (conv_x3, conv_x3s on h2 tensors, conv_up2_x3 with pre_add aliasing the output): the values of the
    NCHW form, bit for bit, laid out [B][Cout/4][H][W][4] -- the private layout coupling_head (pre_aff) and coupling_tail (h_ft) read."""
    B, Ct, Ck, Cout, h, w_ = case
    taps, key = rnd(180, B, Ct, h, w_), rnd(181, B, Ck, 2 * h, 2 * w_)
    w = rnd(182, Cout, Ck + Ct, 3, 3, scale=1.0 / np.sqrt((Ck + Ct) * 9))
    sh, sc = rnd(183, Cout, scale=0.2), torch.exp(rnd(184, Cout, scale=0.2))
    wk, wt = w[:, :Ck].contiguous(), w[:, Ck:].contiguous()
    epi = hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc)
    pk, pt = hip.pack_conv_x3(wk, 2), hip.pack_conv_up2_x3(wt)
    kd, td = hip.to_device(key), hip.to_device(taps)
    ref, out = hip.empty(B, Cout, 2 * h, 2 * w_), hip.empty(B, Cout, 2 * h, 2 * w_)
    hip.conv_x3(kd, pk, ref)
    hip.conv_x3(kd, pk, out, y_fmt=1)
    assert torch.equal(CPU.quads(out.cpu(), inverse=True), ref.cpu()), "conv_x3 y_fmt=1"
    hip.conv_up2_x3(td, pt, ref, epi=epi, act=1, pre_add=ref)
    hip.conv_up2_x3(td, pt, out, epi=epi, act=1, pre_add=out, y_fmt=1)
    assert torch.equal(CPU.quads(out.cpu(), inverse=True), ref.cpu()), "conv_up2_x3 y_fmt=1 (pre_add aliasing the output)"
    # the LDS-DMA kernel over an h2 copy of the key channels (the level-1 key convs of the default path)
    kh = hip.h2_pack(kd, hip.h2_empty(B, Ck, 2 * h, 2 * w_))
    p1 = hip.pack_conv_x3(wk, 1)
    hip.conv_x3s(kh, p1, ref, epi=epi, act=1)
    hip.conv_x3s(kh, p1, out, epi=epi, act=1, y_fmt=1)
    assert torch.equal(CPU.quads(out.cpu(), inverse=True), ref.cpu()), "conv_h2x y_fmt=1"
    # channel-slice views of a wider buffer keep their meaning: slice [a:b] of the NCHW view = quads [a/4:b/4] of the quad-major one
    wide_r, wide_q = hip.zeros(B, Cout + 8, 2 * h, 2 * w_), hip.zeros(B, Cout + 8, 2 * h, 2 * w_)
    hip.conv_x3(kd, pk, wide_r[:, 4:4 + Cout])
    hip.conv_x3(kd, pk, wide_q[:, 4:4 + Cout], y_fmt=1)
    assert torch.equal(CPU.quads(wide_q[:, 4:4 + Cout].cpu().contiguous(), inverse=True), wide_r[:, 4:4 + Cout].cpu())
    assert float(wide_q[:, :4].abs().max()) == 0.0 and float(wide_q[:, 4 + Cout:].abs().max()) == 0.0, "wrote outside its channel slice"


def test_likelihood_reductions(hip):
    """bfsr_logscale_sum / bfsr_gaussian_logp: per-sample float64 sums == torch float64 reference, coef and accumulation."""
    h = rnd(190, 3, 24, 37, 53)
    acc = hip.zeros_f64(3)
    hip.logscale_sum(hip.to_device(h), acc, 1.0)
    ref = torch.log(torch.sigmoid(h[:, 1::2].double() + 2.0) + 1e-4).sum(dim=(1, 2, 3))
    assert ((acc.cpu() - ref).abs() / ref.abs()).max() < 1e-6
    hip.logscale_sum(hip.to_device(h)[:, 4:12], acc, -2.0)             # channel-slice view, accumulate on top
    ref2 = ref - 2.0 * torch.log(torch.sigmoid(h[:, 5:12:2].double() + 2.0) + 1e-4).sum(dim=(1, 2, 3))
    assert ((acc.cpu() - ref2).abs() / ref2.abs()).max() < 1e-6
    x, hh = rnd(191, 3, 6, 37, 53), rnd(192, 3, 12, 37, 53, scale=0.3)
    l2pi = float(np.log(2 * np.pi))
    a0 = hip.gaussian_logp(hip.to_device(x), hip.zeros_f64(3))
    r0 = (-0.5 * (x.double() ** 2 + l2pi)).sum(dim=(1, 2, 3))
    assert ((a0.cpu() - r0).abs() / r0.abs()).max() < 1e-6
    a1 = hip.gaussian_logp(hip.to_device(x), hip.zeros_f64(3), h=hip.to_device(hh), coef=-1.0)
    mean, logs = hh[:, 0::2].double(), hh[:, 1::2].double()
    r1 = -(-0.5 * (logs * 2 + (x.double() - mean) ** 2 / torch.exp(logs * 2) + l2pi)).sum(dim=(1, 2, 3))
    assert ((a1.cpu() - r1).abs() / r1.abs()).max() < 1e-6


@pytest.mark.parametrize("case", [(2, 70, 64, 64, 9, 21), (1, 256, 64, 96, 12, 40), (1, 24, 10, 40, 7, 35)])
def test_conv_up4_bf16x3_with_key_channels(hips, case):
    """x4 parity kernel (25 pre-summed matrices, 3 row-class launches) + key conv == conv over cat[key, nearest_up4(taps)]."""
    hip = hips
    B, Ct, Ck, Cout, h, w_ = case
    taps, key = rnd(200, B, Ct, h, w_), rnd(201, B, Ck, 4 * h, 4 * w_)
    w = rnd(202, Cout, Ck + Ct, 3, 3, scale=1.0 / np.sqrt((Ck + Ct) * 9))
    sh, sc = rnd(203, Cout, scale=0.2), torch.exp(rnd(204, Cout, scale=0.2))
    wk, wt = w[:, :Ck].contiguous(), w[:, Ck:].contiguous()
    xin = torch.cat([key, torch.nn.functional.interpolate(taps, scale_factor=4, mode="nearest")], 1)
    truth = torch.relu((torch.nn.functional.conv2d(xin.double(), w.double(), padding=1) + sh.double().view(1, -1, 1, 1))
                       * sc.double().view(1, -1, 1, 1))
    out = hip.empty(B, Cout, 4 * h, 4 * w_)
    hip.conv_x3(hip.to_device(key), hip.pack_conv_x3(wk, 2), out)
    hip.conv_up4_x3(hip.to_device(taps), hip.pack_conv_up4_x3(wt), out, epi=hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc),
                    act=1, pre_add=out)
    ref32 = torch.relu((torch.nn.functional.conv2d(xin, w, padding=1) + sh.view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1))
    o32 = hip.conv(hip.to_device(xin), hip.pack_conv(w, 2), hip.empty(B, Cout, 4 * h, 4 * w_),
                   epi=hip.pack_epilogue(Cout, aff_shift=sh, aff_scale=sc), act=1)        # native fp32 MFMA on the materialised input
    ex3 = (out.cpu().double() - truth).abs().max().item()
    e32 = max((ref32.double() - truth).abs().max().item(), (o32.cpu().double() - truth).abs().max().item())
    assert ex3 <= (2.0 if hip.split == "bf16x3" else 4.0) * e32 + 1e-7, (ex3, e32)
    close(out, ref32, 1e-5, "conv_up4_x3 %s" % (case,))


@pytest.mark.parametrize("case", [(2, 1024, 256, 9, 40), (1, 256, 540, 17, 33), (3, 100, 130, 5, 7), (1, 256, 256, 64, 64)])
@pytest.mark.parametrize("x3", [True, False])
def test_wide_conv1x1(hip, case, x3):
    """conv1x1.hip (128 px x 256 couts per workgroup): x3 mode is fp32-accurate, f16 mode == fp16-rounded operands."""
    B, Cin, Cout, H, W = case
    x = rnd(210, B, Cin, H, W)
    w = rnd(211, Cout, Cin, 1, 1, scale=1.0 / np.sqrt(Cin))
    b, r1 = rnd(212, Cout, scale=0.2), rnd(213, B, Cout, H, W)
    out = hip.conv1x1(hip.to_device(x), hip.pack_conv1x1(w, x3=x3), hip.empty(B, Cout, H, W), x3=x3,
                      epi=hip.pack_epilogue(Cout, bias=b), act=1, res1=hip.to_device(r1), alpha1=0.5)
    ref = CPU.conv1x1(x, CPU.pack_conv1x1(w, x3=x3), torch.empty(B, Cout, H, W), x3=x3, bias=b, act=1, res1=r1, alpha1=0.5)
    close(out, ref, 1e-5 if x3 else 2e-5, "conv1x1 %s x3=%s" % (case, x3))
    if x3:
        truth = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double())) * 0.5 + r1.double()
        o32 = hip.conv(hip.to_device(x), hip.pack_conv(w, 2), hip.empty(B, Cout, H, W), epi=hip.pack_epilogue(Cout, bias=b), act=1,
                       res1=hip.to_device(r1), alpha1=0.5)                      # native fp32 MFMA kernel
        e = (out.cpu().double() - truth).abs().max().item()
        e32 = max((ref.double() - truth).abs().max().item(), (o32.cpu().double() - truth).abs().max().item())
        assert e <= 2.0 * e32 + 1e-7, (e, e32)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(2, 16, 37, 45), (1, 64, 64, 96), (3, 8, 17, 17)])
def test_h2_glue_equals_launches(hip, B, C, H, W):
    """Round 6: maxpool2_h2 (= h2_unpack + maxpool2 [+ h2_pack]) and resize_h2 (= resize + h2_pack, bilinear align_corners True with the Up layer's pad window, and
    align_corners False) of the learned priors' h2 levels (models/unet.py:58-98): the same bits as the launches they replace, odd sizes included."""
    from bfsr_amd.ops import MODE_BILINEAR, MODE_BILINEAR_AC
    x = hip.to_device(rnd(900 + W, B, C + 8, H, W))
    xh = hip.h2_pack(x, hip.h2_empty(B, C + 8, H, W))[:, 1:]                                   # a channel-slice view, like the skip half of a concat buffer
    un = hip.h2_unpack(xh, hip.empty(B, C, H, W))
    pool = hip.maxpool2(un, hip.empty(B, C, H // 2, W // 2))
    pool_h = hip.h2_pack(pool, hip.h2_empty(B, C, H // 2, W // 2))
    both_h, both_f = hip.h2_empty(B, C + 8, H // 2, W // 2), hip.empty(B, C, H // 2, W // 2).fill_(float("nan"))
    hip.maxpool2_h2(xh, out_h2=both_h[:, 1:], out_f32=both_f)
    assert torch.equal(both_f, pool) and torch.equal(both_h[:, 1:], pool_h)
    only_f = hip.maxpool2_h2(xh, out_f32=hip.empty(B, C, H // 2, W // 2))
    assert torch.equal(only_f, pool)
    # Up: bilinear x2 (align_corners True) into a window of a larger image (the pad of unet.py:88-92), then joined to the concat buffer as h2
    OH, OW = 2 * H + 1, 2 * W + 3
    r_h, r_w = float(H - 1) / float(2 * H - 1), float(W - 1) / float(2 * W - 1)
    win = (0, 1, 2 * H, 2 * W)
    up = hip.resize(un, hip.empty(B, C, OH, OW), MODE_BILINEAR_AC, r_h, r_w, window=win)
    up_h = hip.h2_pack(up, hip.h2_empty(B, C, OH, OW))
    cat = hip.h2_empty(B, C + 16, OH, OW)
    hip.resize_h2(un, cat[:, 2:], MODE_BILINEAR_AC, r_h, r_w, window=win)
    assert torch.equal(cat[:, 2:], up_h)
    oh, ow = 3 * H - 1, 3 * W + 2
    a = hip.h2_pack(hip.resize(un, hip.empty(B, C, oh, ow), MODE_BILINEAR, float(H) / oh, float(W) / ow), hip.h2_empty(B, C, oh, ow))
    b = hip.resize_h2(un, hip.h2_empty(B, C, oh, ow), MODE_BILINEAR, float(H) / oh, float(W) / ow)
    assert torch.equal(a, b)
    # pad + pack (the 6- / 27-channel latents entering the priors' DenseBlock_5C): = copy into a zero-initialised tensor + h2_pack
    xs = un[:, :C - 5].contiguous()
    xs[0, 0, 0, 0] = -0.0
    padded = hip.zeros(B, C, H, W)
    hip.axpb_clamp(xs, padded[:, :C - 5])
    ref_p = hip.h2_pack(padded, hip.h2_empty(B, C, H, W))
    got_p = hip.h2_pack_pad(xs, hip.h2_empty(B, C, H, W))
    assert torch.equal(got_p.view(torch.int16), ref_p.view(torch.int16))
