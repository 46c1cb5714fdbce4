"""-m gpu: the coupled FlowStep of the wide level (C = 96; coupling_wide.hip) through the C ABI -- against the CPU semantics (tests/cpu_ops.py) and,
bit for bit where the arithmetic is the same, against the launch sequence it replaces (h2_pack, conv_h2x, the 1x1-only coupling_head, conv_h2x,
flow_pointwise).  FlowAffineCouplingsAblation.py:57-97, FlowStep.py:88-129."""
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


def close(a, b, tol, what=""):
    a = a.detach().cpu()
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert not torch.isnan(a).any(), what + ": NaN in output"
    assert err <= tol * ref, "%s: max-abs %.3e > %.1e * %.2f" % (what, err, tol, ref)
    return err


def _h2_values(t):
    return sum(CPU._h2_planes(t.cpu()))


def _step_params(seed):
    w0, w2 = rnd(seed + 1, 64, 48, 3, 3, scale=0.05), rnd(seed + 2, 64, 64, 1, 1, scale=0.1)
    s0, c0 = rnd(seed + 3, 64, scale=0.1), torch.exp(rnd(seed + 4, 64, scale=0.1))
    s2, c2 = rnd(seed + 5, 64, scale=0.1), torch.exp(rnd(seed + 6, 64, scale=0.1))
    w4, b4, ps = rnd(seed + 7, 96, 64, 3, 3, scale=0.02), rnd(seed + 8, 96, scale=0.2), torch.exp(rnd(seed + 9, 96, scale=0.2))
    return w0, w2, s0, c0, s2, c2, w4, b4, ps


HW = [(16, 40), (9, 33), (80, 80), (5, 3), (37, 91), (8, 32)]


@pytest.mark.parametrize("hw", HW)
def test_wide_head(hip, hw):
    """hid = relu(AN2(W2 . relu(AN0(conv3x3(z1) + pre)))) with z1 and pre handed over as h2 tensors: against the CPU semantics on the SAME
    (hi + lo) input values, and bit-identical to conv_h2x + the 1x1-only coupling_head.  Ragged tiles both ways, images smaller than a tile;
    B = 3 makes the persistent workgroups walk several items."""
    H, W = hw
    B = 3
    z, pre = rnd(61, B, 96, H, W), rnd(62, B, 3 * 64, H, W, scale=0.5)
    w0, w2, s0, c0, s2, c2, w4, b4, ps = _step_params(100)
    pk = hip.pack_coupling_wide(w0, w2, s0, c0, s2, c2, w4, b4, ps)
    zh = hip.h2_pack(hip.to_device(z)[:, :48], hip.h2_empty(B, 48, H, W))
    pre_h2 = hip.h2_pack(hip.to_device(pre), hip.h2_empty(B, 3 * 64, H, W))
    hid = hip.coupling_wide_head(zh, pk, pre_h2[:, 8:16], hip.h2_empty(B, 64, H, W))          # a channel-slice view of the level's hoist, as in the engine
    ref = CPU.coupling_head(_h2_values(zh), CPU.pack_coupling_head(w0, w2, s0, c0, s2, c2), _h2_values(pre_h2)[:, 64:128], torch.empty(B, 64, H, W))
    close(_h2_values(hid), ref, 2e-5, "wide head")
    # the launches it replaces: same arithmetic, same summation order -> the same bits
    raw = hip.empty(B, 64, H, W)
    hip.conv_h2x(zh, hip.pack_conv_x3(w0, 2), raw, res1=pre_h2[:, 8:16], alpha1=1.0)
    old = hip.coupling_head(None, hip.pack_coupling_head(None, w2, s0, c0, s2, c2), raw, hip.h2_empty(B, 64, H, W), pre_fmt=0)
    assert torch.equal(old.cpu(), hid.cpu()), "wide head differs from conv_h2x + coupling_head(1x1 only)"
    hip.check_range()


@pytest.mark.parametrize("hw", HW)
@pytest.mark.parametrize("reverse", [0, 1])
def test_wide_tail(hip, hw, reverse):
    """Conv2dZeros 64 -> 96 over the h2 tensor hid + the FlowStep pointwise chain (C = 96) in place, both directions, h_ft NCHW and quad-major,
    forward also without a following step (no matvec); and the h2 copy of the first 48 result channels."""
    H, W = hw
    B, C = 3, 96
    hid, z = rnd(71, B, 64, H, W).abs(), rnd(72, B, C, H, W)
    w0, w2, s0, c0, s2, c2, w4, b4, ps = _step_params(200)
    h_ft = rnd(76, B, 2 * C, H, W, scale=0.5)
    Wm = torch.from_numpy(np.linalg.qr(np.random.Generator(np.random.PCG64(7)).standard_normal((C, C)))[0].astype(np.float32))
    bias, es = rnd(77, C, scale=0.1), torch.exp(rnd(78, C, scale=0.1))
    hid_h2 = CPU.h2_pack(hid, CPU.h2_empty(B, 64, H, W))
    hid22 = sum(CPU._h2_planes(hid_h2))
    pk = hip.pack_coupling_wide(w0, w2, s0, c0, s2, c2, w4, b4, ps)
    wp = hip.pack_wide_wmat(Wm)
    cases = [dict(h_ft=h_ft, w=Wm, an_bias=bias, an_escale=es)]
    if not reverse:
        cases += [dict(w=Wm, an_bias=bias, an_escale=es), dict()]
    for kw in cases:
        ref = CPU.coupling_tail(hid22, CPU.pack_coupling_tail(w4, b4, ps), z, torch.empty_like(z), reverse,
                                **{k: (v.reshape(-1) if k == "w" else v) for k, v in kw.items()})
        zd = hip.to_device(z).clone()
        dkw = {k: (wp if k == "w" else hip.vec(v) if v.dim() <= 2 else hip.to_device(v)) for k, v in kw.items()}
        z1h = hip.h2_empty(B, 48, H, W) if "w" in kw else None
        hip.coupling_wide_tail(hid_h2.to(hip.device), pk, zd, zd, reverse, z1h=z1h, **dkw)
        close(zd, ref, 2e-5, "wide tail rev=%d %s" % (reverse, sorted(kw)))
        if z1h is not None:                     # exactly what h2_pack gives on the result
            want = hip.h2_pack(zd[:, :48], hip.h2_empty(B, 48, H, W))
            assert torch.equal(z1h.cpu(), want.cpu()), "z1h differs from h2_pack(z_out[:, :48])"
        if "h_ft" in kw:                        # h_ft handed over quad-major: bit-identical result
            z1 = hip.to_device(z).clone()
            dk2 = dict(dkw, h_ft=hip.to_device(CPU.quads(h_ft).contiguous()))
            hip.coupling_wide_tail(hid_h2.to(hip.device), pk, z1, z1, reverse, h_ft_fmt=1, **dk2)
            assert torch.equal(z1.cpu(), zd.cpu()), "coupling_wide_tail h_ft_fmt=1 differs from h_ft_fmt=0"
    hip.check_range()


def test_wide_tail_on_batch_slices_is_batch_invariant(hip):
    """Every item is one sample's tile: a batch and its halves give the same bits (the engine runs the level as two half-batch lanes)."""
    B, C, H, W = 4, 96, 24, 40
    hid, z = rnd(171, B, 64, H, W).abs(), rnd(172, B, C, H, W)
    w0, w2, s0, c0, s2, c2, w4, b4, ps = _step_params(300)
    h_ft = hip.to_device(rnd(176, B, 2 * C, H, W, scale=0.5))
    Wm = torch.from_numpy(np.linalg.qr(np.random.Generator(np.random.PCG64(9)).standard_normal((C, C)))[0].astype(np.float32))
    bias, es = hip.vec(rnd(177, C, scale=0.1)), hip.vec(torch.exp(rnd(178, C, scale=0.1)))
    pk, wp = hip.pack_coupling_wide(w0, w2, s0, c0, s2, c2, w4, b4, ps), hip.pack_wide_wmat(Wm)
    hid_h2 = hip.h2_pack(hip.to_device(hid), hip.h2_empty(B, 64, H, W))
    za = hip.to_device(z).clone()
    hip.coupling_wide_tail(hid_h2, pk, za, za, 1, h_ft=h_ft, w=wp, an_bias=bias, an_escale=es)
    zb = hip.to_device(z).clone()
    for b0, b1 in ((0, 2), (2, 4)):
        hip.coupling_wide_tail(hid_h2[b0:b1], pk, zb[b0:b1], zb[b0:b1], 1, h_ft=h_ft[b0:b1], w=wp, an_bias=bias, an_escale=es)
    assert torch.equal(za.cpu(), zb.cpu())


def test_wide_pair_is_reproducible_run_to_run(hip):
    """The same launches on the same inputs, 40 times at the size of config 2's level 3: identical bits every time."""
    B, C, H, W = 8, 96, 80, 80
    z, pre = rnd(261, B, C, H, W), rnd(262, B, 64, H, W, scale=0.5)
    w0, w2, s0, c0, s2, c2, w4, b4, ps = _step_params(400)
    h_ft = hip.to_device(rnd(276, B, 2 * C, H, W, scale=0.5))
    Wm = torch.from_numpy(np.linalg.qr(np.random.Generator(np.random.PCG64(11)).standard_normal((C, C)))[0].astype(np.float32))
    bias, es = hip.vec(rnd(277, C, scale=0.1)), hip.vec(torch.exp(rnd(278, C, scale=0.1)))
    pk, wp = hip.pack_coupling_wide(w0, w2, s0, c0, s2, c2, w4, b4, ps), hip.pack_wide_wmat(Wm)
    zd0 = hip.to_device(z)
    zh = hip.h2_pack(zd0[:, :48], hip.h2_empty(B, 48, H, W))
    pre_h2 = hip.h2_pack(hip.to_device(pre), hip.h2_empty(B, 64, H, W))
    first = None
    for _ in range(40):
        hid = hip.coupling_wide_head(zh, pk, pre_h2, hip.h2_empty(B, 64, H, W))
        zd = zd0.clone()
        z1h = hip.h2_empty(B, 48, H, W)
        hip.coupling_wide_tail(hid, pk, zd, zd, 1, h_ft=h_ft, w=wp, an_bias=bias, an_escale=es, z1h=z1h)
        got = (hid.clone(), zd.clone(), z1h.clone())
        if first is None:
            first = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, got)), "run-to-run difference"
