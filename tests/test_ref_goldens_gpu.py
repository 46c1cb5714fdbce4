"""-m gpu: the per-op / per-FlowStep golden vectors emitted from the GENUINE reference (tests/golden/srflow_ops.npz,
srflow_steps.npz; generator tests/golden/make_golden.py) fed straight to the HIP kernels through the C ABI -- no engine
schedule, no CPU test double in between.  Tolerance: 1e-5 * max(1, |ref|) per op (fp32), 2e-5 for a whole FlowStep."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bfsr_amd import synth                      # noqa: E402
from bfsr_amd.srflow import options, spec       # noqa: E402

T = torch.from_numpy


@pytest.fixture(scope="module")
def hip():
    from bfsr_amd.ops import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "srflow_ops.npz"))


def close(got, ref, tol, what):
    got = got.detach().cpu()
    assert not torch.isnan(got).any(), what + " NaN"
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), "%s: %.3e" % (what, err)


def test_actnorm_and_invconv_goldens(hip, g):
    """FlowActNorms.py:61-113 and Permutations.py:34-58 through bfsr_flow_pointwise."""
    x, bias, logs = T(g["actnorm_x"]), T(g["actnorm_bias"]).reshape(-1), T(g["actnorm_logs"]).reshape(-1)
    d = hip.to_device
    fwd = hip.flow_pointwise(d(x), hip.empty(*x.shape), False, an_bias=hip.vec(bias), an_escale=hip.vec(torch.exp(logs)))
    close(fwd, T(g["actnorm_fwd"]), 1e-6, "actnorm fwd")
    rev = hip.flow_pointwise(d(x), hip.empty(*x.shape), True, an_bias=hip.vec(bias), an_escale=hip.vec(torch.exp(-logs)))
    close(rev, T(g["actnorm_rev"]), 1e-6, "actnorm rev")
    for C in (12, 24, 96):
        w, x = T(g["invconv%d_w" % C]), T(g["invconv%d_x" % C])
        winv = torch.inverse(w.double()).float()                     # Permutations.py:41
        y = hip.flow_pointwise(d(x), hip.empty(*x.shape), False, w=hip.vec(w), wt=hip.vec(w.t().contiguous()))
        close(y, T(g["invconv%d_fwd" % C]), 1e-5, "invconv%d fwd" % C)
        y = hip.flow_pointwise(d(x), hip.empty(*x.shape), True, w=hip.vec(winv), wt=hip.vec(winv.t().contiguous()))
        close(y, T(g["invconv%d_rev" % C]), 1e-5, "invconv%d rev" % C)


def test_flow_conv_goldens(hip, g):
    """flow.Conv2d (conv + ActNorm, flow.py:26-65) and Conv2dZeros (flow.py:68-83) through bfsr_conv2d's fused epilogue."""
    d = hip.to_device
    for k in (3, 1):
        w, x = T(g["fconv%d_w" % k]), T(g["fconv%d_x" % k])
        y = hip.conv(d(x), hip.pack_conv(w), hip.empty(*g["fconv%d_y" % k].shape), aff_shift=T(g["fconv%d_b" % k]).reshape(-1),
                     aff_scale=torch.exp(T(g["fconv%d_logs" % k]).reshape(-1)))
        close(y, T(g["fconv%d_y" % k]), 1e-5, "flow.Conv2d %dx%d" % (k, k))
    y = hip.conv(d(T(g["czero_x"])), hip.pack_conv(T(g["czero_w"])), hip.empty(*g["czero_y"].shape), bias=T(g["czero_b"]),
                 post_scale=torch.exp(T(g["czero_logs"]).reshape(-1) * 3))
    close(y, T(g["czero_y"]), 1e-5, "Conv2dZeros")


def test_squeeze_split_standardize_goldens(hip, g):
    d = hip.to_device
    x = T(g["squeeze_x"])
    y = hip.squeeze2d(d(x), hip.empty(*g["squeeze_y"].shape))
    assert torch.equal(y.cpu(), T(g["squeeze_y"]))                   # pure permutation: bit exact (flow.py:122-134)
    assert torch.equal(hip.unsqueeze2d(y, hip.empty(*x.shape)).cpu(), T(g["unsqueeze_y"]))
    # Split2d (Split.py:48-77): h = Conv2dZeros(z1); eps = (z2 - mean) / exp(logs); reverse z2 = mean + exp(logs) * eps
    xs = d(T(g["split_x"]))
    h = hip.conv(xs[:, :6], hip.pack_conv(T(g["split_w"])), hip.empty(2, 12, 6, 6), bias=T(g["split_b"]),
                 post_scale=torch.exp(T(g["split_logs"]).reshape(-1) * 3))
    eps = hip.split2d(h, xs[:, 6:], hip.empty(2, 6, 6, 6), False)
    close(eps, T(g["split_eps"]), 1e-5, "split eps")
    full = hip.empty(2, 12, 6, 6)
    full[:, :6].copy_(xs[:, :6])
    hip.split2d(h, eps, full[:, 6:], True)
    close(full, T(g["split_rev"]), 1e-5, "split reverse")
    close(hip.standardize(d(T(g["std_x"])), hip.empty(*g["std_x"].shape)), T(g["std_y"]), 1e-5, "eps standardisation")


def _coupling_net(hip, sd, p, x, cout):
    """F() of FlowAffineCouplingsAblation.py:127-135: Conv2d 3x3 + ActNorm, ReLU, Conv2d 1x1 + ActNorm, ReLU, Conv2dZeros."""
    B, _, H, W = x.shape
    h = hip.conv(x, hip.pack_conv(sd[p + "0.weight"]), hip.empty(B, 64, H, W), aff_shift=sd[p + "0.actnorm.bias"].reshape(-1),
                 aff_scale=torch.exp(sd[p + "0.actnorm.logs"].reshape(-1)), act=1)
    h = hip.conv(h, hip.pack_conv(sd[p + "2.weight"]), hip.empty(B, 64, H, W), aff_shift=sd[p + "2.actnorm.bias"].reshape(-1),
                 aff_scale=torch.exp(sd[p + "2.actnorm.logs"].reshape(-1)), act=1)
    return hip.conv(h, hip.pack_conv(sd[p + "4.weight"]), hip.empty(B, cout, H, W), bias=sd[p + "4.bias"],
                    post_scale=torch.exp(sd[p + "4.logs"].reshape(-1) * 3))


@pytest.mark.parametrize("li", [3, 23, 42, 1])
def test_flowstep_goldens(hip, golden_dir, li):
    """One whole FlowStep (FlowStep.py:88-129) of the reference, forward and reverse, composed from the HIP kernels in the
    reference's UN-hoisted form: coupling nets on cat[z1, ft] / ft, then the fused pointwise chain."""
    opt = options.load(options.DEFAULT_CONF)
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234)
    g = np.load(os.path.join(golden_dir, "srflow_steps.npz"))
    if bytes(g["weights_sha256"]).decode() != synth.digest(sd):
        pytest.skip("synthetic weights differ on this machine (numpy/LAPACK build)")
    p = "flowUpsamplerNet.layers.%d." % li
    d = hip.to_device
    z0 = T(g["step%d_z" % li])
    B, C, H, W = z0.shape
    Wm = sd[p + "invconv.weight"]
    Winv = torch.inverse(Wm.double()).float()
    logs = sd[p + "actnorm.logs"].reshape(-1)
    an = dict(an_bias=hip.vec(sd[p + "actnorm.bias"]))
    coupled = li != 1
    ft = d(T(g["step%d_ft" % li])) if coupled else None
    cn = C // 2

    def h_aff(z):
        cat = hip.empty(B, cn + 320, H, W)
        cat[:, :cn].copy_(z[:, :cn])
        cat[:, cn:].copy_(ft)
        return _coupling_net(hip, sd, p + "affine.fAffine.", cat, 2 * (C - cn))

    # ---- forward: actnorm -> invconv -> feature-conditional affine -> self-conditional affine
    z = d(z0).clone()
    h_ft = _coupling_net(hip, sd, p + "affine.fFeatures.", ft, 2 * C) if coupled else None
    hip.flow_pointwise(z, z, False, an_escale=hip.vec(torch.exp(logs)), w=hip.vec(Wm), wt=hip.vec(Wm.t().contiguous()), h_ft=h_ft, **an)
    if coupled:
        hip.flow_pointwise(z, z, False, h_aff=h_aff(z))
    close(z, T(g["step%d_fwd" % li]), 2e-5, "FlowStep %d forward" % li)
    # ---- reverse: self-conditional^-1 -> feature-conditional^-1 -> invconv^-1 -> actnorm^-1
    z = d(z0).clone()
    hip.flow_pointwise(z, z, True, h_aff=h_aff(z) if coupled else None, h_ft=h_ft, w=hip.vec(Winv), wt=hip.vec(Winv.t().contiguous()),
                       an_escale=hip.vec(torch.exp(-logs)), **an)
    close(z, T(g["step%d_rev" % li]), 2e-5, "FlowStep %d reverse" % li)


@pytest.mark.parametrize("tag,la,lb", [("l1", 3, 4), ("l2", 23, 24)])
@pytest.mark.parametrize("quads", [True, False])
def test_flowstep_goldens_through_the_fused_pair(hip, golden_dir, tag, la, lb, quads):
    """Reference goldens of two CONSECUTIVE coupled FlowSteps (FlowStep.py:88-129 on the genuine modules, tests/golden/make_golden_steps_fused.py)
    through the engine's DEFAULT hot path, not the generic composition above: the level's hoist producers (level 1: conv_up2_h2t over the h2
    taps + space-to-depth key planes, the 1x1-only coupling_head, conv_h2r; level 2: the batched 320 -> 1024 hoists) -> coupling_head ->
    coupling_tail, in decode order (b then a) and in encode order (a then b: the tail of step a applies the head of step b), with the
    quad-major hand-over of pre_aff / h_ft (the default) and with NCHW tensors.  B = 2, 36 x 40 / 18 x 20: several 16 x 32 tiles, ragged."""
    from test_srflow_gpu import build
    m, prior, opt, sd, psd = build(hip, 4)
    g = np.load(os.path.join(golden_dir, "srflow_steps_fused.npz"))
    if bytes(g["weights_sha256"]).decode() != synth.digest(sd):
        pytest.skip("synthetic weights differ on this machine (numpy/LAPACK build)")
    eng = m.netG.module.engine()
    ops, d = eng.ops, hip.to_device
    z0 = d(T(g[tag + "_z"]))
    B, C, H, W = z0.shape
    level = [ly.level for ly in eng.layers if ly.index == la][0]
    ft = {}
    if tag == "l1":
        assert eng._taps_up2(level) == 1
        ft[level] = d(T(g["l1_key"]))                                    # this level keeps its 64 key channels only; the taps live at LR resolution
        lrl = eng._lr_level()
        ft[lrl] = hip.zeros(B, 320, H // 2, W // 2)
        ft[lrl][:, 64:].copy_(d(T(g["l1_taps"])))
    else:
        ft[level] = d(T(g["l2_ft"]))
    hz = eng.hoist[level]
    eng._hoist_buffers(level, hz, ft, B)
    cnd = eng._hoist_level(level, hz, ft, B, quads)
    assert cnd["pre_fmt"] == int(quads) or not quads
    sa, sb = eng.steps[la], eng.steps[lb]
    assert getattr(sa, "fused", False) and getattr(sb, "fused", False), "the default path of levels 1 and 2 is the fused pair"

    def hft(idx):
        k = cnd["slot"][idx]
        return cnd["h_ft"][:, 2 * C * k: 2 * C * (k + 1)]

    def pre(idx):
        k = cnd["slot"][idx]
        return cnd["pre_aff"][:, 64 * k: 64 * (k + 1)]

    # ---- decode order: step b reversed, then step a (engine.decode)
    z = z0.clone()
    for st, idx, ref in ((sb, lb, "_rev_b"), (sa, la, "_rev_ba")):
        kw = dict(h_ft=hft(idx), h_ft_fmt=cnd["h_ft_fmt"][idx], w=st.w_inv, an_bias=st.an_bias, an_escale=st.an_expneg)
        z = eng._pair(st, z, pre(idx), "gold_dec", True, kw, cnd["pre_fmt"])
        close(z, T(g[tag + ref]), 2e-5, "fused pair, reverse, step %d (quads %s)" % (idx, quads))
    # ---- encode order (engine.encode): the head of step a on the generic kernel (its h_ft slot stays NCHW: step a is the level's first coupled
    # step), then pair(a) whose tail applies step b's ActNorm, W and feature-conditional affine, then pair(b)
    assert cnd["h_ft_fmt"][la] == 0
    z = z0.clone()
    ops.flow_pointwise(z, z, False, an_bias=sa.an_bias, an_escale=sa.an_exp, w=sa.w_fwd, wt=sa.w_fwd_t, h_ft=hft(la))
    z1 = eng._pair(sa, z.clone(), pre(la), "gold_enc", False, {}, cnd["pre_fmt"])
    close(z1, T(g[tag + "_fwd_a"]), 2e-5, "fused pair, forward, step %d alone (quads %s)" % (la, quads))
    kw = dict(an_bias=sb.an_bias, an_escale=sb.an_exp, w=sb.w_fwd, h_ft=hft(lb), h_ft_fmt=cnd["h_ft_fmt"][lb])
    z = eng._pair(sa, z, pre(la), "gold_enc", False, kw, cnd["pre_fmt"])
    z = eng._pair(sb, z, pre(lb), "gold_enc", False, {}, cnd["pre_fmt"])
    close(z, T(g[tag + "_fwd_ab"]), 2e-5, "fused pairs, forward, steps %d + %d (quads %s)" % (la, lb, quads))
    ops.check_range()


@pytest.mark.parametrize("quads", [True, False])
def test_flowstep_goldens_through_the_wide_pair(hip, golden_dir, quads):
    """Reference goldens of two CONSECUTIVE coupled FlowSteps of level 3 (C = 96; FlowStep.py:88-129 on the genuine modules,
    tests/golden/make_golden_steps_wide.py) through the engine's level-3 hot path of round 6: the batched 320 -> 16*64 hoists (pre_aff as an h2
    tensor), the fFeatures nets, then coupling_wide_head -> coupling_wide_tail (coupling_wide.hip) -- in decode order (b then a: step a's z1 is the
    h2 copy step b's tail wrote, no pack launch) and in encode order (the tail of step a applies the head of step b), with quad-major and NCHW h_ft.
    B = 2, 10 x 36: two 8-row x two 32-pixel tiles, ragged both ways."""
    from test_srflow_gpu import build
    m, prior, opt, sd, psd = build(hip, 4)
    g = np.load(os.path.join(golden_dir, "srflow_steps_wide.npz"))
    if bytes(g["weights_sha256"]).decode() != synth.digest(sd):
        pytest.skip("synthetic weights differ on this machine (numpy/LAPACK build)")
    eng = m.netG.module.engine()
    ops, d = eng.ops, hip.to_device
    la, lb = 42, 43
    lya, lyb = [ly for ly in eng.layers if ly.index == la][0], [ly for ly in eng.layers if ly.index == lb][0]
    z0 = d(T(g["l3_z"]))
    B, C, H, W = z0.shape
    level = lya.level
    assert (level, C, lya.coupled, lyb.coupled) == (3, 96, True, True)
    ft = {level: d(T(g["l3_ft"]))}
    hz = eng.hoist[level]
    eng._hoist_buffers(level, hz, ft, B)
    cnd = eng._hoist_level(level, hz, ft, B, quads)
    sa, sb = eng.steps[la], eng.steps[lb]
    assert sa.wide is not None and sb.wide is not None and cnd["pre_h2"] is not None, "the default path of level 3 is the wide pair"
    assert cnd["h_ft_fmt"][la] == 0 and cnd["h_ft_fmt"][lb] == int(quads)

    def hft(idx):
        k = cnd["slot"][idx]
        return cnd["h_ft"][:, 2 * C * k: 2 * C * (k + 1)]

    def pre(idx):
        k = cnd["slot"][idx]
        return cnd["pre_aff"][:, 64 * k: 64 * (k + 1)]

    def wide(st, ly, nxt):
        w = eng._wide_info(st, cnd, cnd["slot"][ly.index], ly, nxt)
        assert w is not None and (w["nxt"] is None) == (nxt is None)
        return w

    # ---- decode order: step b reversed (its tail writes step a's z1 as an h2 tensor), then step a
    z = z0.clone()
    eng._z1h_next.clear()
    for st, ly, nxt, ref in ((sb, lyb, lya, "_rev_b"), (sa, lya, None, "_rev_ba")):
        kw = dict(h_ft=hft(ly.index), h_ft_fmt=cnd["h_ft_fmt"][ly.index], w=st.w_inv, an_bias=st.an_bias, an_escale=st.an_expneg)
        z = eng._pair(st, z, pre(ly.index), "gold_dec3", True, kw, cnd["pre_fmt"], wide=wide(st, ly, nxt))
        close(z, T(g["l3" + ref]), 2e-5, "wide pair, reverse, step %d (quads %s)" % (ly.index, quads))
        if nxt is not None:
            assert eng._z1h_next.get("gold_dec3") == (z.data_ptr(), nxt.index), "the tail must have left the next step's z1 behind"
    # ---- encode order: the head of step a on the generic kernel (its h_ft slot stays NCHW: the level's first coupled step), then pair(a)
    # whose tail applies step b's ActNorm, W and feature-conditional affine, then pair(b)
    z = z0.clone()
    ops.flow_pointwise(z, z, False, an_bias=sa.an_bias, an_escale=sa.an_exp, w=sa.w_fwd, wt=sa.w_fwd_t, h_ft=hft(la))
    z1 = eng._pair(sa, z.clone(), pre(la), "gold_enc3", False, {}, cnd["pre_fmt"], wide=wide(sa, lya, None))
    close(z1, T(g["l3_fwd_a"]), 2e-5, "wide pair, forward, step %d alone (quads %s)" % (la, quads))
    kw = dict(an_bias=sb.an_bias, an_escale=sb.an_exp, w=sb.w_fwd, h_ft=hft(lb), h_ft_fmt=cnd["h_ft_fmt"][lb])
    z = eng._pair(sa, z, pre(la), "gold_enc3", False, kw, cnd["pre_fmt"], wide=wide(sa, lya, lyb))
    z = eng._pair(sb, z, pre(lb), "gold_enc3", False, {}, cnd["pre_fmt"], wide=wide(sb, lyb, None))
    close(z, T(g["l3_fwd_ab"]), 2e-5, "wide pairs, forward, steps %d + %d (quads %s)" % (la, lb, quads))
    ops.check_range()

