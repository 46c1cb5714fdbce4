"""-m gpu: LINF-LP kernels and end-to-end parity on the HIP path."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bfsr_amd import synth                       # noqa: E402
from bfsr_amd.linf import spec as lspec          # noqa: E402
from cpu_ops import CpuOps                       # noqa: E402
from test_linf_cpu import CASES, mspec, weights  # noqa: E402

T = torch.from_numpy
CPU = CpuOps()


@pytest.fixture(scope="module")
def hip():
    from bfsr_amd.ops import HipOps
    return HipOps("cuda:0")


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def close(a, b, tol, what=""):
    a = a.detach().cpu()
    assert not torch.isnan(a).any(), what + " NaN"
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), "%s: %.3e" % (what, err)


@pytest.mark.parametrize("hw,q", [((8, 12), (21, 33)), ((16, 16), (22, 22)), ((5, 7), (30, 9))])
def test_linf_features(hip, hw, q):
    import oracle.linf_ref as O
    h, w = hw
    qh, qw = q
    B, HD = 2, 256
    cf = rnd(1, B, 2 * HD, h, w)
    phase = rnd(2, HD // 2, 2, scale=0.5)
    H, W = qh * 3 - 1, qw * 3 - 2
    prep = O.batch_prep(torch.rand(B, 3, h, w), (H, W))
    coord, cell = prep["coord"], prep["cell"]
    assert coord.shape[1:3] == (qh, qw)
    ref = CPU.linf_features(cf, coord, cell, phase.reshape(-1), torch.empty(B, 4 * HD, qh, qw), HD)
    out = hip.linf_features(hip.to_device(cf), hip.to_device(coord), hip.to_device(cell), hip.vec(phase),
                            hip.empty(B, 4 * HD, qh, qw), HD)
    close(out, ref, 5e-6, "linf_features")


@pytest.mark.parametrize("hw,q", [((8, 12), (21, 33)), ((16, 16), (22, 22)), ((5, 7), (30, 9)), ((24, 20), (65, 55))])
@pytest.mark.parametrize("x3", [True, False, "bf16x3"])
def test_linf_mlp_fused(hip, hw, q, x3):
    """The fused conditioning kernel (linf_mlp.hip: Fourier features -> 1024-256-256-256-540 MLP) against the unfused semantics
    (features + 1x1 convs): fp32-accurate in the 3xBF16 mode; in fp16 mode against the same chain with every layer's operands
    rounded to fp16.  Tiles of 64 query points: sizes with ragged last tiles and several tiles per image."""
    import oracle.linf_ref as O
    split0 = hip.split
    if x3 == "bf16x3":                                                  # the fp32-accurate mode under the other split (default: f16x2)
        hip.split, x3 = "bf16x3", True
    try:
        _linf_mlp_fused_body(hip, O, hw, q, x3)
    finally:
        hip.split = split0


def _linf_mlp_fused_body(hip, O, hw, q, x3):
    h, w = hw
    qh, qw = q
    B, HD, Cout = 2, 256, 540
    cf = rnd(1, B, 2 * HD, h, w)
    phase = rnd(2, HD // 2, 2, scale=0.5)
    ws = [rnd(11, HD, 4 * HD, scale=1.0 / 32), rnd(12, HD, HD, scale=1.0 / 16), rnd(13, HD, HD, scale=1.0 / 16), rnd(14, Cout, HD, scale=1.0 / 16)]
    bs = [rnd(15, HD, scale=0.1), rnd(16, HD, scale=0.1), rnd(17, HD, scale=0.1), rnd(18, Cout, scale=0.1)]
    H, W = qh * 3 - 1, qw * 3 - 2
    prep = O.batch_prep(torch.rand(B, 3, h, w), (H, W))
    coord, cell = prep["coord"], prep["cell"]
    assert coord.shape[1:3] == (qh, qw)
    ref = CPU.linf_mlp(cf, coord, cell, phase.reshape(-1), CPU.pack_linf_mlp(ws, bs, x3=x3), torch.empty(B, Cout, qh, qw), HD, x3=x3)
    out = hip.linf_mlp(hip.to_device(cf), hip.to_device(coord), hip.to_device(cell), hip.vec(phase), hip.pack_linf_mlp(ws, bs, x3=x3),
                       hip.empty(B, Cout, qh, qw), HD, x3=x3)
    close(out, ref, 2e-5 if x3 else 3e-3, "linf_mlp x3=%s" % x3)
    # the private quad-major layout (16-byte stores; what linf_flow(ai_fmt=1) reads): the same values, rows padded per flow layer
    L, D = 10, 27
    outq = hip.linf_mlp(hip.to_device(cf), hip.to_device(coord), hip.to_device(cell), hip.vec(phase), hip.pack_linf_mlp(ws, bs, x3=x3, quad_layers=(L, D)),
                        hip.empty(B, 56 * L, qh, qw), HD, x3=x3)
    assert torch.equal(CPU._ai_quads(outq.cpu(), L, D, inverse=True), out.cpu()), "out_fmt=1 must hold the values of out_fmt=0"
    # 128-point workgroup tiles (optional, BFSR_MLP_TILE=128; the bf16x3 mode has no such form): same summation order per point -> identical bits
    o128 = hip.linf_mlp(hip.to_device(cf), hip.to_device(coord), hip.to_device(cell), hip.vec(phase), hip.pack_linf_mlp(ws, bs, x3=x3),
                        hip.empty(B, Cout, qh, qw), HD, x3=x3, tile=128)
    assert torch.equal(o128.cpu(), out.cpu()), "64- and 128-point tiles must give identical bits"
    # cf handed over as an h2 tensor (cf_fmt = 1: 16-byte gathers of 8 channels per plane): bit-identical to the fp32 map holding hi + lo
    cfh = hip.h2_pack(hip.to_device(cf), hip.h2_empty(B, 2 * HD, h, w))
    cf22 = hip.h2_unpack(cfh, hip.empty(B, 2 * HD, h, w))
    pk = hip.pack_linf_mlp(ws, bs, x3=x3)
    a32 = hip.linf_mlp(cf22, hip.to_device(coord), hip.to_device(cell), hip.vec(phase), pk, hip.empty(B, Cout, qh, qw), HD, x3=x3)
    ah2 = hip.linf_mlp(cfh, hip.to_device(coord), hip.to_device(cell), hip.vec(phase), pk, hip.empty(B, Cout, qh, qw), HD, x3=x3)
    assert torch.equal(a32, ah2), "cf_fmt=1 must equal the fp32 map of the same 22-bit values"


@pytest.mark.parametrize("D", [27, 3])
@pytest.mark.parametrize("reverse", [0, 1])
def test_linf_flow(hip, D, reverse):
    L, B, qh, qw = 10, 2, 13, 17
    q = np.stack([np.linalg.qr(np.random.Generator(np.random.PCG64(i)).standard_normal((D, D)))[0] for i in range(L + 1)])
    Wm = torch.from_numpy(q.astype(np.float32)) * torch.from_numpy(np.random.Generator(np.random.PCG64(5)).uniform(0.8, 1.25, (L + 1, 1, D)).astype(np.float32))
    Wuse = torch.inverse(Wm.double()).float() if reverse else Wm
    bb = rnd(3, L + 1, D, scale=0.1)
    x, ai = rnd(4, B, D, qh, qw), rnd(5, B, 2 * D * L, qh, qw, scale=0.5)
    ref = CPU.linf_flow(x, ai, torch.empty_like(x), Wuse.reshape(-1), bb.reshape(-1), L, reverse)
    out = hip.linf_flow(hip.to_device(x), hip.to_device(ai), hip.empty(*x.shape), hip.vec(Wuse), hip.vec(bb), L, reverse)
    close(out, ref, 1e-5, "linf_flow")
    # conditioning handed over in the quad-major padded layout.  D = 3: the same kernel, bit-identical; D = 27: the matrix-pipe kernel
    # (linf_flow27_mfma_kernel: fp32 MFMAs in another summation order, v_exp / v_rcp sigmoid) -- held to the same bound against the CPU semantics
    aiq = hip.to_device(CPU._ai_quads(ai, L, D).contiguous())
    outq = hip.linf_flow(hip.to_device(x), aiq, hip.empty(*x.shape), hip.vec(Wuse), hip.vec(bb), L, reverse, ai_fmt=1)
    if D == 27:
        close(outq, ref, 1e-5, "linf_flow (matrix pipe)")
    else:
        assert torch.equal(outq.cpu(), out.cpu()), "linf_flow ai_fmt=1 differs from ai_fmt=0"
    if not reverse:     # the log-density form keeps the one-point-per-lane kernel in both layouts: identical z, identical log_p
        lp0, lp1 = hip.empty(B * qh * qw), hip.empty(B * qh * qw)
        z0 = hip.linf_flow(hip.to_device(x), hip.to_device(ai), hip.empty(*x.shape), hip.vec(Wuse), hip.vec(bb), L, reverse, log_p=lp0, logdet_const=0.25)
        z1 = hip.linf_flow(hip.to_device(x), aiq, hip.empty(*x.shape), hip.vec(Wuse), hip.vec(bb), L, reverse, log_p=lp1, logdet_const=0.25, ai_fmt=1)
        assert torch.equal(z0.cpu(), z1.cpu()) and torch.equal(lp0.cpu(), lp1.cpu()), "log_p form differs between the two layouts"
        close(z0, ref, 1e-5, "linf_flow (log_p form)")


@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 16, 16), (2, 31, 33), (1, 70, 70)])
def test_linf_flow_matrix_pipe_ragged(hip, shape, reverse):
    """linf_flow27_mfma_kernel on query grids that are not multiples of its 64-point waves / 256-point chunks (clamped loads, masked stores),
    several chunks per block, batch slices; a forward / inverse round trip returns the input."""
    B, qh, qw = shape
    L, D = 10, 27
    q = np.stack([np.linalg.qr(np.random.Generator(np.random.PCG64(40 + i)).standard_normal((D, D)))[0] for i in range(L + 1)])
    Wm = torch.from_numpy(q.astype(np.float32)) * torch.from_numpy(np.random.Generator(np.random.PCG64(6)).uniform(0.8, 1.25, (L + 1, 1, D)).astype(np.float32))
    Winv = torch.inverse(Wm.double()).float()
    bb = rnd(13, L + 1, D, scale=0.1)
    x, ai = rnd(14, B, D, qh, qw), rnd(15, B, 2 * D * L, qh, qw, scale=0.5)
    Wuse = Winv if reverse else Wm
    ref = CPU.linf_flow(x, ai, torch.empty_like(x), Wuse.reshape(-1), bb.reshape(-1), L, reverse)
    aiq = hip.to_device(CPU._ai_quads(ai, L, D).contiguous())
    ybuf = hip.empty(B, D, qh, qw).fill_(float("nan"))
    out = hip.linf_flow(hip.to_device(x), aiq, ybuf, hip.vec(Wuse), hip.vec(bb), L, reverse, ai_fmt=1)
    close(out, ref, 1e-5, "linf_flow matrix pipe %s" % (shape,))
    back = hip.linf_flow(out, aiq, hip.empty(B, D, qh, qw), hip.vec(Wm if reverse else Winv), hip.vec(bb), L, not reverse, ai_fmt=1)
    close(back, x, 2e-5, "round trip")
    if B > 1:           # per-sample: a batch slice gives the bits of the batch
        part = hip.linf_flow(hip.to_device(x)[1:], aiq[1:], hip.empty(B - 1, D, qh, qw), hip.vec(Wuse), hip.vec(bb), L, reverse, ai_fmt=1)
        assert torch.equal(part.cpu(), out[1:].cpu())


def test_fold_unfold_direct_conv(hip):
    import torch.nn.functional as F
    p = rnd(6, 2, 27, 5, 7)
    for H, W in ((15, 21), (13, 19)):
        ref = CPU.patch_fold(p, torch.empty(2, 3, H, W), 3)
        assert torch.equal(hip.patch_fold(hip.to_device(p), hip.empty(2, 3, H, W), 3).cpu(), ref)
        img = rnd(7, 2, 3, H, W)
        ref = CPU.patch_unfold(img, torch.empty(2, 27, 5, 7), 3)
        assert torch.equal(hip.patch_unfold(hip.to_device(img), hip.empty(2, 27, 5, 7), 3).cpu(), ref)
    x, w, b = rnd(8, 2, 3, 16, 23), rnd(9, 27, 3, 3, 3, scale=0.3), rnd(10, 27, scale=0.1)
    ref = F.leaky_relu(F.conv2d(x, w, b, 3, 1), 0.2)
    out = hip.conv_direct(hip.to_device(x), hip.to_device(w), hip.vec(b), hip.empty(*ref.shape), 3, 1, act=2, slope=0.2)
    close(out, ref, 1e-5, "conv_direct")


@pytest.mark.parametrize("tag,enc,seed,c", CASES)
def test_golden_e2e(hip, golden_dir, tag, enc, seed, c):
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import infer_from_lr, lp_infer
    g = np.load(os.path.join(golden_dir, "linf_e2e_%s_%s.npz" % (tag, c)))
    sd, psd = weights(enc, seed)
    m = make(mspec(enc), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": hip}).eval()
    prior.load_state_dict(psd)
    s, lr = int(g["scale"]), T(g["lr"])
    H, W = s * lr.shape[2], s * lr.shape[3]
    batch = dict(inp=lr, coord=T(g["coord"]), cell=T(g["cell"]), gt_lr_up=T(g["gt_lr_up"]))
    out = lp_infer(m, prior, batch, (H, W), return_all=True)
    for k in ("z_lr", "z_learned", "pred_raw"):
        close(out[k], T(g[k]), 1e-4, k)
    assert (out["pred"].cpu() - T(g["pred"])).abs().max() <= 1e-4
    assert (infer_from_lr(m, prior, lr, s).cpu() - T(g["pred"])).abs().max() <= 1e-4
    if "feat" in g.files:
        close(m("gen_feat", inp=(lr - 0.5) / 0.5), T(g["feat"]), 2e-5, "encoder")


@pytest.mark.parametrize("B,h,w,H,W,ps,pad", [(2, 24, 20, 96, 80, 3, True), (1, 16, 16, 42, 45, 3, True), (2, 15, 17, 45, 51, 3, False), (1, 12, 12, 27, 31, 1, True),
                                              (3, 32, 32, 192, 192, 3, True)])
def test_fused_glue_equals_launches(hip, monkeypatch, B, h, w, H, W, ps, pad):
    """The fused harness glue of round 6 (bfsr_linf_prep_down / _prep_residual: datasets/wrappers.py:203-228; bfsr_linf_fold_skip: LINF-LP/test.py:168-171, 217)
    against the launch sequences it replaces (resize x3, axpb_clamp x2, patch_unfold | patch_fold, resize, axpb_clamp x2): the same BITS, for HR sizes that are
    and are not multiples of the patch size / of four, with and without the wrapper's always-pad rule, and for the pixel-wise wrapper (ps = 1)."""
    from bfsr_amd.linf import prep
    from bfsr_amd.ops import MODE_BILINEAR
    inp01 = hip.to_device(synth.lr_batch(700 + H, B, h, w))
    res = {}
    for mode in ("fused", "launches"):
        monkeypatch.setenv("BFSR_LINF_GLUE", mode)
        bt = prep.prepare_batch_pixelwise(hip, inp01, (H, W)) if ps == 1 else prep.prepare_batch(hip, inp01, (H, W), ps, pad)
        torch.cuda.synchronize()
        res[mode] = bt["gt_lr_up"].clone()
    assert res["fused"].shape == res["launches"].shape
    assert torch.equal(res["fused"], res["launches"]), "prep: max diff %.3e" % float((res["fused"] - res["launches"]).abs().max())
    if ps == 1:
        return
    qh, qw = res["fused"].shape[2:]
    p = hip.to_device(rnd(800 + W, B, 3 * ps * ps, qh, qw))
    inp = hip.axpb_clamp(inp01, hip.empty(B, 3, h, w), 2.0, -1.0)
    full = hip.patch_fold(p, hip.empty(B, 3, ps * qh, ps * qw), ps)
    pred = hip.patch_fold(p, hip.empty(B, 3, H, W), ps)
    assert torch.equal(pred, full[..., :H, :W])
    skip = hip.resize(inp, hip.empty(B, 3, H, W), MODE_BILINEAR, float(h) / H, float(w) / W)
    raw_l = hip.axpb_clamp(pred, hip.empty(B, 3, H, W), 1.0, 0.0, r=skip)
    out_l = hip.axpb_clamp(raw_l, hip.empty(B, 3, H, W), 0.5, 0.5, 0.0, 1.0)
    raw_f, out_f = hip.linf_fold_skip(p, inp, H, W, ps, raw=hip.empty(B, 3, H, W), out=hip.empty(B, 3, H, W))
    assert torch.equal(raw_f, raw_l) and torch.equal(out_f, out_l), "fold + skip + clamp: max diff %.3e" % float((raw_f - raw_l).abs().max())
    _, out_only = hip.linf_fold_skip(p, inp, H, W, ps, out=hip.empty(B, 3, H, W))
    assert torch.equal(out_only, out_l)


def test_lp_infer_fused_glue_equals_launches(hip, monkeypatch):
    """The whole LP pass (rrdb, x4 and a x2.5 case whose HR width is cropped inside a patch) with the fused glue and with the launches: identical outputs."""
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import infer_from_lr
    sd, psd = weights("rrdb", 2024)
    m = make(mspec("rrdb"), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": hip}).eval()
    prior.load_state_dict(psd)
    for (hh, ww, scale) in ((32, 24, 4), (20, 20, 2.5), (48, 48, 8)):      # the last one: a 129 x 129 query grid, the prior's top level on the LDS-DMA kernels (h2 glue)
        lr = synth.smooth_lr_batch(5, 2, hh, ww)
        res = {}
        for mode in ("fused", "launches"):
            monkeypatch.setenv("BFSR_LINF_GLUE", mode)
            monkeypatch.setenv("BFSR_PRIOR_GLUE", mode)
            res[mode] = infer_from_lr(m, prior, lr, scale, return_all=True)
        for k in ("z_lr", "z_learned", "pred_raw", "pred"):
            assert torch.equal(res["fused"][k], res["launches"][k]), "%s at x%s: max diff %.3e" % (k, scale, float((res["fused"][k] - res["launches"][k]).abs().max()))


def test_vs_oracle_and_roundtrip_bigger(hip):
    """Fresh input at a larger size (rrdb, x4, 64x48 LR, B=2) vs the oracle + decode(encode(x)) == x."""
    import oracle.linf_ref as O
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import lp_infer
    sd, psd = weights("rrdb", 2024)
    m = make(mspec("rrdb"), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": hip}).eval()
    prior.load_state_dict(psd)
    lr = synth.smooth_lr_batch(3, 2, 64, 48)
    H, W = 256, 192
    prep = O.batch_prep(lr, (H, W))
    ref = O.lp_pipeline(prep, sd, psd, mspec("rrdb"), (H, W), return_all=True)
    out = lp_infer(m, prior, prep, (H, W), return_all=True)
    close(out["z_lr"], ref["z_lr"], 1e-4, "z_lr")
    assert (out["pred"].cpu() - ref["pred"]).abs().max() <= 1e-4
    inp = hip.to_device((lr - 0.5) / 0.5)
    feat = m("gen_feat", inp=inp)
    coord, cell, gt = (hip.to_device(prep[k]) for k in ("coord", "cell", "gt_lr_up"))
    z = m("query_log_p", feat=feat, coord=coord, cell=cell, gt=gt)[1]
    back = hip.patch_unfold(m("query_rgb", feat=feat, coord=coord, cell=cell, zmap=z), hip.empty(*gt.shape), 3)
    assert (back - gt).abs().max().item() <= 1e-4


def test_stochastic_path_and_randomness_loop(hip):
    """tau path (no prior): z ~ N(0, tau^2) on device; `--randomness`: 5 samples, PSNR + diversity."""
    import oracle.linf_ref as O
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import eval_psnr
    sd, psd = weights("edsr-baseline", 2025)
    m = make(mspec("edsr-baseline"), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    lr = synth.smooth_lr_batch(9, 1, 16, 16)
    prep = dict(O.batch_prep(lr, (64, 64)), gt=torch.rand(1, 3, 64, 64))
    zero = eval_psnr([prep], m, None, temperature=0.0)                      # tau = 0 is deterministic
    assert zero == eval_psnr([prep], m, None, temperature=0.0)
    r = eval_psnr([prep], m, None, temperature=0.8, randomness=True)
    assert r["diversity"] > 0 and np.isfinite(r["psnr"])


@pytest.mark.parametrize("c", ["s4", "s3"])
def test_pixelwise_linf_golden(hip, golden_dir, c):
    from bfsr_amd.linf.test import infer_from_lr, lp_infer
    from test_linf_cpu import pixelwise_models
    m, prior = pixelwise_models(hip)
    g = np.load(os.path.join(golden_dir, "linf_e2e_pixelwise_%s.npz" % c))
    s, lr = int(g["scale"]), T(g["lr"])
    H, W = s * lr.shape[2], s * lr.shape[3]
    batch = dict(inp=lr, coord=T(g["coord"]), cell=T(g["cell"]), gt_lr_up=T(g["gt_lr_up"]))
    out = lp_infer(m, prior, batch, (H, W), return_all=True)
    for k in ("z_lr", "z_learned", "pred_raw"):
        close(out[k], T(g[k]), 1e-4, k)
    assert (out["pred"].cpu() - T(g["pred"])).abs().max() <= 1e-4
    assert (infer_from_lr(m, prior, lr, s).cpu() - T(g["pred"])).abs().max() <= 1e-4


def test_grid_sample_add(hip):
    import torch.nn.functional as F
    x, acc = rnd(150, 2, 3, 7, 9), rnd(151, 2, 3, 11, 13)
    coord = (torch.rand(2, 11, 13, 2) * 2.4 - 1.2).contiguous()        # includes out-of-range points (border padding)
    ref = acc + F.grid_sample(x, coord.flip(-1), mode="bilinear", padding_mode="border", align_corners=False)
    out = hip.grid_sample_add(hip.to_device(x), hip.to_device(coord), hip.to_device(acc), hip.empty(2, 3, 11, 13))
    close(out, ref, 2e-6, "grid_sample_add")


FP16_TOL_PRED = 1e-3      # stated tolerance of the fp16-MFMA path against the fp32 REFERENCE on the [0,1] output image
FP16_TOL_LATENT = 2e-3    # ... and on the latents, relative to max(1, |ref|)


@pytest.mark.parametrize("c", ["s6", "s4"])
def test_fp16_mfma_path_vs_reference_golden(hip, golden_dir, c):
    """BASELINE config 5 in miniature (rrdb-linf-LP, fp16 MFMA path; s6 = the out-of-distribution x6 scale) against the GENUINE
    reference's fp32 result.  Reduced precision is outside the 1e-4 fp32 bar by construction; the bar for this path is
    FP16_TOL_PRED on `pred` (north_star states no number for fp16; measured 1-3e-4, asserted at 1e-3)."""
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import lp_infer
    g = np.load(os.path.join(golden_dir, "linf_e2e_rrdb_%s.npz" % c))
    sd, psd = weights("rrdb", 2024)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}},
                 args={"ops": hip, "precision": "fp16"}).eval()
    prior.load_state_dict(psd)
    m = make(mspec("rrdb"), args={"ops": hip, "precision": "fp16"}).eval()
    m.load_state_dict(sd)
    s, lr = int(g["scale"]), T(g["lr"])
    H, W = s * lr.shape[2], s * lr.shape[3]
    batch = dict(inp=lr, coord=T(g["coord"]), cell=T(g["cell"]), gt_lr_up=T(g["gt_lr_up"]))
    out = lp_infer(m, prior, batch, (H, W), return_all=True)
    dev = (out["pred"].cpu() - T(g["pred"])).abs().max().item()
    dz = (out["z_lr"].cpu() - T(g["z_lr"])).abs().max().item() / max(1.0, float(np.abs(g["z_lr"]).max()))
    dzl = (out["z_learned"].cpu() - T(g["z_learned"])).abs().max().item() / max(1.0, float(np.abs(g["z_learned"]).max()))
    print("fp16 MFMA path vs reference (%s): max-abs pred %.3e, rel z_lr %.3e, rel z_learned %.3e" % (c, dev, dz, dzl))
    assert torch.isfinite(out["pred"]).all()
    assert 0 < dev <= FP16_TOL_PRED and dz <= FP16_TOL_LATENT and dzl <= FP16_TOL_LATENT


def _bench_size_models(hip, precision):
    from bfsr_amd.linf.models import make
    sd, psd = weights("rrdb", 2024)
    m = make(mspec("rrdb"), args={"ops": hip, "precision": precision}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}},
                 args={"ops": hip, "precision": precision}).eval()
    prior.load_state_dict(psd)
    return m, prior, sd, psd


@pytest.mark.parametrize("cfg", ["cfg3_x2", "cfg3_x3", "cfg3_x4", "cfg5_x6_fp16"])
def test_bench_size_properties(hip, cfg):
    """BASELINE configs 3 and 5 at FULL size (config 3: B=16, 256x256 LR, x2/x3/x4; config 5 per GPU: B=16, 128x128 LR, x6, fp16
    MFMA path): size-independent properties -- query_rgb(query_log_p(x)) == x within 1e-4 (the flow is inverted exactly on the
    shared conditioning), bit-identical results under batch sharding, the re-fold branch when the HR width is not a multiple
    of 3 -- plus an oracle comparison on a B=1 crop of the SAME input (fp32 configs)."""
    import oracle.linf_ref as O
    from bfsr_amd.linf import prep
    from bfsr_amd.linf.test import lp_infer
    scale = int(cfg.split("_x")[1].split("_")[0])
    fp16 = cfg.endswith("fp16")
    B, n = (16, 128) if cfg.startswith("cfg5") else (16, 256)
    m, prior, sd, psd = _bench_size_models(hip, "fp16" if fp16 else "fp32")
    lr = hip.to_device(synth.smooth_lr_batch(40 + scale, B, n, n))
    H = W = n * scale
    batch = prep.prepare_batch(hip, lr, (H, W), 3, True)
    Q = batch["coord"].shape[1]
    assert Q == (H + (3 - H % 3)) // 3                      # wrappers.py:218-219: pad = ps - H % ps even when divisible
    inp = hip.axpb_clamp(lr, hip.empty(B, 3, n, n), 2.0, -1.0)
    feat = m("gen_feat", inp=inp)
    z = m("query_log_p", feat=feat, coord=batch["coord"], cell=batch["cell"], gt=batch["gt_lr_up"])[1]
    back = hip.patch_unfold(m("query_rgb", feat=feat, coord=batch["coord"], cell=batch["cell"], zmap=z), hip.empty(*batch["gt_lr_up"].shape), 3)
    rt = (back - batch["gt_lr_up"]).abs().max().item()
    assert rt <= 1e-4, "round trip %.3e" % rt
    out = lp_infer(m, prior, batch, (H, W), return_all=True)
    assert out["pred"].shape == (B, 3, H, W) and torch.isfinite(out["pred"]).all()
    # batch sharding: samples 5..6 alone give bit-identical latents and images
    sub = {k: v[5:7].contiguous() for k, v in batch.items()}
    out2 = lp_infer(m, prior, sub, (H, W), return_all=True)
    for k in ("z_lr", "z_learned", "pred"):
        assert torch.equal(out[k][5:7], out2[k]), "batch sharding changed %s" % k
    if not fp16:    # oracle on a 24x20 crop of sample 3 of the same input
        crop = lr[3:4, :, 100:124, 60:80].contiguous().cpu()
        h, w = crop.shape[-2:]
        pb = O.batch_prep(crop, (h * scale, w * scale))
        ref = O.lp_pipeline(pb, sd, psd, mspec("rrdb"), (h * scale, w * scale), return_all=True)
        got = lp_infer(m, prior, pb, (h * scale, w * scale), return_all=True)
        close(got["z_lr"], ref["z_lr"], 1e-4, "crop z_lr")
        assert (got["pred"].cpu() - ref["pred"]).abs().max() <= 1e-4
    m.engine().ws.bufs.clear()
    torch.cuda.empty_cache()


def test_eval_psnr_detail_vs_reference(hip, golden_dir):
    """`eval_psnr(detail=True)` on the HIP path == the genuine reference's dict (psnr / ssim / LR recon), eval_type div2k-4 and
    benchmark-4 (tests/golden/linf_detail.npz)."""
    import oracle.linf_ref as O
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.test import eval_psnr
    rd = np.load(os.path.join(golden_dir, "linf_detail.npz"))
    sd, psd = weights("edsr-baseline", 2025)
    m = make(mspec("edsr-baseline"), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    prior = make({"name": "unet", "args": {"in_chans": 27, "depth": 3, "dim": 64, "bilinear": True}}, args={"ops": hip}).eval()
    prior.load_state_dict(psd)
    batch = dict(O.batch_prep(T(rd["lr"]), (192, 192)), gt=T(rd["hr"]))
    for et in ("div2k-4", "benchmark-4"):
        d = eval_psnr([batch], m, prior, eval_type=et, detail=True)
        k = et.replace("-", "")
        assert abs(d["psnr"] - float(rd[k + "_psnr"])) <= 1e-3
        assert abs(d["ssim"] - float(rd[k + "_ssim"])) <= 1e-4
        assert abs(d["LR recon"] - float(rd[k + "_LR_recon"])) <= 1e-2


def test_query_log_p_values_golden(hip, golden_dir):
    """log_p per query point from the flow kernel == the genuine reference's (tests/golden/linf_logp.npz)."""
    from test_linf_cpu import _logp_case
    log_p, z, ref_lp, ref_z = _logp_case(golden_dir, hip)
    close(z, ref_z, 1e-4, "z")
    assert ((log_p.cpu() - ref_lp).abs() / ref_lp.abs().clamp_min(1.0)).max() <= 1e-5


def test_linf_flow_vjp_of_inverse(hip):
    """linf_flow mode 2 = vector-Jacobian product of the inverse flow w.r.t. its input, against torch.autograd through the fp32
    torch semantics of the inverse (cpu_ops.linf_flow)."""
    D, L, B, qh, qw = 27, 10, 2, 13, 17
    q = np.stack([np.linalg.qr(np.random.Generator(np.random.PCG64(i)).standard_normal((D, D)))[0] for i in range(L + 1)])
    Wm = torch.from_numpy(q.astype(np.float32)) * torch.from_numpy(np.random.Generator(np.random.PCG64(5)).uniform(0.8, 1.25, (L + 1, 1, D)).astype(np.float32))
    Winv = torch.inverse(Wm.double()).float()
    bb, ai, gy = rnd(6, L + 1, D, scale=0.1), rnd(7, B, 2 * D * L, qh, qw), rnd(8, B, D, qh, qw)
    with torch.enable_grad():
        z = rnd(9, B, D, qh, qw).requires_grad_(True)
        v = z.permute(0, 2, 3, 1).reshape(-1, D)
        a = ai.permute(0, 2, 3, 1).reshape(-1, 2 * D * L)
        v = torch.nn.functional.linear(v - bb[L], Winv[L])
        for i in reversed(range(L)):
            v = (v - a[:, 2 * D * i + D: 2 * D * (i + 1)]) / (torch.sigmoid(a[:, 2 * D * i: 2 * D * i + D] + 2.0) + 1e-4)
            v = torch.nn.functional.linear(v - bb[i], Winv[i])
        (v.reshape(B, qh, qw, D).permute(0, 3, 1, 2) * gy).sum().backward()
    out = hip.linf_flow(hip.to_device(gy), hip.to_device(ai), hip.empty(B, D, qh, qw), hip.vec(Winv.transpose(1, 2).contiguous()), hip.vec(bb), L, reverse=2)
    close(out, z.grad, 1e-5, "linf_flow vjp")
    close(CPU.linf_flow(gy, ai, torch.empty(B, D, qh, qw), Winv.transpose(1, 2).contiguous().reshape(-1), bb.reshape(-1), L, 2), z.grad, 1e-5, "cpu double vjp")


def test_query_rgb_backward_vs_reference_autograd(hip, golden_dir):
    """f4, the hot-path part of latent-module training: d/dzmap of the frozen model's query_rgb
    (LINF-LP/train.py:143) on the HIP path against the genuine reference's autograd gradient (tests/golden/linf_vjp.npz)."""
    from bfsr_amd.linf.models import make
    g = np.load(os.path.join(golden_dir, "linf_vjp.npz"))
    sd, _ = weights("edsr-baseline", int(g["weights_seed"]))
    m = make(mspec("edsr-baseline"), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    inp = hip.to_device((T(g["lr"]) - 0.5) / 0.5)
    coord, cell = hip.to_device(T(g["coord"])), hip.to_device(T(g["cell"]))
    feat = m("gen_feat", inp=inp)
    with torch.enable_grad():
        z = hip.to_device(T(g["zmap"])).clone().requires_grad_(True)
        pred = m("query_rgb", inp=inp, feat=feat, coord=coord, cell=cell, zmap=z)
        (pred * hip.to_device(T(g["cotangent"]))).sum().backward()
    close(pred, T(g["pred"]), 1e-4, "query_rgb forward")
    close(z.grad, T(g["grad_z"]), 1e-5, "d query_rgb / d zmap")


def test_latent_module_train_step_gradients_vs_oracle(hip, golden_dir):
    """linf/train.py::train_step on the HIP path (frozen model on the kernels, a small torch latent module on the GPU): loss and
    parameter gradients against the same objective written on the oracle under CPU autograd (LINF-LP/train.py:118-160)."""
    import oracle.linf_ref as O
    import torch.nn.functional as F
    from bfsr_amd.linf.models import make
    from bfsr_amd.linf.train import train_step
    g = np.load(os.path.join(golden_dir, "linf_vjp.npz"))
    sd, _ = weights("edsr-baseline", int(g["weights_seed"]))
    m = make(mspec("edsr-baseline"), args={"ops": hip}).eval()
    m.load_state_dict(sd)
    lr, coord, cell = T(g["lr"]), T(g["coord"]), T(g["cell"])
    H, W = 48, 40
    prep = O.batch_prep(lr, (H, W))
    gt = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(3))
    batch = dict(inp=lr, gt=gt, coord=coord, cell=cell, gt_lr_up=prep["gt_lr_up"], gt_patch=prep["gt_lr_up"] * 0.5 + 0.1)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(11)
            self.c = torch.nn.Conv2d(27, 27, 3, padding=1)

        def forward(self, z, inp):
            return z + 0.1 * self.c(z)

    prior = Tiny().to(hip.device)
    out = train_step(prior, m, batch, optimizer=None, latent_weight=0.7, image_weight=1.3)
    ref_prior = Tiny()
    espec = mspec("edsr-baseline")["args"]["encoder_spec"]
    inp = (lr - 0.5) / 0.5
    feat = O.encoder(inp, sd, espec)
    z_lr, z_hr = O.query_log_p(feat, coord, cell, batch["gt_lr_up"], sd), O.query_log_p(feat, coord, cell, batch["gt_patch"], sd)
    with torch.enable_grad():
        zl = ref_prior(z_lr, inp)
        pred = O.query_rgb(feat, coord, cell, zl, sd)[..., :H, :W] + F.interpolate(inp, (H, W), mode="bilinear", align_corners=False)
        loss = 1.3 * F.l1_loss(torch.clamp(pred * 0.5 + 0.5, 0, 1), gt) + 0.7 * F.l1_loss(zl, z_hr)
        loss.backward()
    assert abs(out["loss"] - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
    for a, b in zip(prior.parameters(), ref_prior.parameters()):
        err, ref = (a.grad.cpu() - b.grad).abs().max().item(), b.grad.abs().max().item()
        assert ref > 0 and err <= 1e-3 * ref, "latent-module parameter gradient: %.3e of %.3e" % (err, ref)


def test_train_step_vs_reference_train_golden_gpu(hip, golden_dir):
    """The training iteration on the HIP path (frozen model on the kernels incl. the `interpolate_coord` skip on
    bfsr_grid_sample_add, latent module + feature net in torch on the GPU) against the GENUINE reference's `train()` output
    (tests/golden/linf_train_step.npz, LINF-LP/train.py:88-172): losses within 1e-5 relative, gradients within 1e-3."""
    from test_linf_cpu import check_train_step, run_train_step_case
    out, prior, g = run_train_step_case(golden_dir, hip, device=hip.device)
    check_train_step(out, prior, g, 1e-5, 1e-3)
