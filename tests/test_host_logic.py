"""not gpu: host-side logic of the product -- checkpoint schema, config surface, registry, the engine's
restructured schedule (run on the CPU test double against the genuine reference's golden vectors), the C-ABI
library (loads, exports every declared symbol, host-side weight packing), and loud failure without a GPU."""
import json
import os
import re

import numpy as np
import pytest
import torch

from bfsr_amd import synth
from bfsr_amd.srflow import options, spec
from cpu_ops import CpuOps, CpuOpsX3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def test_schema_matches_reference_checkpoints(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "srflow_schema.json")))
    opt = options.load(options.DEFAULT_CONF)
    for tag, o in (("srflownet_4x", opt), ("srflownet_8x", options.derive_scale(opt, 8))):
        mine = [[k, list(s)] for k, (s, _) in spec.srflownet_schema(o).items()]
        assert mine == ref[tag]
    mine = [[k, list(s)] for k, (s, _) in spec.srflow_prior_schema().items()]
    assert mine == ref["prior_unet"]


def test_model_state_dict_roundtrip_and_registry():
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.models.networks import find_model_using_name
    opt = options.load(options.DEFAULT_CONF)
    assert find_model_using_name("SRFlowNet").__name__ == "SRFlowNet"
    m = create_model(opt, ops=CpuOps())
    sd = synth.state_dict_from_schema(spec.srflownet_schema(opt), 5)
    m.load_network({"module." + k: v for k, v in sd.items()})           # 'module.' prefix stripped (base_model.py:117-124)
    back = m.netG.module.state_dict()
    assert list(back.keys()) == list(sd.keys())
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    with pytest.raises(RuntimeError):
        m.netG.module.load_state_dict({k: v for k, v in list(sd.items())[:-1]}, strict=True)
    assert "unet" in registry.models
    p = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True}}, args={"ops": CpuOps()})
    assert list(p.state_dict().keys()) == list(spec.srflow_prior_schema().keys())
    assert m.get_z(0.9, seed=1, batch_size=2, lr_shape=(2, 3, 160, 160)).shape == (2, 96, 80, 80)
    assert float(m.get_z(0, batch_size=1, lr_shape=(1, 3, 16, 16)).abs().sum()) == 0.0


def test_options_surface():
    opt = options.load(options.DEFAULT_CONF)
    assert opt["network_G"]["flow"]["K"] == 16 and opt["missing_key"] is None
    assert options.opt_get(opt, ["network_G", "flow", "split", "enable"]) is True
    assert options.opt_get(opt, ["network_G", "nope", "x"], 7) == 7
    assert opt["network_G"]["scale"] == 4 and opt["is_train"] is False
    ls = spec.flow_layers(opt)
    assert len(ls) == 58 and sum(l.type == "step" for l in ls) == 54 and ls[19].type == "split"
    assert [l.C for l in ls if l.type == "squeeze"] == [12, 24, 96]


@pytest.mark.parametrize("fx,scale", [("srflow_e2e_4x_a", 4), ("srflow_e2e_4x_b", 4), ("srflow_e2e_8x", 8)])
def test_engine_schedule_on_cpu_double_vs_reference_golden(golden_dir, fx, scale):
    """The hoisted / deduplicated schedule reproduces the reference (fp32 re-association only)."""
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.test import lp_infer
    ops = CpuOps()
    opt = options.load(options.DEFAULT_CONF)
    if scale == 8:
        opt = options.derive_scale(opt, 8)
    m = create_model(opt, ops=ops)
    m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops},
                           "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
    g = np.load(os.path.join(golden_dir, fx + ".npz"))
    out = lp_infer(m, prior, T(g["lr"]), return_all=True)
    for i in (0, 1):
        assert (out["epses"][i] - T(g["eps%d" % i])).abs().max() <= 5e-5
        assert (out["epses_learned"][i] - T(g["epsl%d" % i])).abs().max() <= 5e-5
    assert (out["sr_raw"] - T(g["sr_raw"])).abs().max() <= 1e-4
    assert (out["sr"] - T(g["sr"])).abs().max() <= 1e-4
    # decode(encode(x)) == x and the caller's eps list is not consumed (decode copies then pops)
    eng = m.netG.module.engine()
    n = len(out["epses"])
    rt = eng.decode(T(g["lr"]), epses=out["epses"])
    assert len(out["epses"]) == n
    assert (rt - out["lr_up"]).abs().max() <= 1e-4


@pytest.mark.parametrize("fx,scale", [("srflow_e2e_4x_b", 4), ("srflow_e2e_8x", 8)])
def test_engine_schedule_x3_mode_on_cpu_double(golden_dir, fx, scale):
    """Host logic of the default (x3) schedule: RRDB dense blocks as x3 tensors (pack after conv_first, octet-sliced views,
    x3 residuals, unpack at the taps, fp32 trunk output) + conv_up2/up4 hoists, against the genuine reference's golden."""
    from bfsr_amd.srflow.models import create_model, models as registry
    from bfsr_amd.srflow.test import lp_infer
    ops = CpuOpsX3()
    opt = options.load(options.DEFAULT_CONF)
    if scale == 8:
        opt = options.derive_scale(opt, 8)
    m = create_model(opt, ops=ops)
    m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops},
                           "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
    g = np.load(os.path.join(golden_dir, fx + ".npz"))
    calls, orig = [], ops.conv_up2_h2t
    ops.conv_up2_h2t = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    out = lp_infer(m, prior, T(g["lr"]), return_all=True)
    assert m.netG.module.engine().rrdb.x3s
    assert len(calls) == 2, "the level whose taps are the x2-upsampled LR taps (level 1 of the 4x model, level 2 of the 8x model) runs conv_up2_h2t twice (fFeatures, fAffine)"
    for i in (0, 1):
        assert (out["epses"][i] - T(g["eps%d" % i])).abs().max() <= 5e-5
    assert (out["sr"] - T(g["sr"])).abs().max() <= 1e-4


def test_library_exports_every_declared_symbol():
    from bfsr_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "bfsr_hip.h")).read()
    declared = set(re.findall(r"\b(bfsr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS.keys())
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.bfsr_abi_version() == 8


def test_weight_packing_layout():
    from bfsr_amd import _lib
    lib = _lib.load()
    Cout, Cin, KS, mt = 70, 13, 3, 2
    w = torch.randn(Cout, Cin, KS, KS)
    n = lib.bfsr_conv_packed_size(Cout, Cin, KS, mt)
    cin_pad, groups = 16, 2
    assert n == groups * cin_pad * 9 * 64
    packed = torch.empty(n)
    assert lib.bfsr_pack_conv_weight(w.data_ptr(), Cout, Cin, KS, mt, packed.data_ptr()) == 0
    P = packed.view(groups, cin_pad, 9, 64)
    for co, ci, t in ((0, 0, 0), (69, 12, 8), (64, 3, 4), (33, 7, 2)):
        assert P[co // 64, ci, t, co % 64] == w[co, ci, t // 3, t % 3]
    assert float(P[1, :, :, 6:].abs().sum()) == 0 and float(P[:, 13:].abs().sum()) == 0     # zero padding
    assert lib.bfsr_pack_conv_weight(w.data_ptr(), Cout, Cin, 5, mt, packed.data_ptr()) != 0     # unsupported KS


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bfsr_amd.ops import HipOps
    from bfsr_amd.srflow.models import create_model
    with pytest.raises(RuntimeError):
        HipOps()
    m = create_model(options.load(options.DEFAULT_CONF))
    with pytest.raises(RuntimeError):
        m.get_sr(torch.rand(1, 3, 16, 16), epses=[torch.zeros(1, 6, 32, 32), torch.zeros(1, 96, 8, 8)])


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "bfsr_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_conditioning_cache_is_keyed_on_tensor_identity():
    """A fresh tensor that happens to reuse a freed tensor's storage must not hit the conditioning cache."""
    from bfsr_amd.srflow.engine import SRFlowEngine
    ops = CpuOps()
    opt = options.load(options.DEFAULT_CONF)
    eng = SRFlowEngine(opt, synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234), ops)
    a = synth.lr_batch(1, 1, 16, 16)
    c1 = eng.conditioning(a)
    assert eng.conditioning(a) is c1                       # same object, same version: hit
    h1 = c1[1]["h_ft"].clone()
    a.copy_(synth.lr_batch(2, 1, 16, 16))                  # in-place update bumps the version: miss
    c2 = eng.conditioning(a)
    assert not torch.equal(c2[1]["h_ft"], h1)
    b = a.clone()                                          # different object with equal content: recomputed, equal
    h2 = c2[1]["h_ft"].clone()
    assert torch.equal(eng.conditioning(b)[1]["h_ft"], h2)


@pytest.mark.parametrize("tag,scale", [("a", 4), ("c", 8)])
def test_nll_and_logdet_vs_reference_golden(golden_dir, tag, scale):
    """SRFlowNet.forward return values (epses, nll, logdet) / (sr, logdet) == the genuine reference's (srflow_logdet.npz),
    through the engine's schedule on the CPU double of the ops."""
    import torch.nn.functional as F
    from bfsr_amd.srflow.models import create_model
    ops = CpuOps()
    opt = options.load(options.DEFAULT_CONF)
    if scale == 8:
        opt = options.derive_scale(opt, 8)
    m = create_model(opt, ops=ops)
    m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
    net = m.netG.module
    g = np.load(os.path.join(golden_dir, "srflow_logdet.npz"))
    lr = T(g["lr_" + tag])
    lr_up = F.interpolate(lr, scale_factor=scale, mode="bilinear", align_corners=False)
    epses, nll, logdet = net(gt=lr_up, lr=lr, reverse=False, epses=[], add_gt_noise=False)
    assert nll.dtype == torch.float32 and logdet.shape == (lr.shape[0],)
    rel = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1.0)).max())
    assert rel(nll, T(g["nll_" + tag])) <= 1e-5
    assert rel(logdet, T(g["logdet_" + tag])) <= 1e-5
    sr, logdet_rev = net(lr=lr, reverse=True, epses=list(epses))
    assert rel(logdet_rev, T(g["logdet_rev_" + tag])) <= 1e-5
    assert rel(logdet_rev, -logdet) <= 1e-5                      # the inverse undoes every term


def test_split_packing_is_exact_and_laid_out_as_documented():
    """Host-side packers of the 16-bit kernels (no GPU needed): the 3xBF16 triple sums EXACTLY to the fp32 weight and sits at
    [group][chunk][plane][tap][k half][cout][8]; the fp16 / wide-1x1 packers round to nearest-even at the documented slots."""
    from bfsr_amd import _lib
    lib = _lib.load()
    Cout, Cin, KS, mt = 70, 37, 3, 2
    g = np.random.Generator(np.random.PCG64(5))
    w = torch.from_numpy((g.standard_normal((Cout, Cin, KS, KS)) * np.exp(g.uniform(-12, 4, (Cout, Cin, KS, KS)))).astype(np.float32))
    n = lib.bfsr_conv_packed_size_bf16x3(Cout, Cin, KS, mt)
    nchunk, groups = 3, 2
    assert n == groups * nchunk * 3 * 9 * 2 * 64 * 8
    packed = torch.zeros(n, dtype=torch.int16)
    assert lib.bfsr_pack_conv_weight_bf16x3(w.data_ptr(), Cout, Cin, KS, mt, packed.data_ptr()) == 0
    P = packed.view(groups, nchunk, 3, 9, 2, 64, 8)
    as_f32 = lambda t: (t.to(torch.int32) << 16).view(torch.float32)          # bf16 bits -> float
    full = torch.zeros(Cout, Cin, 9)
    for co in range(Cout):
        blk = P[co // 64, :, :, :, :, co % 64, :]                             # [chunk, plane, tap, khalf, 8]
        v = as_f32(blk).double().sum(1)                                       # h + m + l  -> [chunk, tap, khalf, 8]
        full[co] = v.permute(0, 2, 3, 1).reshape(nchunk * 16, 9)[:Cin].float()
    assert torch.equal(full.view(Cout, Cin, 3, 3), w)                         # exact, bit for bit
    assert int(P[1, :, :, :, :, 6:, :].abs().sum()) == 0                      # cout padding of the second group
    assert int(P[:, 2, :, :, 0, :, 5:].abs().sum()) == 0 and int(P[:, 2, :, :, 1].abs().sum()) == 0    # cin 37..47 padding
    # fp16 packer: RNE of the weight at [group][chunk][tap][k half][cout][8]
    n16 = lib.bfsr_conv_packed_size_f16(Cout, Cin, KS, mt)
    p16 = torch.zeros(n16, dtype=torch.int16)
    w16 = torch.randn(Cout, Cin, KS, KS)
    assert lib.bfsr_pack_conv_weight_f16(w16.data_ptr(), Cout, Cin, KS, mt, p16.data_ptr()) == 0
    Q = p16.view(groups, nchunk, 9, 2, 64, 8).view(torch.float16)
    for co, ci, t in ((0, 0, 0), (69, 36, 8), (64, 17, 4), (33, 7, 2)):
        assert Q[co // 64, ci // 16, t, (ci % 16) // 8, co % 64, ci % 8] == w16[co, ci, t // 3, t % 3].half()
    # wide 1x1 packers
    Co1, Ci1 = 300, 70
    w1 = torch.randn(Co1, Ci1)
    for x3, CK, PL in ((1, 32, 3), (0, 64, 1)):
        n1 = lib.bfsr_conv1x1_packed_size(Co1, Ci1, x3)
        nch = (Ci1 + CK - 1) // CK
        assert n1 == 2 * nch * PL * (CK // 8) * 256 * 8
        p1 = torch.zeros(n1, dtype=torch.int16)
        assert lib.bfsr_pack_conv1x1_weight(w1.data_ptr(), Co1, Ci1, x3, p1.data_ptr()) == 0
        R = p1.view(2, nch, PL, CK // 8, 256, 8)
        for co, ci in ((0, 0), (299, 69), (256, 33), (17, 64)):
            cell = R[co // 256, ci // CK, :, (ci % CK) // 8, co % 256, ci % 8]
            if x3:
                assert float(as_f32(cell).double().sum()) == float(w1[co, ci])
            else:
                assert cell.view(torch.float16)[0] == w1[co, ci].half()


def test_fp16_pair_packers_layout_and_accuracy():
    """Host-side packers of the two-term fp16 split (no GPU needed): hi + lo reproduces w * scale to 2^-22 relative (to 2^-25 * |w|max
    absolutely once the lo term is subnormal) at the documented slots of the register-staged kernels (taps layout, 2 planes), of
    conv3x3_h2x_kernel ([group][chunk][plane][tap = dx*3 + dy][m tile][k half][32][8]) and of its coupling-tail form."""
    from bfsr_amd import _lib
    from bfsr_amd.ops import HipOps
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(9))
    Cout, Cin = 70, 48
    w = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3)) * np.exp(g.uniform(-6, 1, (Cout, 1, 1, 1)))).astype(np.float32))
    scale = HipOps.pow2_scale(w)
    assert 512.0 <= float(w.abs().max()) * scale < 1024.0 and np.log2(scale) == int(np.log2(scale))
    f16 = lambda t: t.view(torch.float16).double()
    tol = lambda ref: 2.0 ** -22 * ref.abs() + 2.0 ** -24           # normal lo terms: 22 bits; subnormal lo terms: 2^-25 absolute
    # --- register-staged kernels: [group][chunk][plane][tap][k half][mtile*32][8]
    for mt in (1, 2):
        MW, groups, nchunk = 32 * mt, (Cout + 32 * mt - 1) // (32 * mt), Cin // 16
        n = lib.bfsr_conv_packed_size_taps_f16x2(Cout, Cin, 9, mt)
        assert n == groups * nchunk * 2 * 9 * 2 * MW * 8
        p = torch.zeros(n, dtype=torch.int16)
        assert lib.bfsr_pack_conv_weight_taps_f16x2(w.reshape(Cout, Cin, 9).contiguous().data_ptr(), Cout, Cin, 9, mt, scale, p.data_ptr()) == 0
        P = f16(p).view(groups, nchunk, 2, 9, 2, MW, 8)
        for co, ci, t in ((0, 0, 0), (69, 47, 8), (33, 17, 4), (64, 8, 2)):
            got = P[co // MW, ci // 16, :, t, (ci % 16) // 8, co % MW, ci % 8].sum()
            ref = w[co, ci, t // 3, t % 3].double() * scale
            assert abs(got - ref) <= tol(ref), (co, ci, t, float(got), float(ref))
    # --- conv3x3_h2x_kernel
    for mt in (1, 2):
        MW, groups, nchunk = 32 * mt, (Cout + 32 * mt - 1) // (32 * mt), Cin // 16
        n = lib.bfsr_conv_packed_size_h2x(Cout, Cin, mt)
        assert n == groups * nchunk * 2 * 9 * mt * 2 * 32 * 8
        p = torch.zeros(n, dtype=torch.int16)
        assert lib.bfsr_pack_conv_weight_h2x(w.data_ptr(), Cout, Cin, mt, scale, p.data_ptr()) == 0
        P = f16(p).view(groups, nchunk, 2, 9, mt, 2, 32, 8)
        for co, ci, dy, dx in ((0, 0, 0, 0), (69, 47, 2, 2), (33, 17, 1, 0), (64, 8, 0, 2)):
            got = P[co // MW, ci // 16, :, dx * 3 + dy, (co % MW) // 32, (ci % 16) // 8, co % 32, ci % 8].sum()
            ref = w[co, ci, dy, dx].double() * scale
            assert abs(got - ref) <= tol(ref), (co, ci, dy, dx)
        if mt == 1:                                                       # cout padding of the last group (couts 70..95) is zero
            assert float(P[-1, :, :, :, 0, :, 70 - 64:, :].abs().sum()) == 0.0
    # --- coupling tail (coupling_tail.hip): [octet][plane][tap = dx*3 + dy][32 rows][8] + one block of zeros.  Cout <= 16 -> the
    #     two-instruction form: plane 0 = [rows 0-15: hi | rows 16-31: lo], plane 1 = [rows 0-15: hi | 0]; Cout > 16 -> plane 0 = hi, plane 1 = lo
    w4 = torch.randn(12, 64, 3, 3) * 0.01
    s4 = HipOps.pow2_scale(w4)
    n = lib.bfsr_coupling_tail_packed_size(64, 12)
    assert n == 8 * 2 * 9 * 32 * 8 + 32 * 8 and lib.bfsr_coupling_tail_packed_size(48, 12) < 0 and lib.bfsr_coupling_tail_packed_size(64, 40) < 0
    p = torch.zeros(n, dtype=torch.int16)
    assert lib.bfsr_pack_coupling_tail(w4.data_ptr(), 64, 12, s4, p.data_ptr()) == 0
    assert float(f16(p[-256:]).abs().sum()) == 0.0                       # the zero block (tenth tap)
    P = f16(p[:-256]).view(8, 2, 9, 32, 8)
    for co, ci, dy, dx in ((0, 0, 0, 0), (11, 47, 2, 2), (5, 17, 1, 0)):
        hi0, lo0, hi1 = P[ci // 8, 0, dx * 3 + dy, co, ci % 8], P[ci // 8, 0, dx * 3 + dy, 16 + co, ci % 8], P[ci // 8, 1, dx * 3 + dy, co, ci % 8]
        ref = w4[co, ci, dy, dx].double() * s4
        assert hi0 == hi1 and abs((hi0 + lo0) - ref) <= tol(ref), (co, ci, dy, dx)
        assert float(P[ci // 8, 1, dx * 3 + dy, 16 + co, ci % 8]) == 0.0
    assert float(P[:, :, :, 12:16].abs().sum()) == 0.0 and float(P[:, :, :, 28:].abs().sum()) == 0.0
    w4b = torch.randn(24, 64, 3, 3) * 0.01
    assert lib.bfsr_pack_coupling_tail(w4b.data_ptr(), 64, 24, s4, p.data_ptr()) == 0
    P = f16(p[:-256]).view(8, 2, 9, 32, 8)
    for co, ci, dy, dx in ((0, 0, 0, 0), (23, 47, 2, 2), (5, 17, 1, 0)):
        ref = w4b[co, ci, dy, dx].double() * s4
        assert abs(P[ci // 8, :, dx * 3 + dy, co, ci % 8].sum() - ref) <= tol(ref)
    # --- fused MLP, per-layer scales
    HD, Co4 = 256, 72
    ws = [torch.randn(HD, 4 * HD) * 0.03, torch.randn(HD, HD) * 0.06, torch.randn(HD, HD) * 0.5, torch.randn(Co4, HD) * 2.0]
    sc = [HipOps.pow2_scale(t) for t in ws]
    import ctypes as C
    n = lib.bfsr_linf_mlp_packed_size(HD, Co4, 2)
    p = torch.zeros(n, dtype=torch.int16)
    arr = (C.c_float * 4)(*sc)
    assert lib.bfsr_pack_linf_mlp_f16x2(ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ws[3].data_ptr(), HD, Co4, C.cast(arr, C.c_void_p), p.data_ptr()) == 0
    o1 = (HD // 32) * (4 * HD // 16) * 2 * 64 * 8                          # layer 2 starts here: [m tile][k chunk][plane][lane][8]
    L2 = f16(p[o1: o1 + (HD // 32) * (HD // 16) * 2 * 64 * 8]).view(HD // 32, HD // 16, 2, 64, 8)
    for row, kidx in ((0, 0), (255, 255), (100, 77)):
        got = L2[row // 32, kidx // 16, :, ((kidx % 16) // 8) * 32 + row % 32, kidx % 8].sum()
        ref = ws[1][row, kidx].double() * sc[1]
        assert abs(got - ref) <= tol(ref)
    assert lib.bfsr_pack_linf_mlp(ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ws[3].data_ptr(), HD, Co4, 2, p.data_ptr()) != 0     # needs the scales


def test_pad_lr_to_even_on_tensor_equals_numpy_reflect():
    """test.py:126-130's np.pad(..., 'reflect') up to even H and W, done by slicing on the uploaded tensor"""
    from bfsr_amd.srflow.test import pad_lr_to_even, pad_lr_to_even_t
    g = np.random.Generator(np.random.PCG64(4))
    for h, w in ((7, 9), (8, 5), (6, 4), (3, 3)):
        lr = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = pad_lr_to_even(lr).transpose(2, 0, 1)[None].astype(np.float32) / 255
        got = pad_lr_to_even_t(torch.from_numpy(lr.transpose(2, 0, 1)[None].astype(np.float32)) / 255)
        assert got.shape == ref.shape and np.array_equal(got.numpy(), ref)


def test_cli_writes_images_and_measures_csv(tmp_path, monkeypatch):
    """`python -m bfsr_amd.srflow.test conf.yml` on two tiny images (odd sizes -> reflect pad) with the CPU double: SR PNGs cropped to
    scale * (h, w) and measure_full.csv with the reference's columns (test.py:150-171); PSNR / SSIM rows equal Measure's own values."""
    import pandas as pd
    from PIL import Image
    from bfsr_amd.srflow import test as cli
    from bfsr_amd.srflow.Measure import Measure
    from bfsr_amd.srflow.models import create_model, models as registry
    ops = CpuOpsX3()
    opt = options.load(options.DEFAULT_CONF)
    m = create_model(opt, ops=ops)
    m.load_network(synth.state_dict_from_schema(spec.srflownet_schema(opt), 1234))
    prior = registry.make({"name": "unet", "args": {"depth": 3, "dim": 64, "bilinear": True, "ops": ops},
                           "sd": synth.state_dict_from_schema(spec.srflow_prior_schema(), 4321)}, load_sd=True).eval()
    lr_dir, hr_dir, conf_dir = tmp_path / "lr", tmp_path / "hr", tmp_path / "confs"
    for d in (lr_dir, hr_dir, conf_dir):
        d.mkdir()
    g = np.random.Generator(np.random.PCG64(2))
    sizes = [(17, 20), (18, 19)]
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(str(lr_dir / ("%d.png" % i)))
        Image.fromarray(g.integers(0, 256, size=(4 * h, 4 * w, 3), dtype=np.uint8)).save(str(hr_dir / ("%d.png" % i)))
    opt["dataroot_LR"], opt["dataroot_GT"] = str(lr_dir), str(hr_dir)
    monkeypatch.setattr(cli, "load_model", lambda path: (m, opt))
    monkeypatch.setattr(cli, "load_prior", lambda o: prior)
    cli.main([str(conf_dir / "SRFlow-LP_test.yml")])
    out = tmp_path / "results" / "SRFlow-LP"
    df = pd.read_csv(str(out / "measure_full.csv"))
    assert list(df.columns) == ["conf", "name", "PSNR", "SSIM", "LPIPS", "LRC PSNR"] and len(df) == 2 and set(df["name"]) == {0, 1}
    me = Measure(ops)
    for i, (h, w) in enumerate(sizes):
        sr = np.asarray(Image.open(str(out / ("%06d.png" % i))))
        assert sr.shape == (4 * h, 4 * w, 3)
        hr = np.asarray(Image.open(str(hr_dir / ("%d.png" % i))))
        row = df[df["name"] == i].iloc[0]
        assert abs(row["PSNR"] - me.psnr(sr, hr)) <= 1e-9 and abs(row["SSIM"] - me.ssim(sr, hr)) <= 1e-9 and np.isnan(row["LPIPS"])
        assert 5.0 < row["LRC PSNR"] < 100.0


def test_no_wide_buffer_store_with_register_soffset():
    """The gfx950 store-data hazard of DESIGN.md section 3 item 8, as a build check: a >64-bit buffer store whose soffset is a register is exempt
    from hipcc's hazard padding, and on this chip a VALU write of a data register in the next issue slot leaks into the store.  Disassemble
    every gfx950 code object of the built library and require the literal soffset (bfsr::store_b128 in launch_util.h) on all such stores."""
    import struct
    import subprocess
    import tempfile
    from bfsr_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    _lib.load()
    path = os.environ.get("BFSR_HIP_LIB") or os.path.join(ROOT, "bfsr_amd", "lib", "libbfsr_hip.so")
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    objs = wide = 0
    bad = []
    for m in re.finditer(re.escape(magic), data):
        b0 = m.start()
        cnt, = struct.unpack_from("<Q", data, b0 + 24)
        p = b0 + 32
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" not in triple or not size:
                continue
            objs += 1
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[b0 + off: b0 + off + size])
                f.flush()
                dis = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True, check=True).stdout
            for line in dis.splitlines():
                mm = re.search(r"buffer_store_dwordx[34]\s+(.*)", line)
                if mm:
                    wide += 1
                    soffset = [o.strip() for o in mm.group(1).split(",")][3].split()[0]      # vdata, vaddr, srsrc, soffset [modifiers]
                    if re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi|ttmp\d+", soffset):
                        bad.append(line.strip())
    assert objs >= 10 and wide >= 50, "the disassembly did not find the kernels (%d code objects, %d wide buffer stores)" % (objs, wide)
    assert not bad, "wide buffer stores with a register soffset: %s" % bad[:4]


def test_library_has_no_packed_fp32_instructions(tmp_path):
    """The device code must be built without packed-fp32 VALU instructions (bfsr_amd/csrc/build.sh, NOPK): on MI355X their swizzled form (`v_pk_add_f32 ...
    op_sel`) returns wrong values when the wave shares a SIMD with MFMA waves of another kernel (two streams; DESIGN.md section 5 round 5,
    profiles/r05_packed_fp32_hazard.txt).  Disassembles every code object of the built library; skipped when the LLVM tools of the ROCm image are absent."""
    import re
    import shutil
    import subprocess
    from bfsr_amd import _lib
    bundler, objdump = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(bundler) and os.path.exists(objdump) and shutil.which("objcopy")):
        pytest.skip("LLVM / binutils tools not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    assert len(starts) >= 10, "expected one offload bundle per kernel source"
    packed = mfma = 0
    for n, i in enumerate(starts):
        one, co = str(tmp_path / "b.bin"), str(tmp_path / "dev.co")
        open(one, "wb").write(data[i: starts[n + 1] if n + 1 < len(starts) else len(data)])
        subprocess.check_call([bundler, "--type=o", "--input=" + one, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co, "--unbundle"])
        asm = subprocess.run([objdump, "-d", co], capture_output=True, text=True, check=True).stdout
        packed += len(re.findall(r"v_pk_\w+_f32", asm))
        mfma += asm.count("v_mfma_")
    assert mfma > 1000, "disassembly looks empty"
    assert packed == 0, "%d packed-fp32 instructions in libbfsr_hip.so: build it with bfsr_amd/csrc/build.sh" % packed



def test_bench_evidence_readers_accept_every_committed_file():
    """bench.py reads committed evidence (profiles/rNN*_pmc_traffic*.json, *_kernel_stats.csv) for `roofline.traffic` / `frac_rocprof`: every committed file must parse
    (an empty JSON once took the whole benchmark down), the readers must survive a broken one, and files are taken in the order they were produced (r06z before r06aa)."""
    import glob
    import importlib.util
    import json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    for p in glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]*_pmc_traffic*.json")):
        assert isinstance(json.load(open(p)), dict), p          # (round-1 files have another schema and no "kernels": load_traffic takes them as empty)
    spec_ = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(b)
    assert len(b.load_traffic()) > 0
    for cfg in (2, 3, 4, 5):
        ms, src = b.rocprof_avg_ms(cfg, "conv_chain_kernel")
        assert ms is None or ms > 0
    order = sorted(["profiles/r06aa_x.json", "profiles/r06z_x.json", "profiles/r05g_x.json", "profiles/r06b_x.json"], key=b._evidence_order)
    assert [os.path.basename(o).split("_")[0] for o in order] == ["r05g", "r06b", "r06z", "r06aa"]
