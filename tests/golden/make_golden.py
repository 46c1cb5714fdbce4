"""Emit golden vectors from the GENUINE reference (imported from /root/reference; build container
only -- see ref_import.py).  Usage:  python tests/golden/make_golden.py [srflow|linf|all]

Every fixture is data: seeded inputs (or seeds) and the reference's outputs.  Large weight sets
are not stored: they are regenerated from `bfsr_amd.synth` seeds (the sha256 of the state_dict is
stored so a mismatch on another machine is detectable).  While generating, the oracle
(`oracle/*.py`) is checked against the reference and the max-abs differences are written to
MANIFEST.json -- that is what "oracle pinned" means for this repo.
"""
import importlib
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import as R  # noqa: E402

torch.set_grad_enabled(False)
MANIFEST = {}


def npy(t):
    return t.detach().cpu().numpy()


def maxdiff(a, b):
    return float((a - b).abs().max())


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def rnd(seed, *shape, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


# ------------------------------------------------------------------------------------------
def gen_srflow():
    from bfsr_amd import synth
    from bfsr_amd.srflow import options, spec
    import oracle.srflow_ref as O

    man = MANIFEST.setdefault("srflow", {})
    opt_ref = R.srflow_opt(4)
    opt = options.load(options.DEFAULT_CONF)

    # ---- schema (checkpoint contract) ----
    net = R.build_srflownet(opt_ref)
    ref_sd = net.state_dict()
    unet_mod = importlib.import_module("models.unet")
    prior = unet_mod.make_unet(depth=3, dim=64, bilinear=True).eval()
    schema = {
        "srflownet_4x": [[k, list(v.shape)] for k, v in ref_sd.items()],
        "prior_unet": [[k, list(v.shape)] for k, v in prior.state_dict().items()],
    }
    sch = spec.srflownet_schema(opt)
    assert [k for k, _ in schema["srflownet_4x"]] == list(sch.keys())
    sd = synth.state_dict_from_schema(sch, 1234)
    net.load_state_dict(sd, strict=True)          # acceptance: synthetic sd strict-loads
    psch = spec.srflow_prior_schema()
    psd = synth.state_dict_from_schema(psch, 4321)
    prior.load_state_dict(psd, strict=True)

    # ---- per-op goldens from the reference's own modules ----
    ops = OrderedDict()
    FlowActNorms = importlib.import_module("models.modules.FlowActNorms")
    Permutations = importlib.import_module("models.modules.Permutations")
    flowmod = importlib.import_module("models.modules.flow")
    Split = importlib.import_module("models.modules.Split")

    # actnorm
    an = FlowActNorms.ActNorm2d(12)
    an.inited = True
    an.bias.data = rnd(1, 1, 12, 1, 1, scale=0.3)
    an.logs.data = rnd(2, 1, 12, 1, 1, scale=0.3)
    x = rnd(3, 2, 12, 5, 7)
    y, _ = an(x, None, reverse=False)
    xr, _ = an(x, None, reverse=True)
    ops.update(actnorm_bias=an.bias.data, actnorm_logs=an.logs.data, actnorm_x=x, actnorm_fwd=y, actnorm_rev=xr)
    man["actnorm"] = max(maxdiff(O.actnorm(x, an.bias.data, an.logs.data, False), y),
                         maxdiff(O.actnorm(x, an.bias.data, an.logs.data, True), xr))

    # invconv C=12/24/96
    for C in (12, 24, 96):
        np.random.seed(C)
        ic = Permutations.InvertibleConv1x1(C)
        ic.weight.data = ic.weight.data + rnd(10 + C, C, C, scale=0.05)     # not exactly orthogonal
        x = rnd(20 + C, 2, C, 4, 6)
        y, _ = ic(x, None, reverse=False)
        xr, _ = ic(x, None, reverse=True)
        ops["invconv%d_w" % C] = ic.weight.data
        ops["invconv%d_x" % C] = x
        ops["invconv%d_fwd" % C] = y
        ops["invconv%d_rev" % C] = xr
        man["invconv%d" % C] = max(maxdiff(O.invconv(x, ic.weight.data, False), y),
                                   maxdiff(O.invconv(x, ic.weight.data, True), xr))

    # flow.Conv2d 3x3 / 1x1, Conv2dZeros
    for k in (3, 1):
        c = flowmod.Conv2d(7, 10, kernel_size=[k, k])
        c.actnorm.inited = True
        c.weight.data = rnd(30 + k, 10, 7, k, k, scale=0.2)
        c.actnorm.bias.data = rnd(31 + k, 1, 10, 1, 1, scale=0.2)
        c.actnorm.logs.data = rnd(32 + k, 1, 10, 1, 1, scale=0.2)
        x = rnd(33 + k, 2, 7, 6, 5)
        y = c(x)
        ops.update({"fconv%d_w" % k: c.weight.data, "fconv%d_b" % k: c.actnorm.bias.data,
                    "fconv%d_logs" % k: c.actnorm.logs.data, "fconv%d_x" % k: x, "fconv%d_y" % k: y})
        sdl = {"c.weight": c.weight.data, "c.actnorm.bias": c.actnorm.bias.data, "c.actnorm.logs": c.actnorm.logs.data}
        man["flow_conv2d_%d" % k] = maxdiff(O.flow_conv2d(x, sdl, "c", k), y)
    cz = flowmod.Conv2dZeros(7, 8)
    cz.weight.data = rnd(40, 8, 7, 3, 3, scale=0.2)
    cz.bias.data = rnd(41, 8, scale=0.2)
    cz.logs.data = rnd(42, 8, 1, 1, scale=0.2)
    x = rnd(43, 2, 7, 6, 5)
    y = cz(x)
    ops.update(czero_w=cz.weight.data, czero_b=cz.bias.data, czero_logs=cz.logs.data, czero_x=x, czero_y=y)
    man["conv2d_zeros"] = maxdiff(O.conv2d_zeros(x, {"c.weight": cz.weight.data, "c.bias": cz.bias.data,
                                                     "c.logs": cz.logs.data}, "c"), y)

    # squeeze / unsqueeze
    x = rnd(50, 2, 3, 6, 8)
    ys = flowmod.squeeze2d(x, 2)
    ops.update(squeeze_x=x, squeeze_y=ys, unsqueeze_y=flowmod.unsqueeze2d(ys, 2))
    man["squeeze"] = max(maxdiff(O.squeeze2d(x), ys), maxdiff(O.unsqueeze2d(ys), x))

    # Split2d fwd/rev
    sp = Split.Split2d(num_channels=12, opt=opt_ref)
    sp.conv.weight.data = rnd(60, 12, 6, 3, 3, scale=0.1)
    sp.conv.bias.data = rnd(61, 12, scale=0.1)
    sp.conv.logs.data = rnd(62, 12, 1, 1, scale=0.1)
    x = rnd(63, 2, 12, 6, 6)
    z1, _, e = sp(x, 0.0, reverse=False, eps=None, ft=None)
    zr, _ = sp(z1, 0.0, reverse=True, eps=e, ft=None)
    ops.update(split_w=sp.conv.weight.data, split_b=sp.conv.bias.data, split_logs=sp.conv.logs.data,
               split_x=x, split_z1=z1, split_eps=e, split_rev=zr)
    sdl = {"s.conv.weight": sp.conv.weight.data, "s.conv.bias": sp.conv.bias.data, "s.conv.logs": sp.conv.logs.data}
    oz1, oe = O.split2d(x, sdl, "s", 6, False)
    man["split2d"] = max(maxdiff(oz1, z1), maxdiff(oe, e), maxdiff(O.split2d(z1, sdl, "s", 6, True, eps=e), zr))

    # eps standardisation (test.py:141-145)
    e = rnd(70, 2, 6, 5, 5, scale=2.0) + 0.3
    m = torch.mean(e, dim=[1], keepdim=True)
    s = torch.std(e, dim=[1], keepdim=True)
    en = (e - m) / (s + 1e-8)
    ops.update(std_x=e, std_y=en)
    man["standardize"] = maxdiff(O.standardize_eps(e), en)
    save("srflow_ops.npz", **ops)

    # ---- one coupled FlowStep per level (weights regenerated from the model seed) ----
    steps = OrderedDict()
    layers = spec.flow_layers(opt)
    for li in (3, 23, 42):                       # first coupled step of L1 / L2 / L3
        ly = layers[li]
        mod = net.flowUpsamplerNet.layers[li]
        z = rnd(100 + li, 1, ly.C, 6, 8)
        ft = rnd(200 + li, 1, 320, 6, 8, scale=0.5)
        y, _ = mod(z, torch.zeros(1), reverse=False, rrdbResults=ft)
        x, _ = mod(z, torch.zeros(1), reverse=True, rrdbResults=ft)
        steps["step%d_z" % li] = z
        steps["step%d_ft" % li] = ft
        steps["step%d_fwd" % li] = y
        steps["step%d_rev" % li] = x
        p = "flowUpsamplerNet.layers.%d" % li
        man["flowstep_%d" % li] = max(maxdiff(O.flow_step(z, ft, sd, p, True, False), y),
                                      maxdiff(O.flow_step(z, ft, sd, p, True, True), x))
    li = 1                                        # a noCoupling step
    mod = net.flowUpsamplerNet.layers[li]
    z = rnd(101, 1, 12, 6, 8)
    y, _ = mod(z, torch.zeros(1), reverse=False, rrdbResults=None)
    x, _ = mod(z, torch.zeros(1), reverse=True, rrdbResults=None)
    steps.update(step1_z=z, step1_fwd=y, step1_rev=x)
    steps["weights_seed"] = np.int64(1234)
    steps["weights_sha256"] = np.frombuffer(synth.digest(sd).encode(), dtype=np.uint8)
    save("srflow_steps.npz", **steps)

    # ---- RRDB preprocessing on a tiny LR (nb=23) ----
    lr = synth.lr_batch(5, 1, 8, 10)
    res = net.rrdbPreprocessing(lr)
    ores = O.rrdb_preprocessing(lr, sd, opt, 23)
    keys = ["fea_up2", "fea_up1", "fea_up0"]
    man["rrdb_preprocessing"] = max(maxdiff(res[k], ores[k]) for k in keys)
    save("srflow_rrdb.npz", lr=lr, **{k.replace("-", "m"): res[k] for k in keys})

    # ---- prior UNet ----
    e0 = rnd(300, 1, 6, 32, 40)
    e1 = rnd(301, 1, 96, 8, 10)
    out = prior([e0, e1])
    oout = O.srflow_prior([e0, e1], psd, 3)
    man["prior_unet"] = max(maxdiff(out[0], oout[0]), maxdiff(out[1], oout[1]))
    # odd sizes exercise the pad in Up (unet.py:86-92)
    e0b = rnd(302, 1, 6, 20, 12)
    e1b = rnd(303, 1, 96, 10, 12)
    outb = prior([e0b, e1b])
    ooutb = O.srflow_prior([e0b, e1b], psd, 3)
    man["prior_unet_pad"] = max(maxdiff(outb[0], ooutb[0]), maxdiff(outb[1], ooutb[1]))
    save("srflow_prior.npz", e0=e0, e1=e1, z0=out[0], z1=out[1], e0b=e0b, e1b=e1b, z0b=outb[0], z1b=outb[1],
         weights_seed=np.int64(4321), weights_sha256=np.frombuffer(synth.digest(psd).encode(), dtype=np.uint8))

    # ---- end-to-end LP pipeline (test.py:126-151) ----
    def run_ref(net_, scale, lr):
        lr_up = F.interpolate(lr, scale_factor=scale, mode="bilinear", align_corners=False)
        epses = []
        net_(gt=lr_up, lr=lr, reverse=False, epses=epses, add_gt_noise=False)
        ep = [e.detach() for e in epses]
        epn = []
        for e in ep:
            m = torch.mean(e, dim=[1], keepdim=True)
            s = torch.std(e, dim=[1], keepdim=True)
            epn.append((e - m) / (s + 1e-8))
        epl = prior(epn)
        sr, _ = net_(lr=lr, z=None, eps_std=None, reverse=True, epses=epl, reverse_with_grad=True)
        rt, _ = net_(lr=lr, z=None, eps_std=None, reverse=True, epses=ep, reverse_with_grad=True)
        return dict(lr=lr, eps0=ep[0], eps1=ep[1], epsn0=epn[0], epsn1=epn[1], epsl0=epl[0], epsl1=epl[1],
                    sr_raw=sr, sr=torch.clamp(sr, 0, 1), roundtrip=rt)

    for tag, lr in (("a", synth.lr_batch(0, 2, 16, 16)), ("b", synth.smooth_lr_batch(1, 1, 16, 24))):
        g = run_ref(net, 4, lr)
        o = O.lp_pipeline(lr, sd, psd, opt, 23, return_all=True)
        man["e2e_4x_%s" % tag] = dict(eps=max(maxdiff(o["epses"][i], g["eps%d" % i]) for i in (0, 1)),
                                      sr_raw=maxdiff(o["sr_raw"], g["sr_raw"]),
                                      roundtrip_vs_lr_up=maxdiff(g["roundtrip"], o["lr_up"]))
        save("srflow_e2e_4x_%s.npz" % tag, weights_seed=np.int64(1234), prior_seed=np.int64(4321),
             weights_sha256=np.frombuffer(synth.digest(sd).encode(), dtype=np.uint8), **g)

    # 8x variant (scale: 8, L: 3): the stack is conditioned on fea_up4/2/1, RRDB gains upconv3
    opt8_ref = R.srflow_opt(8)
    net8 = R.build_srflownet(opt8_ref)
    opt8 = options.derive_scale(opt, 8)
    sch8 = spec.srflownet_schema(opt8)
    ref8 = net8.state_dict()
    assert list(ref8.keys()) == list(sch8.keys())
    for k, v in ref8.items():
        assert tuple(v.shape) == tuple(sch8[k][0]), k
    schema["srflownet_8x"] = [[k, list(v.shape)] for k, v in ref8.items()]
    sd8 = synth.state_dict_from_schema(sch8, 1234)
    net8.load_state_dict(sd8, strict=True)
    lr = synth.lr_batch(2, 1, 8, 12)
    g = run_ref(net8, 8, lr)
    o = O.lp_pipeline(lr, sd8, psd, opt8, 23, return_all=True)
    man["e2e_8x"] = dict(eps=max(maxdiff(o["epses"][i], g["eps%d" % i]) for i in (0, 1)),
                         sr_raw=maxdiff(o["sr_raw"], g["sr_raw"]),
                         roundtrip_vs_lr_up=maxdiff(g["roundtrip"], o["lr_up"]))
    save("srflow_e2e_8x.npz", weights_seed=np.int64(1234), prior_seed=np.int64(4321),
         weights_sha256=np.frombuffer(synth.digest(sd8).encode(), dtype=np.uint8), **g)

    with open(os.path.join(HERE, "srflow_schema.json"), "w") as f:
        json.dump(schema, f)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if not R.reference_available():
        raise SystemExit("reference not available at %s" % R.REF_ROOT)
    mpath = os.path.join(HERE, "MANIFEST.json")
    if os.path.exists(mpath):
        MANIFEST.update(json.load(open(mpath)))
    if what in ("srflow", "all"):
        gen_srflow()
    if what in ("linf", "all"):
        from make_golden_linf import gen_linf
        gen_linf(MANIFEST)
    MANIFEST["torch"] = torch.__version__
    MANIFEST["note"] = ("max-abs difference oracle vs genuine reference at generation time; "
                        "0.0 = bit-identical")
    with open(mpath, "w") as f:
        json.dump(MANIFEST, f, indent=1, sort_keys=True)
    print(json.dumps(MANIFEST, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
