"""ORACLE (test infrastructure, not product): CPU restatement of the SRFlow-LP hot path.

Plain torch-CPU fp32 ops, written from the reference's behaviour, in the reference's own op
order and with its redundancies (RRDB recomputed in decode, `inverse` per call, 320-ch concat
every step).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the product (`bfsr_amd/`) never does.

Parity status: PINNED by golden vectors emitted from the genuine reference imported in the
build container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`); the reference itself has
no tests / known-answer vectors, so those fixtures are the only pin.

All citations are `path:line` under /root/reference/SRFlow-LP/code/.
State-dict key names are the reference's.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# config helpers
# --------------------------------------------------------------------------------------------
def opt_get(opt, keys, default=None):
    """utils/util.py:167-175"""
    if opt is None:
        return default
    ret = opt
    for k in keys:
        ret = ret.get(k, None) if hasattr(ret, "get") else None
        if ret is None:
            return default
    return ret


def level_to_name(scale):
    """models/modules/FlowUpsamplerNet.py:49-74"""
    if scale == 16:
        return {0: "fea_up16", 1: "fea_up8", 2: "fea_up4", 3: "fea_up2", 4: "fea_up1"}
    if scale == 8:
        return {0: "fea_up8", 1: "fea_up4", 2: "fea_up2", 3: "fea_up1", 4: "fea_up0"}
    if scale == 4:
        return {0: "fea_up4", 1: "fea_up2", 2: "fea_up1", 3: "fea_up0", 4: "fea_up-1"}
    raise ValueError(scale)


def build_layers(opt):
    """Layer list as constructed by FlowUpsamplerNet.__init__ (FlowUpsamplerNet.py:30-115) for a
    (160,160,3) image: per level [squeeze, additionalFlowNoAffine x noCoupling step, K x coupled
    step, optional Split2d].  Returns list of dicts {type, C, size} with `size` = H recorded at
    construction (used for level = log2(160/size), FlowUpsamplerNet.py:230,280)."""
    flow = opt["network_G"]["flow"]
    L = flow["L"]
    K = flow["K"]
    Ks = [K] * (L + 1) if isinstance(K, int) else list(K)
    n_add = int(flow.get("additionalFlowNoAffine", 0) or 0)
    split_on = bool(opt_get(opt, ["network_G", "flow", "split", "enable"]))
    correct_splits = opt_get(opt, ["network_G", "flow", "split", "correct_splits"], False)
    correction = 0 if correct_splits else 1
    consume_ratio = opt_get(opt, ["network_G", "flow", "split", "consume_ratio"]) or 0.5
    H, C = 160, 3
    layers = []
    for level in range(1, L + 1):
        C, H = C * 4, H // 2                                   # arch_squeeze :183-187
        layers.append(dict(type="squeeze", C=C, size=H))
        for _ in range(n_add):                                 # arch_additionalFlowAffine :169-181
            layers.append(dict(type="step", C=C, size=H, coupled=False))
        for _ in range(Ks[level]):                             # arch_FlowStep :122-142
            layers.append(dict(type="step", C=C, size=H, coupled=True))
        if split_on and level < L - correction:                # arch_split :149-167
            consume = int(round(C * consume_ratio))
            layers.append(dict(type="split", C=C, size=H, C_pass=C - consume, C_consume=consume))
            C = C - consume
    return layers


# --------------------------------------------------------------------------------------------
# primitive ops (Appendix B of SURVEY.md)
# --------------------------------------------------------------------------------------------
def actnorm(x, bias, logs, reverse):
    """FlowActNorms.py:61-113: fwd (x+b)*exp(logs); rev x*exp(-logs) - b."""
    if not reverse:
        return (x + bias) * torch.exp(logs)
    return x * torch.exp(-logs) - bias


def invconv(x, weight, reverse):
    """Permutations.py:34-58: rev uses inverse(W.double()).float() on every call."""
    C = weight.shape[0]
    if not reverse:
        w = weight.view(C, C, 1, 1)
    else:
        # `.float()` in the reference; `.to(x.dtype)` is the same for fp32 inputs and lets the tests run this restatement in
        # fp64 (weights and inputs cast to double) as the ground truth for error measurements
        w = torch.inverse(weight.double()).to(x.dtype).view(C, C, 1, 1)
    return F.conv2d(x, w)


def flow_conv2d(x, sd, p, k):
    """flow.py:26-65 Conv2d: conv(no bias, 'same') then ActNorm2d forward."""
    y = F.conv2d(x, sd[p + ".weight"], None, 1, (k - 1) // 2)
    return actnorm(y, sd[p + ".actnorm.bias"], sd[p + ".actnorm.logs"], False)


def conv2d_zeros(x, sd, p):
    """flow.py:68-83 Conv2dZeros: (conv3x3(x)+bias) * exp(logs*3)."""
    y = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], 1, 1)
    return y * torch.exp(sd[p + ".logs"] * 3)


def coupling_net(x, sd, p):
    """FlowAffineCouplingsAblation.py:127-135 F(): Conv2d 3x3 -> ReLU -> Conv2d 1x1 -> ReLU ->
    Conv2dZeros 3x3."""
    h = F.relu(flow_conv2d(x, sd, p + ".0", 3))
    h = F.relu(flow_conv2d(h, sd, p + ".2", 1))
    return conv2d_zeros(h, sd, p + ".4")


def cross_split(h):
    """thops.py:52-60 'cross'."""
    return h[:, 0::2], h[:, 1::2]


def _sum123(t):
    """thops.sum(t, dim=[1,2,3]) (thops.py:4-18: the dims are reduced one at a time, last first)."""
    for d in (3, 2, 1):
        t = t.sum(dim=d)
    return t


def coupling(z, ft, sd, p, reverse, eps=1e-4, ld=None):
    """CondAffineSeparatedAndCond.forward, FlowAffineCouplingsAblation.py:57-97.
    `ld` (optional one-element list) carries the running logdet [B]: += sum log(scale) forward (:66,:75), -= reverse (:86,:92)."""
    C = z.shape[1]
    cn = C // 2
    if not reverse:
        shiftFt, scaleFt = cross_split(coupling_net(ft, sd, p + ".fFeatures"))
        scaleFt = torch.sigmoid(scaleFt + 2.0) + eps
        z = z + shiftFt
        z = z * scaleFt
        if ld is not None:
            ld[0] = ld[0] + _sum123(torch.log(scaleFt))
        z1, z2 = z[:, :cn], z[:, cn:]
        shift, scale = cross_split(coupling_net(torch.cat([z1, ft], 1), sd, p + ".fAffine"))
        scale = torch.sigmoid(scale + 2.0) + eps
        z2 = z2 + shift
        z2 = z2 * scale
        if ld is not None:
            ld[0] = ld[0] + _sum123(torch.log(scale))
        return torch.cat((z1, z2), 1)
    z1, z2 = z[:, :cn], z[:, cn:]
    shift, scale = cross_split(coupling_net(torch.cat([z1, ft], 1), sd, p + ".fAffine"))
    scale = torch.sigmoid(scale + 2.0) + eps
    z2 = z2 / scale
    z2 = z2 - shift
    z = torch.cat((z1, z2), 1)
    if ld is not None:
        ld[0] = ld[0] - _sum123(torch.log(scale))
    shiftFt, scaleFt = cross_split(coupling_net(ft, sd, p + ".fFeatures"))
    scaleFt = torch.sigmoid(scaleFt + 2.0) + eps
    z = z / scaleFt
    z = z - shiftFt
    if ld is not None:
        ld[0] = ld[0] - _sum123(torch.log(scaleFt))
    return z


def _pixels(t):
    return int(t.shape[2] * t.shape[3])          # thops.pixels, thops.py:67-68


def flow_step(z, ft, sd, p, coupled, reverse, ld=None):
    """FlowStep.normal_flow :88-111 / reverse_flow :113-129.  logdet terms: actnorm sum(logs)*pixels
    (FlowActNorms.py:85-91), invconv slogdet(W)[1]*pixels (Permutations.py:37,51-57)."""
    if ld is not None:
        d_an = sd[p + ".actnorm.logs"].sum() * _pixels(z)
        d_w = torch.slogdet(sd[p + ".invconv.weight"])[1] * _pixels(z)
    if not reverse:
        z = actnorm(z, sd[p + ".actnorm.bias"], sd[p + ".actnorm.logs"], False)
        if ld is not None:
            ld[0] = ld[0] + d_an
        z = invconv(z, sd[p + ".invconv.weight"], False)
        if ld is not None:
            ld[0] = ld[0] + d_w
        if coupled:
            z = coupling(z, ft, sd, p + ".affine", False, ld=ld)
        return z
    if coupled:
        z = coupling(z, ft, sd, p + ".affine", True, ld=ld)
    z = invconv(z, sd[p + ".invconv.weight"], True)
    if ld is not None:
        ld[0] = ld[0] - d_w
    z = actnorm(z, sd[p + ".actnorm.bias"], sd[p + ".actnorm.logs"], True)
    if ld is not None:
        ld[0] = ld[0] + d_an * -1
    return z


def squeeze2d(x, factor=2):
    """flow.py:122-135."""
    B, C, H, W = x.shape
    x = x.view(B, C, H // factor, factor, W // factor, factor)
    x = x.permute(0, 1, 3, 5, 2, 4).contiguous()
    return x.view(B, C * factor * factor, H // factor, W // factor)


def unsqueeze2d(x, factor=2):
    """flow.py:138-152."""
    B, C, H, W = x.shape
    f2 = factor * factor
    x = x.view(B, C // f2, factor, factor, H, W)
    x = x.permute(0, 1, 4, 2, 5, 3).contiguous()
    return x.view(B, C // f2, H * factor, W * factor)


LOG2PI = float(np.log(2 * np.pi))


def gaussian_logp(mean, logs, x):
    """flow.py:86-107 GaussianDiag.logp."""
    if mean is None and logs is None:
        return _sum123(-0.5 * (x ** 2 + LOG2PI))
    return _sum123(-0.5 * (logs * 2. + ((x - mean) ** 2) / torch.exp(logs * 2.) + LOG2PI))


def split2d(z, sd, p, C_pass, reverse, eps=None, ld=None):
    """Split.py:48-77 (position=None => no ft; logs_eps=0).  logdet += / -= GaussianDiag.logp(mean, logs, z2) (:56,:74)."""
    if not reverse:
        z1, z2 = z[:, :C_pass], z[:, C_pass:]
        mean, logs = cross_split(conv2d_zeros(z1, sd, p + ".conv"))
        e = (z2 - mean) / torch.exp(logs)
        if ld is not None:
            ld[0] = ld[0] + gaussian_logp(mean, logs, z2)
        return z1, e
    z1 = z
    mean, logs = cross_split(conv2d_zeros(z1, sd, p + ".conv"))
    z2 = mean + torch.exp(logs) * eps
    if ld is not None:
        ld[0] = ld[0] - gaussian_logp(mean, logs, z2)
    return torch.cat((z1, z2), 1)


# --------------------------------------------------------------------------------------------
# RRDB encoder
# --------------------------------------------------------------------------------------------
def lrelu(x):
    return F.leaky_relu(x, 0.2)


def rdb(x, sd, p):
    """RRDBNet_arch.py:39-45."""
    c = lambda i, t: F.conv2d(t, sd[f"{p}.conv{i}.weight"], sd[f"{p}.conv{i}.bias"], 1, 1)
    x1 = lrelu(c(1, x))
    x2 = lrelu(c(2, torch.cat((x, x1), 1)))
    x3 = lrelu(c(3, torch.cat((x, x1, x2), 1)))
    x4 = lrelu(c(4, torch.cat((x, x1, x2, x3), 1)))
    x5 = c(5, torch.cat((x, x1, x2, x3, x4), 1))
    return x5 * 0.2 + x


def rrdb(x, sd, p):
    """RRDBNet_arch.py:58-64."""
    out = rdb(x, sd, p + ".RDB1")
    out = rdb(out, sd, p + ".RDB2")
    out = rdb(out, sd, p + ".RDB3")
    return out * 0.2 + x


def rrdbnet(x, sd, opt, nb, p="RRDB"):
    """RRDBNet.forward(get_steps=True), RRDBNet_arch.py:89-148.  Dead heads (`out`, HRconv,
    conv_last) are computed like the reference does."""
    scale = opt["scale"]
    conv = lambda n, t: F.conv2d(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1, 1)
    fea = conv("conv_first", x)
    block_idxs = opt_get(opt, ["network_G", "flow", "stackRRDB", "blocks"]) or []
    res = {}
    for idx in range(nb):
        fea = rrdb(fea, sd, f"{p}.RRDB_trunk.{idx}")
        if idx in block_idxs:
            res[f"block_{idx}"] = fea
    trunk = conv("trunk_conv", fea)
    last_lr_fea = fea + trunk
    # NB: the reference's `self.lrelu` is LeakyReLU(inplace=True) (RRDBNet_arch.py:87), so
    # `fea = self.lrelu(fea_up2)` (:106) overwrites fea_up2 itself: the tensors stored in the
    # results dict are POST-activation (pinned by tests/golden/srflow_rrdb.npz).
    fea_up2 = lrelu(conv("upconv1", F.interpolate(last_lr_fea, scale_factor=2, mode="nearest")))
    fea = fea_up2
    fea_up4 = lrelu(conv("upconv2", F.interpolate(fea, scale_factor=2, mode="nearest")))
    fea = fea_up4
    fea_up8 = None
    if scale >= 8:
        fea_up8 = lrelu(conv("upconv3", F.interpolate(fea, scale_factor=2, mode="nearest")))
        fea = fea_up8
    out = conv("conv_last", lrelu(conv("HRconv", fea)))
    res.update(last_lr_fea=last_lr_fea, fea_up1=last_lr_fea, fea_up2=fea_up2, fea_up4=fea_up4,
               fea_up8=fea_up8, out=out)
    if opt_get(opt, ["network_G", "flow", "fea_up0"]):
        res["fea_up0"] = F.interpolate(last_lr_fea, scale_factor=1 / 2, mode="bilinear",
                                       align_corners=False, recompute_scale_factor=True)
    return res


def rrdb_preprocessing(lr, sd, opt, nb):
    """SRFlowNet.rrdbPreprocessing, SRFlowNet_arch.py:118-138."""
    res = rrdbnet(lr, sd, opt, nb)
    block_idxs = opt_get(opt, ["network_G", "flow", "stackRRDB", "blocks"]) or []
    if len(block_idxs) > 0:
        concat = torch.cat([res[f"block_{i}"] for i in block_idxs], 1)
        if opt_get(opt, ["network_G", "flow", "stackRRDB", "concat"]):
            keys = ["last_lr_fea", "fea_up1", "fea_up2", "fea_up4"]
            if "fea_up0" in res:
                keys.append("fea_up0")
            if opt["scale"] >= 8:
                keys.append("fea_up8")
            for k in keys:
                h, w = res[k].shape[2:]
                res[k] = torch.cat([res[k], F.interpolate(concat, (h, w))], 1)
    return res


# --------------------------------------------------------------------------------------------
# flow encode / decode
# --------------------------------------------------------------------------------------------
def _level(size):
    return int(np.log(160 / size) / np.log(2))


def flow_encode(gt, lr_enc, sd, opt, p="flowUpsamplerNet", ld=None):
    """FlowUpsamplerNet.encode :217-251 with epses=[] => returns [eps_split..., z_final]."""
    names = level_to_name(opt["scale"])
    z = gt
    epses = []
    for i, ly in enumerate(build_layers(opt)):
        ft = lr_enc[names[_level(ly["size"])]]
        if ly["type"] == "squeeze":
            z = squeeze2d(z)
        elif ly["type"] == "step":
            z = flow_step(z, ft, sd, f"{p}.layers.{i}", ly["coupled"], False, ld=ld)
        else:
            z, e = split2d(z, sd, f"{p}.layers.{i}", ly["C_pass"], False, ld=ld)
            epses.append(e)
    epses.append(z)
    return epses


def flow_decode(epses, lr_enc, sd, opt, p="flowUpsamplerNet", ld=None):
    """FlowUpsamplerNet.decode :267-296 (epses copied then popped from the end)."""
    names = level_to_name(opt["scale"])
    epses = list(epses)
    z = epses.pop()
    layers = build_layers(opt)
    for i in reversed(range(len(layers))):
        ly = layers[i]
        ft = lr_enc[names[_level(ly["size"])]]
        if ly["type"] == "squeeze":
            z = unsqueeze2d(z)
        elif ly["type"] == "step":
            z = flow_step(z, ft, sd, f"{p}.layers.{i}", ly["coupled"], True, ld=ld)
        else:
            z = split2d(z, sd, f"{p}.layers.{i}", ly["C_pass"], True, eps=epses.pop(), ld=ld)
    return z


def srflow_encode(gt, lr, sd, opt, nb):
    """SRFlowNet.normal_flow (add_gt_noise=False), SRFlowNet_arch.py:83-116; logdet/nll discarded
    by the caller (test.py:139)."""
    return flow_encode(gt, rrdb_preprocessing(lr, sd, opt, nb), sd, opt)


def srflow_normal_flow(gt, lr, sd, opt, nb):
    """SRFlowNet.normal_flow with add_gt_noise=False and epses=[] (SRFlowNet_arch.py:83-116): -> (epses, nll, logdet)."""
    ld = [torch.zeros_like(gt[:, 0, 0, 0])]
    pixels = _pixels(gt)
    epses = flow_encode(gt, rrdb_preprocessing(lr, sd, opt, nb), sd, opt, ld=ld)
    objective = ld[0].clone() + gaussian_logp(None, None, epses[-1])
    nll = (-objective) / float(np.log(2.) * pixels)
    return epses, nll, ld[0]


def srflow_reverse_flow(lr, epses, sd, opt, nb):
    """SRFlowNet.reverse_flow with add_gt_noise=False (SRFlowNet_arch.py:145-158): -> (sr, logdet)."""
    ld = [torch.zeros_like(lr[:, 0, 0, 0])]
    x = flow_decode(epses, rrdb_preprocessing(lr, sd, opt, nb), sd, opt, ld=ld)
    return x, ld[0]


def srflow_decode(lr, epses, sd, opt, nb):
    """SRFlowNet.reverse_flow, SRFlowNet_arch.py:145-158 (RRDB recomputed, :152-153)."""
    return flow_decode(epses, rrdb_preprocessing(lr, sd, opt, nb), sd, opt)


# --------------------------------------------------------------------------------------------
# prior UNet (models/unet.py)
# --------------------------------------------------------------------------------------------
def dense_block(x, sd, p):
    """unet.py:10-36 DenseBlock_5C (no residual)."""
    c = lambda i, t: F.conv2d(t, sd[f"{p}.conv{i}.weight"], sd[f"{p}.conv{i}.bias"], 1, 1)
    x1 = lrelu(c(1, x))
    x2 = lrelu(c(2, torch.cat((x, x1), 1)))
    x3 = lrelu(c(3, torch.cat((x, x1, x2), 1)))
    x4 = lrelu(c(4, torch.cat((x, x1, x2, x3), 1)))
    return c(5, torch.cat((x, x1, x2, x3, x4), 1))


def bn_eval(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, eps)


def double_conv(x, sd, p):
    """unet.py:38-55: [conv3x3(no bias) -> BN(eval) -> LeakyReLU 0.2] x2."""
    x = lrelu(bn_eval(F.conv2d(x, sd[p + ".double_conv.0.weight"], None, 1, 1), sd, p + ".double_conv.1"))
    x = lrelu(bn_eval(F.conv2d(x, sd[p + ".double_conv.3.weight"], None, 1, 1), sd, p + ".double_conv.4"))
    return x


def unet_up(x1, x2, sd, p):
    """unet.py:72-98 Up (bilinear): upsample x2 align_corners=True, pad to skip, cat([x2,x1])."""
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([x2, x1], 1), sd, p + ".conv")


def _unet_branch(z, sd, tag, depth):
    z = dense_block(z, sd, f"input_proj{tag}")
    feats = []
    z = double_conv(z, sd, f"inc{tag}")
    feats.append(z)
    for i in range(depth):
        z = double_conv(F.max_pool2d(z, 2), sd, f"down_layers{tag}.{i}.maxpool_conv.1")
        feats.append(z)
    for i in range(depth):
        z = unet_up(z, feats[depth - 1 - i], sd, f"up_layers{tag}.{i}")
    return F.conv2d(z, sd[f"outc{tag}.conv.weight"], sd[f"outc{tag}.conv.bias"])


def srflow_prior(epses, sd, depth=3):
    """SRFlow prior UNet.forward(epses), models/unet.py:154-181: two independent branches."""
    return [_unet_branch(epses[0], sd, 0, depth), _unet_branch(epses[1], sd, 1, depth)]


# --------------------------------------------------------------------------------------------
# test.py LP block
# --------------------------------------------------------------------------------------------
def standardize_eps(e):
    """test.py:141-145: per-pixel mean / unbiased std over channels."""
    mean = torch.mean(e, dim=[1], keepdim=True)
    std = torch.std(e, dim=[1], keepdim=True)
    return (e - mean) / (std + 1e-8)


def lp_pipeline(lr, sd_g, sd_prior, opt, nb, prior_depth=3, return_all=False):
    """test.py:126-151 from the already-padded LR tensor: lr_up -> encode -> standardise ->
    prior -> decode -> clamp."""
    with torch.no_grad():
        lr_up = F.interpolate(lr, scale_factor=opt["scale"], mode="bilinear", align_corners=False)
        epses = srflow_encode(lr_up, lr, sd_g, opt, nb)
        epses_n = [standardize_eps(e) for e in epses]
        epses_l = srflow_prior(epses_n, sd_prior, prior_depth)
        sr = srflow_decode(lr, epses_l, sd_g, opt, nb)
        sr_c = torch.clamp(sr, 0, 1)
    if return_all:
        return dict(lr_up=lr_up, epses=epses, epses_norm=epses_n, epses_learned=epses_l, sr_raw=sr,
                    sr=sr_c)
    return sr_c
