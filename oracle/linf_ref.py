"""ORACLE (test infrastructure, not product): CPU restatement of the LINF-LP hot path.

torch-CPU fp32 ops in the reference's op order and with its redundancies (encoder run twice, coef/freq and
the whole per-point conditioning recomputed per 256-row chunk and per direction, `linalg.solve` per call).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Parity status: PINNED by golden vectors emitted from the genuine reference (`tests/golden/make_golden_linf.py`
-> `tests/golden/linf_*.npz`, max-abs recorded in MANIFEST.json); the reference has no tests of its own.

Citations are `path:line` under /root/reference/LINF-LP/.  State-dict keys are the reference's.
"""
import numpy as np
import torch
import torch.nn.functional as F


def make_coord(shape, flatten=False):
    """utils.py:105-120: pixel-centre coordinates in [-1,1]."""
    seqs = []
    for n in shape:
        r = 1.0 / n
        seqs.append(-1 + r + (2 * r) * torch.arange(n).float())
    ret = torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)
    return ret.view(-1, ret.shape[-1]) if flatten else ret


# --------------------------------------------------------------------------------------------
# input preparation (datasets/wrappers.py)
# --------------------------------------------------------------------------------------------
def input_prep(lr, hr_hw, patch_size=3, always_pad=True):
    """coord / cell / gt_lr_up for one LR image `lr` [3,h,w] in [0,1] and an HR size (H,W).
    always_pad=True : SRImplicitPairedFastPatch (wrappers.py:203-238): pad = ps - H % ps even when divisible.
    always_pad=False: SRImplicitDownsampledFastPatchTest (wrappers.py:572-613)."""
    H, W = hr_hw
    ps = patch_size
    hr_coord = make_coord([H, W])
    x = ((lr - 0.5) / 0.5).unsqueeze(0)
    lr_up = F.interpolate(x, (H, W), mode="bilinear", align_corners=False)
    lr_up_down = F.interpolate(lr_up, lr.shape[1:], mode="bilinear", align_corners=False)
    res = (lr_up - F.interpolate(lr_up_down, (H, W), mode="bilinear", align_corners=False)).squeeze(0)
    if always_pad:
        pad_h, pad_w = ps - H % ps, ps - W % ps
    else:
        pad_h = ps - H % ps if H % ps else 0
        pad_w = ps - W % ps if W % ps else 0
    coord_pad = F.pad(hr_coord.permute(2, 0, 1), (0, pad_w, 0, pad_h), "constant", 0)
    cu = coord_pad.unfold(1, ps, ps).unfold(2, ps, ps)
    coord = cu[:, :, :, ps // 2, ps // 2].permute(1, 2, 0)
    p = F.pad(res, (0, pad_w, 0, pad_h), "constant", 0).unfold(1, ps, ps).unfold(2, ps, ps)
    c, qh, qw, _, _ = p.shape
    gt_lr_up = p.contiguous().view(c, qh, qw, ps * ps).permute(0, 3, 1, 2).contiguous().view(c * ps * ps, qh, qw)
    cell = torch.tensor([2 / H, 2 / W], dtype=torch.float32)
    return dict(inp=lr, coord=coord.contiguous(), cell=cell, gt_lr_up=gt_lr_up)


def input_prep_pixelwise(lr, hr_hw):
    """SRImplicitPairedFast (wrappers.py:92-152): coord = full HR grid, gt_lr_up = residual [3,H,W]."""
    H, W = hr_hw
    x = ((lr - 0.5) / 0.5).unsqueeze(0)
    lr_up = F.interpolate(x, (H, W), mode="bilinear", align_corners=False)
    lr_up_down = F.interpolate(lr_up, lr.shape[1:], mode="bilinear", align_corners=False)
    res = (lr_up - F.interpolate(lr_up_down, (H, W), mode="bilinear", align_corners=False)).squeeze(0)
    return dict(inp=lr, coord=make_coord([H, W]), cell=torch.tensor([2 / H, 2 / W], dtype=torch.float32), gt_lr_up=res)


def batch_prep(lr_batch, hr_hw, patch_size=3, always_pad=True):
    if patch_size == 1:
        items = [input_prep_pixelwise(lr_batch[i], hr_hw) for i in range(lr_batch.shape[0])]
        return {k: torch.stack([it[k] for it in items]) for k in items[0]}
    items = [input_prep(lr_batch[i], hr_hw, patch_size, always_pad) for i in range(lr_batch.shape[0])]
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


# --------------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------------
def lrelu(x):
    return F.leaky_relu(x, 0.2)


def _rdb(x, sd, p):
    """models/rrdb.py:52-58."""
    c = lambda i, t: F.conv2d(t, sd[f"{p}.conv{i}.weight"], sd[f"{p}.conv{i}.bias"], 1, 1)
    x1 = lrelu(c(1, x))
    x2 = lrelu(c(2, torch.cat((x, x1), 1)))
    x3 = lrelu(c(3, torch.cat((x, x1, x2), 1)))
    x4 = lrelu(c(4, torch.cat((x, x1, x2, x3), 1)))
    x5 = c(5, torch.cat((x, x1, x2, x3, x4), 1))
    return x5 * 0.2 + x


def rrdb_encoder(x, sd, p, nb):
    """RRDBNet.forward with no_upsampling=True, models/rrdb.py:105-111."""
    conv = lambda n, t: F.conv2d(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1, 1)
    fea = conv("conv_first", x)
    t = fea
    for b in range(nb):
        q = f"{p}.RRDB_trunk.{b}"
        o = _rdb(_rdb(_rdb(t, sd, q + ".RDB1"), sd, q + ".RDB2"), sd, q + ".RDB3")
        t = o * 0.2 + t
    return fea + conv("trunk_conv", t)


def edsr_encoder(x, sd, p, n_resblocks, res_scale=1):
    """EDSR.forward with no_upsampling=True, models/edsr.py:134-146 (+ ResBlock :30-51)."""
    conv = lambda n, t: F.conv2d(t, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1, 1)
    x = conv("head.0", x)
    res = x
    for i in range(n_resblocks):
        r = conv(f"body.{i}.body.2", F.relu(conv(f"body.{i}.body.0", res)))
        res = r * res_scale + res
    res = conv(f"body.{n_resblocks}", res)
    return res + x


def encoder(x, sd, spec):
    name = spec["name"]
    args = spec.get("args", {})
    if name == "rrdb":
        return rrdb_encoder(x, sd, "encoder", args.get("nb", 23))
    if name == "edsr-baseline":
        return edsr_encoder(x, sd, "encoder", args.get("n_resblocks", 16), args.get("res_scale", 1))
    raise NotImplementedError(name)


# --------------------------------------------------------------------------------------------
# local implicit conditioning + flow (models/linf.py, models/flow.py)
# --------------------------------------------------------------------------------------------
def affine_info(feat, coord, cell, sd):
    """LINFPatch.query_* up to `affine_info = self.layers(features)`, linf.py:325-391."""
    coef = F.conv2d(feat, sd["coef.weight"], sd["coef.bias"], 1, 1)
    freq = F.conv2d(feat, sd["freq.weight"], sd["freq.bias"], 1, 1)
    h, w = feat.shape[-2:]
    rx, ry = 2 / h / 2, 2 / w / 2
    eps_shift = 1e-6
    feat_coord = make_coord((h, w)).permute(2, 0, 1).unsqueeze(0).expand(feat.shape[0], 2, h, w)
    freqs, coefs, areas = [], [], []
    for vx in (-1, 1):
        for vy in (-1, 1):
            coord_ = coord.clone()
            coord_[:, :, :, 0] += vx * rx + eps_shift
            coord_[:, :, :, 1] += vy * ry + eps_shift
            coord_.clamp_(-1 + 1e-6, 1 - 1e-6)
            q_coord = F.grid_sample(feat_coord, coord_.flip(-1), mode="nearest", align_corners=False)
            rel_coord = coord.permute(0, 3, 1, 2) - q_coord
            rel_coord[:, 0, :, :] *= h
            rel_coord[:, 1, :, :] *= w
            rel_cell = cell.clone()
            rel_cell[:, 0] *= h
            rel_cell[:, 1] *= w
            coef_ = F.grid_sample(coef, coord_.flip(-1), mode="nearest", align_corners=False)
            freq_ = F.grid_sample(freq, coord_.flip(-1), mode="nearest", align_corners=False)
            freq_ = torch.stack(torch.split(freq_, freq.shape[1] // 2, dim=1), dim=2)
            freq_ = torch.mul(freq_, rel_coord.unsqueeze(1))
            freq_ = torch.sum(freq_, dim=2)
            freq_ = freq_ + F.linear(rel_cell, sd["phase.weight"]).unsqueeze(-1).unsqueeze(-1)
            freq_ = torch.cat((torch.cos(np.pi * freq_), torch.sin(np.pi * freq_)), dim=1)
            freqs.append(freq_)
            coefs.append(coef_)
            areas.append(torch.abs(rel_coord[:, 0, :, :] * rel_coord[:, 1, :, :]) + 1e-9)
    tot_area = torch.stack(areas).sum(dim=0)
    areas[0], areas[3] = areas[3], areas[0]
    areas[1], areas[2] = areas[2], areas[1]
    for i in range(4):
        wgt = (areas[i] / tot_area).unsqueeze(1)
        coefs[i] = torch.mul(wgt * coefs[i], freqs[i])
    x = torch.cat(coefs, dim=1)
    n_mlp = len([k for k in sd if k.startswith("layers.") and k.endswith(".weight")])
    for j in range(n_mlp):
        x = F.conv2d(x, sd[f"layers.{2 * j}.weight"], sd[f"layers.{2 * j}.bias"])
        if j < n_mlp - 1:
            x = F.relu(x)
    return x


def _flow_affine(ai, i, D):
    s = ai[:, 2 * D * i: 2 * D * i + D]
    shift = ai[:, 2 * D * i + D: 2 * D * (i + 1)]
    return torch.sigmoid(s + 2.0) + 1e-4, shift


def flow_forward(x, ai, sd, p="imnet", n_layers=10, with_logp=False):
    """Flow.forward, models/flow.py:44-55.  with_logp: also return total_log_det_J + base log-prob per point
    (NaiveLinear logabsdet = slogdet(W)[1] :66-70,:106; affine sum(log scale) :28-36; -0.5*(z^2 + log 2pi) :54)."""
    D = x.shape[1]
    z = x
    tot = x.new_zeros(x.shape[0])
    for i in range(n_layers):
        z = F.linear(z, sd[f"{p}.linears.{i}._weight"], sd[f"{p}.linears.{i}.bias"])
        tot += torch.slogdet(sd[f"{p}.linears.{i}._weight"])[1] * z.new_ones(z.shape[0])
        scale, shift = _flow_affine(ai, i, D)
        z = z * scale + shift
        tot += torch.sum(torch.log(scale), dim=-1)
    z = F.linear(z, sd[f"{p}.last._weight"], sd[f"{p}.last.bias"])
    tot += torch.slogdet(sd[f"{p}.last._weight"])[1] * z.new_ones(z.shape[0])
    tot += torch.sum(-0.5 * (z ** 2 + float(np.log(2 * np.pi))), -1)
    return (z, tot) if with_logp else z


def flow_inverse(z, ai, sd, p="imnet", n_layers=10):
    """Flow.inverse, models/flow.py:57-63; NaiveLinear.inverse :108-122 (linalg.solve per call)."""
    D = z.shape[1]

    def lin_inv(v, q):
        v = v - sd[q + ".bias"]
        return torch.linalg.solve(sd[q + "._weight"], v.t()).t()

    x = lin_inv(z, f"{p}.last")
    for i in reversed(range(n_layers)):
        scale, shift = _flow_affine(ai, i, D)
        x = (x - shift) / scale
        x = lin_inv(x, f"{p}.linears.{i}")
    return x


def query_log_p(feat, coord, cell, gt, sd, n_layers=10, with_logp=False):
    """LINFPatch.query_log_p, linf.py:248-322 -> z [B,D,qh,qw]  (with_logp: (log_p [B*qh*qw], z) like the reference)."""
    ai = affine_info(feat, coord, cell, sd)
    bs, qh, qw, _ = coord.shape
    x = gt.permute(0, 2, 3, 1).contiguous().view(bs * qh * qw, -1)
    a = ai.permute(0, 2, 3, 1).contiguous().view(bs * qh * qw, -1)
    if with_logp:
        z, lp = flow_forward(x, a, sd, n_layers=n_layers, with_logp=True)
        return lp, z.reshape(bs, qh, qw, -1).permute(0, 3, 1, 2)
    z = flow_forward(x, a, sd, n_layers=n_layers)
    return z.reshape(bs, qh, qw, -1).permute(0, 3, 1, 2)


def query_rgb_pixelwise(inp, feat, coord, cell, zmap, sd, n_layers=10):
    """LINF.query_rgb (ps=1), linf.py:122-195: D=3 inverse flow per pixel + bilinear grid_sample skip of `inp`."""
    ai = affine_info(feat, coord, cell, sd)
    bs, qh, qw, _ = coord.shape
    z = zmap.permute(0, 2, 3, 1).contiguous().view(-1, 3)
    a = ai.permute(0, 2, 3, 1).contiguous().view(bs * qh * qw, -1)
    pred = flow_inverse(z, a, sd, n_layers=n_layers)
    pred = pred.clone().view(bs, qh, qw, -1).permute(0, 3, 1, 2).contiguous()
    return pred + F.grid_sample(inp, coord.flip(-1), mode="bilinear", padding_mode="border", align_corners=False)


def query_rgb(feat, coord, cell, zmap, sd, patch_size=3, n_layers=10):
    """LINFPatch.query_rgb, linf.py:324-407 -> folded pred [B,3,ps*qh,ps*qw]."""
    ai = affine_info(feat, coord, cell, sd)
    bs, qh, qw, _ = coord.shape
    D = 3 * patch_size * patch_size
    z = zmap.permute(0, 2, 3, 1).contiguous().view(-1, D)
    a = ai.permute(0, 2, 3, 1).contiguous().view(bs * qh * qw, -1)
    pred = flow_inverse(z, a, sd, n_layers=n_layers)
    pred = pred.clone().view(bs, qh, qw, -1).permute(0, 3, 1, 2).contiguous()
    return F.fold(pred.view(bs, D, -1), output_size=(qh * patch_size, qw * patch_size),
                  kernel_size=(patch_size, patch_size), stride=patch_size)


# --------------------------------------------------------------------------------------------
# prior UNet (models/unet.py)
# --------------------------------------------------------------------------------------------
def _dense(x, sd, p):
    c = lambda i, t: F.conv2d(t, sd[f"{p}.conv{i}.weight"], sd[f"{p}.conv{i}.bias"], 1, 1)
    x1 = lrelu(c(1, x))
    x2 = lrelu(c(2, torch.cat((x, x1), 1)))
    x3 = lrelu(c(3, torch.cat((x, x1, x2), 1)))
    x4 = lrelu(c(4, torch.cat((x, x1, x2, x3), 1)))
    return c(5, torch.cat((x, x1, x2, x3, x4), 1))


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def _double_conv(x, sd, p):
    x = lrelu(_bn(F.conv2d(x, sd[p + ".double_conv.0.weight"], None, 1, 1), sd, p + ".double_conv.1"))
    return lrelu(_bn(F.conv2d(x, sd[p + ".double_conv.3.weight"], None, 1, 1), sd, p + ".double_conv.4"))


def _up(x1, x2, sd, p):
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double_conv(torch.cat([x2, x1], 1), sd, p + ".conv")


def linf_prior(x, lr, sd, depth=3):
    """UNet.forward(x, lr), models/unet.py:144-167."""
    x = _dense(x, sd, "input_proj")
    e = lrelu(F.conv2d(lr, sd["lr_proj.0.weight"], sd["lr_proj.0.bias"], 3, 1))
    e = _dense(e, sd, "lr_proj.2")
    if e.shape != x.shape:
        e = F.interpolate(e, size=x.shape[2:], mode="bilinear", align_corners=False)
    x = torch.cat([x, e], 1)
    feats = []
    x = _double_conv(x, sd, "inc")
    feats.append(x)
    for i in range(depth):
        x = _double_conv(F.max_pool2d(x, 2), sd, f"down_layers.{i}.maxpool_conv.1")
        feats.append(x)
    for i in range(depth):
        x = _up(x, feats[depth - 1 - i], sd, f"up_layers.{i}")
    return F.conv2d(x, sd["outc.conv.weight"], sd["outc.conv.bias"])


# --------------------------------------------------------------------------------------------
# test.py eval_psnr LP branch
# --------------------------------------------------------------------------------------------
def lp_pipeline(batch, sd, sd_prior, model_spec, hr_hw, patch_size=3, return_all=False, chunk=256):
    """LINF-LP/test.py:94-171,217 (eval_bsize set, patch=True, prior given): normalise, encode in 256-row
    chunks, prior (+ bilinear resize if shapes differ), decode in chunks, crop, += bilinear(inp), clamp."""
    with torch.no_grad():
        inp = (batch["inp"] - 0.5) / 0.5
        coord, cell, gt = batch["coord"], batch["cell"], batch["gt_lr_up"]
        enc_spec = model_spec["args"]["encoder_spec"]
        n_layers = model_spec["args"].get("flow_layers", 10)
        feat = encoder(inp, sd, enc_spec)                                     # gen_feat (test.py:38)
        zs = []
        for r in range(0, coord.shape[1], chunk):
            zs.append(query_log_p(feat, coord[:, r:r + chunk], cell, gt[:, :, r:r + chunk], sd, n_layers))
        z_lr = torch.cat(zs, dim=2).contiguous()
        z_learned = linf_prior(z_lr, inp, sd_prior)
        if z_learned.shape != z_lr.shape:
            z_learned = F.interpolate(z_learned, size=z_lr.shape[-2:], mode="bilinear", align_corners=False)
        feat = encoder(inp, sd, enc_spec)                                     # gen_feat again (test.py:22)
        ps = []
        for r in range(0, coord.shape[1], chunk):
            if patch_size == 1:
                ps.append(query_rgb_pixelwise(inp, feat, coord[:, r:r + chunk], cell, z_learned[:, :, r:r + chunk], sd, n_layers))
            else:
                ps.append(query_rgb(feat, coord[:, r:r + chunk], cell, z_learned[:, :, r:r + chunk], sd, patch_size, n_layers))
        pred = torch.cat(ps, dim=2)
        H, W = hr_hw
        pred = pred[..., :H, :W]
        if patch_size != 1:                                                   # `if patch:` (test.py:169-171)
            pred = pred + F.interpolate(inp, pred.shape[-2:], mode="bilinear", align_corners=False)
        out = torch.clamp(pred * 0.5 + 0.5, 0, 1)
    if return_all:
        return dict(z_lr=z_lr, z_learned=z_learned, pred_raw=pred, pred=out)
    return out
