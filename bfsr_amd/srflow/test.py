"""SRFlow-LP evaluation harness -- counterpart of the reference's `test.py`
(SRFlow-LP/code/test.py:85-175).  Usage mirrors it:  python -m bfsr_amd.srflow.test <conf.yml>

`lp_infer` is the LP block (test.py:126-151) on device tensors: lr_up = bilinear x scale -> encode
(`add_gt_noise=False`) -> per-pixel channel standardisation -> learned prior -> decode -> clamp[0,1].
PNG I/O and PSNR/SSIM/LPIPS (Measure.py) are outside the accelerated path."""
import glob
import os
import sys

import numpy as np
import torch

from .models import create_model, models as registry
from .options import load as load_opt, opt_get
from ..ops import MODE_BILINEAR


def load_model(conf_path, ops=None, allow_uninitialised=False):
    """test.py:41-49: parse conf, build the model, load `model_path` into netG.  Like the reference (which calls
    `load_network` unconditionally) a missing checkpoint is an error, not a silent run on default-initialised weights;
    `allow_uninitialised=True` is the explicit opt-out for callers that load a state_dict themselves."""
    opt = load_opt(conf_path)
    model = create_model(opt, ops=ops)
    model_path = opt_get(opt, ['model_path'], None)
    if model_path is None:
        if not allow_uninitialised:
            raise FileNotFoundError("%s: `model_path` is not set" % conf_path)
    elif not os.path.exists(model_path):
        raise FileNotFoundError("%s: model_path %r does not exist" % (conf_path, model_path))
    else:
        model.load_network(load_path=model_path, network=model.netG)
    return model, opt


def load_prior(opt, ops=None):
    """test.py:90-91: `models.make(torch.load(prior_model_path)['prior_model'], load_sd=True)`."""
    path = opt_get(opt, ['prior_model_path'], None)
    if not path or not os.path.exists(path):
        raise FileNotFoundError("prior_model_path %r does not exist" % (path,))
    spec = torch.load(path, map_location='cpu')['prior_model']
    args = dict(spec['args'])
    if ops is not None:
        args['ops'] = ops
    prior = registry.make({'name': spec['name'], 'args': args, 'sd': spec['sd']}, load_sd=True)
    prior.eval()
    return prior


def natsorted(paths):
    """Natural (numeric-aware) ordering like `natsort.natsorted` (test.py:37-38): 2.png sorts before 10.png, so the output
    index `{idx:06d}.png` maps to the same input as upstream."""
    import re
    return sorted(paths, key=lambda p: [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', p)])


def pad_lr_to_even(lr):
    """test.py:126-130: reflect-pad H and W of an HWC uint8 LR image up to a multiple of 2."""
    h, w, _ = lr.shape
    return np.pad(lr, [(0, int(np.ceil(h / 2) * 2 - h)), (0, int(np.ceil(w / 2) * 2 - w)), (0, 0)], 'reflect')


def pad_lr_to_even_t(lr_t):
    """The same pad (test.py:126-130) on an uploaded [B,3,h,w] tensor: at most one reflected row / column (numpy's 'reflect' excludes the
    edge: the appended row is row h-2), by slicing on the device -- no host round trip between the image upload and the engine."""
    h, w = lr_t.shape[2], lr_t.shape[3]
    if h % 2:
        lr_t = torch.cat([lr_t, lr_t[:, :, h - 2:h - 1] if h > 1 else lr_t], 2)
    if w % 2:
        lr_t = torch.cat([lr_t, lr_t[:, :, :, w - 2:w - 1] if w > 1 else lr_t], 3)
    return lr_t.contiguous()


def _lp_lane(eng, prior_eng, lr, scale, sr_out, keep=None):
    """The LP block (test.py:126-151) on engine level; writes clamp(sr) into sr_out.

    Overlap (round 5): the prior's two branches are independent (models/unet.py:154-181).  Branch 0 works on eps0 -- 6 channels at half the HR
    resolution, 3/4 of the prior's convolutions -- which exists as soon as level 1's split has run in encode, and its output is needed only at
    decode's level-1 split, at the very end.  It is enqueued on the engine's side stream at that point and runs under the level-2/3 chains of
    encode, branch 1 and the level-3/2 chains of decode: short, HBM- or latency-bound launches that leave the matrix pipes idle.  Same kernels,
    same results; BFSR_OVERLAP=0 keeps everything on one stream."""
    ops = eng.ops
    B, _, h, w = lr.shape
    lr_up = eng.ws.get("lr_up", B, 3, h * scale, w * scale)
    ops.resize(lr, lr_up, MODE_BILINEAR, 1.0 / scale, 1.0 / scale)               # test.py:137
    side = None
    if getattr(getattr(ops, "device", None), "type", "cpu") == "cuda" and os.environ.get("BFSR_OVERLAP", "auto") != "0" and os.environ.get("BFSR_PRIOR_OVERLAP", "1") != "0" and hasattr(prior_eng, "forward_branch"):
        if eng._side_stream is None:
            eng._side_stream = torch.cuda.Stream(device=ops.device)
        side = eng._side_stream
    early = {}

    def on_eps(i, e):
        if side is None or i != 0:
            return
        main = torch.cuda.current_stream(ops.device)
        n0, o0 = ops.empty(*e.shape), ops.empty(*e.shape)       # allocated on the caller's stream (they outlive the side stream's work)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.standardize(e, n0)
            prior_eng.forward_branch(0, n0, out=o0)
            early[0] = (n0, o0, side.record_event())

    epses_lr = eng.encode(lr_up, lr, on_eps=on_eps)                              # test.py:139 (add_gt_noise=False)
    if 0 in early:                                                               # test.py:141-147 with branch 0 already in flight
        n0, o0, ev = early[0]
        epses = [n0] + [ops.standardize(e, ops.empty(*e.shape)) for e in epses_lr[1:]]
        epses_learned = [o0] + [prior_eng.forward_branch(b, epses[b]) for b in range(1, len(epses))]
        sr_raw = eng.decode(lr, epses=epses_learned, eps_ready={0: ev})          # test.py:148
        if keep is not None:
            torch.cuda.current_stream(ops.device).wait_event(ev)
    else:
        epses = [ops.standardize(e, ops.empty(*e.shape)) for e in epses_lr]      # test.py:141-145
        epses_learned = prior_eng.forward(epses)                                 # test.py:147
        sr_raw = eng.decode(lr, epses=epses_learned)                             # test.py:148
    ops.axpb_clamp(sr_raw, sr_out, 1.0, 0.0, 0.0, 1.0)                           # test.py:150
    if keep is not None:
        keep.update(lr_up=lr_up, epses=epses_lr, epses_norm=epses, epses_learned=epses_learned, sr_raw=sr_raw, sr=sr_out)


def lp_infer(model, prior_model, lr_t, return_all=False, check_range=True):
    """lr_t [B,3,h,w] in [0,1] (h, w even) -> sr [B,3,s*h,s*w] clamped to [0,1].
    check_range (default): the pass runs under the range guard of the two-term fp16 split (guard.run_guarded): if a kernel met a value the
    split cannot hold, the whole pass is re-run under the bf16x3 split (fp32's exponent range) and that result is returned -- never inf / NaN
    from an overflow, never a silently wrong value.  Cost: one 4-byte device->host read at the end of the pass (the reference's loop
    synchronises there anyway, test.py:150 `.cpu()`).  False skips the guard (and the synchronisation)."""
    net = model.netG.module
    scale = model.opt['scale']

    def run():
        eng = net.engine()
        ops = eng.ops
        with torch.no_grad():
            lr = ops.to_device(lr_t)
            B, _, h, w = lr.shape
            sr = ops.empty(B, 3, h * scale, w * scale)
            keep = {} if return_all else None
            _lp_lane(eng, prior_model.engine(), lr, scale, sr, keep)
        return keep if return_all else sr
    if not check_range:
        return run()
    from ..guard import run_guarded
    return run_guarded([net, prior_model], run)


def format_measurements(meas):
    """test.py:176-182."""
    return ", ".join("%s: %s" % (k, ("%0.4f" % v) if isinstance(v, float) else v) for k, v in meas.items())


def lr_reconstruct_uint8(ops, sr_u8, scale):
    """test.py:159 `imresize(sr, 1 / scale)` for a uint8 HWC image: MATLAB-style antialiased bicubic, rows then columns, the result of EACH pass
    clipped and rounded to uint8 (imresize.py:113-125, :166-168) -- on the device: resample_taps + to_uint8 per pass (fp32 taps; the
    reference accumulates in float64, so a value within ~1e-5 of a rounding tie may land on the other side)."""
    from ..linf import metrics
    x = ops.to_device(torch.as_tensor(np.asarray(sr_u8)).permute(2, 0, 1).unsqueeze(0).to(torch.float32) / 255.0)
    B, C, H, W = x.shape
    oh, ow = int(np.ceil(H / scale)), int(np.ceil(W / scale))
    tabs = []
    for n_in, n_out in ((H, oh), (W, ow)):
        w, i = metrics.imresize_tables(n_in, n_out, 1.0 / scale)
        tabs.append((torch.from_numpy(np.ascontiguousarray(i)).to(x.device), torch.from_numpy(np.ascontiguousarray(w.astype(np.float32))).to(x.device)))
    mid = ops.to_uint8(ops.resample_taps(x, ops.empty(B, C, oh, W), tabs[0][0], tabs[0][1], 0)).to(torch.float32) / 255.0
    out = ops.to_uint8(ops.resample_taps(mid.contiguous(), ops.empty(B, C, oh, ow), tabs[1][0], tabs[1][1], 1))
    return out[0].permute(1, 2, 0).cpu().numpy()


def main(argv=None):
    """test.py:85-174: every LR image of `dataroot_LR` through the LP pipeline, SR images to ../results/SRFlow-LP/; when `dataroot_GT` holds the
    HR images, PSNR / SSIM / LR-consistency PSNR per image into measure_full.csv like the reference (the LPIPS column stays empty: its
    pretrained network is not available here)."""
    argv = argv if argv is not None else sys.argv[1:]
    if len(argv) != 1:
        raise SystemExit("usage: python -m bfsr_amd.srflow.test <conf.yml>")
    from collections import OrderedDict
    from PIL import Image
    from .Measure import Measure
    model, opt = load_model(argv[0])
    prior = load_prior(opt)
    scale = opt['scale']
    conf = os.path.basename(argv[0]).replace('.yml', '')
    lr_paths = natsorted(glob.glob(os.path.join(opt['dataroot_LR'], '*.png')))     # test.py:37-38 uses natsort
    hr_dir = opt.get('dataroot_GT') if hasattr(opt, 'get') else None
    hr_paths = natsorted(glob.glob(os.path.join(hr_dir, '*.png'))) if hr_dir else []
    out_dir = os.path.join(os.path.dirname(os.path.abspath(argv[0])), '..', 'results', 'SRFlow-LP')
    os.makedirs(out_dir, exist_ok=True)
    ops = model.netG.module.engine().ops
    measure, rows = Measure(ops), []
    for idx, p in enumerate(lr_paths):
        lr = np.asarray(Image.open(p).convert('RGB'))
        h, w, _ = lr.shape
        lr_t = pad_lr_to_even_t(ops.to_device(torch.from_numpy(lr.transpose(2, 0, 1)[None].astype(np.float32)) / 255))
        sr = lp_infer(model, prior, lr_t)
        img = (np.clip(sr[0].cpu().numpy().transpose(1, 2, 0), 0, 1) * 255).astype(np.uint8)[:h * scale, :w * scale]
        Image.fromarray(img).save(os.path.join(out_dir, "{:06d}.png".format(idx)))
        if idx < len(hr_paths):
            hr = np.asarray(Image.open(hr_paths[idx]).convert('RGB'))
            meas = OrderedDict(conf=conf, name=idx)
            meas['PSNR'], meas['SSIM'] = measure.measure(img, hr, with_lpips=False)
            meas['LPIPS'] = float('nan')
            meas['LRC PSNR'] = measure.psnr(lr, lr_reconstruct_uint8(ops, img, scale))
            rows.append(meas)
            print(format_measurements(meas))
        else:
            print("wrote %06d.png" % idx)
    if rows:
        import pandas as pd
        df = pd.DataFrame(rows[::-1])                                 # the reference prepends each row (test.py:166)
        path = os.path.join(out_dir, 'measure_full.csv')
        df.to_csv(path + "_", index=False)
        os.replace(path + "_", path)
        print("Results in: %s" % path)
        print("Mean: " + format_measurements(OrderedDict((k, float(v)) for k, v in df.drop(columns=['conf']).mean().items())))


if __name__ == "__main__":
    main()
