"""Golden vectors for the evaluation metrics of the genuine reference (build container only, see ref_import.py):
LINF-LP/imresize.py `imresize` (MATLAB-style bicubic, scales 1/2, 1/3, 1/4 on odd sizes) and LINF-LP/utils.py `calc_psnr`
(plain / div2k / benchmark).  The oracle (oracle/metrics_ref.py) is checked against them; differences -> MANIFEST.json "metrics".
SSIM is NOT generated: the reference computes it with cv2, which this image does not have (parity unpinned, see the oracle).
Usage: python tests/golden/make_golden_metrics.py"""
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402

R.use_linf()
ref_imresize = importlib.import_module("imresize")
ref_utils = importlib.import_module("utils")
import oracle.metrics_ref as O  # noqa: E402

g = np.random.Generator(np.random.PCG64(31))
out, man = {}, {}
for tag, (H, W), s in (("a", (37, 41), 2), ("b", (48, 60), 3), ("c", (50, 34), 4)):
    img = g.random((H, W, 3), dtype=np.float32)
    ref = ref_imresize.imresize(img, 1 / s)
    mine = O.imresize(img, 1 / s)
    man["imresize_" + tag] = float(np.abs(ref - mine).max())
    out["img_" + tag], out["scale_" + tag], out["lr_" + tag] = img, np.int64(s), ref
sr = torch.from_numpy(g.random((2, 3, 40, 44), dtype=np.float32))
hr = torch.clamp(sr + torch.from_numpy((0.05 * g.standard_normal((2, 3, 40, 44))).astype(np.float32)), 0, 1)
out["psnr_sr"], out["psnr_hr"] = sr.numpy(), hr.numpy()
for name, kw in (("plain", {}), ("div2k4", dict(dataset="div2k", scale=4)), ("bench3", dict(dataset="benchmark", scale=3))):
    v = float(ref_utils.calc_psnr(sr, hr, **kw))
    out["psnr_" + name] = np.float64(v)
    man["psnr_" + name] = abs(v - float(O.calc_psnr(sr.numpy(), hr.numpy(), **kw)))
np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
mp = os.path.join(HERE, "MANIFEST.json")
manifest = json.load(open(mp))
manifest["metrics"] = man
json.dump(manifest, open(mp, "w"), indent=1, sort_keys=True)
print(json.dumps(man, indent=1))
