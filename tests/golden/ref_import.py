"""Import harness for the genuine reference (liyuantsao/BFSR) -- THIS CONTAINER ONLY.

Used exclusively by `tests/golden/make_golden.py` to emit golden vectors. Nothing here is
imported by the product, by `-m gpu` tests, by `smoke()` or by `bench.py`: `/root/reference`
does not exist on the GPU box.

What it does (SURVEY.md section 8c "Obstacles and workarounds"):
  * stubs the peripheral modules the reference imports but the hot path never calls
    (cv2, natsort, torchvision, lpips, tensorboardX, timm, imageio, skimage);
  * makes `Tensor.cuda()` / `Module.cuda()` no-ops (the reference hard-codes `.cuda()`);
  * puts one of the two reference sub-projects on `sys.path` (they share module names
    such as `models`, `utils`, so only one can be live per process).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("BFSR_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "SRFlow-LP", "code"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch

    # cv2 is not installed.  The reference calls exactly two OpenCV functions on the path to its SSIM (LINF-LP/utils.py:158-166):
    #   cv2.getGaussianKernel(ksize, sigma) -> [ksize,1] float64 column, exp(-(i-(ksize-1)/2)^2 / (2 sigma^2)) normalised to sum 1
    #   cv2.filter2D(img, -1, kernel)       -> same-size CORRELATION with the kernel anchored at its centre, BORDER_REFLECT_101
    # (OpenCV documentation); both are restated here so that the reference's own calculate_ssim runs unmodified.
    import numpy as _np

    def _get_gaussian_kernel(ksize, sigma):
        i = _np.arange(ksize, dtype=_np.float64) - (ksize - 1) / 2.0
        g = _np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (g / g.sum()).reshape(-1, 1)

    def _filter2d(img, ddepth, kernel):
        from scipy.ndimage import correlate
        assert ddepth == -1
        return correlate(_np.asarray(img, dtype=_np.float64), _np.asarray(kernel, dtype=_np.float64), mode="mirror")

    _stub("cv2", getGaussianKernel=_get_gaussian_kernel, filter2D=_filter2d)
    ns = _stub("natsort", natsorted=sorted)
    ns.natsort = ns
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.transforms = _stub("torchvision.transforms")
    tv.transforms.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic", BILINEAR="bilinear")
    tv.models = _stub("torchvision.models")
    # LPIPS needs pretrained AlexNet weights (no network): a stand-in that returns zeros keeps `eval_psnr(detail=True)` runnable;
    # its 'lpips' entry is NOT a golden value and is never stored
    class _LPIPS(object):
        def __init__(self, net="alex"):
            pass

        def to(self, *a, **k):
            return self

        cuda = to

        def __call__(self, a, b):
            return torch.zeros(a.shape[0], 1, 1, 1)

        forward = __call__

    _stub("lpips", LPIPS=_LPIPS)
    _stub("tensorboardX", SummaryWriter=object)
    _stub("imageio")
    sk = _stub("skimage")

    def _psnr(a, b, data_range=None):
        """skimage.metrics.peak_signal_noise_ratio: 10 log10(data_range^2 / mse); data_range from the dtype (255 for uint8)"""
        if data_range is None:
            data_range = 255.0 if a.dtype == _np.uint8 else 1.0
        err = _np.mean((a.astype(_np.float64) - b.astype(_np.float64)) ** 2)
        return 10.0 * _np.log10(data_range ** 2 / err)

    sk.metrics = _stub("skimage.metrics", peak_signal_noise_ratio=_psnr, structural_similarity=None)
    timm = _stub("timm")
    timm.models = _stub("timm.models")
    timm.models.layers = _stub(
        "timm.models.layers",
        DropPath=torch.nn.Identity,
        to_2tuple=lambda x: (x, x),
        trunc_normal_=lambda t, std=0.02: t,
    )
    # the reference hard-codes .cuda(); on this CPU-only container make it a no-op
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def _purge(prefixes):
    for k in list(sys.modules):
        if any(k == p or k.startswith(p + ".") for p in prefixes):
            del sys.modules[k]


_SHARED = ("models", "utils", "options", "datasets", "test", "imresize", "Measure")


def use_srflow():
    """Make `models.modules.*`, `options`, `utils.util` resolve to SRFlow-LP/code."""
    install_stubs()
    _purge(_SHARED)
    root = os.path.join(REF_ROOT, "SRFlow-LP", "code")
    sys.path[:] = [p for p in sys.path if not p.startswith(REF_ROOT)]
    sys.path.insert(0, root)
    return root


def use_linf():
    """Make `models`, `utils`, `datasets` resolve to LINF-LP."""
    install_stubs()
    _purge(_SHARED)
    root = os.path.join(REF_ROOT, "LINF-LP")
    sys.path[:] = [p for p in sys.path if not p.startswith(REF_ROOT)]
    sys.path.insert(0, root)
    return root


def srflow_opt(scale=4, conf="confs/SRFlow-LP_DF2K_4X.yml"):
    """Parse the shipped 4X yml through the reference's own options.parse; optionally derive the
    8x variant the way SURVEY section 8d describes (scale: 8, network_G.upscale: 8, L: 3)."""
    root = use_srflow()
    option = importlib.import_module("options.options")
    cwd = os.getcwd()
    os.chdir(root)
    try:
        opt = option.parse(os.path.join(root, conf), is_train=False)
    finally:
        os.chdir(cwd)
    opt["gpu_ids"] = None
    if scale != 4:
        opt["scale"] = scale
        opt["network_G"]["upscale"] = scale
    opt = option.dict_to_nonedict(opt)
    return opt


def build_srflownet(opt, nb=None):
    """Instantiate the reference SRFlowNet directly (bypasses SRFlowModel which insists on
    loading ../pretrained_models/*.pth)."""
    net_mod = importlib.import_module("models.modules.SRFlowNet_arch")
    g = opt["network_G"]
    net = net_mod.SRFlowNet(in_nc=g["in_nc"], out_nc=g["out_nc"], nf=g["nf"],
                            nb=(nb if nb is not None else g["nb"]), scale=opt["scale"],
                            K=g["flow"]["K"], opt=opt, step=None)
    net.eval()
    return net
