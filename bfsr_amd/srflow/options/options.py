"""YAML config surface of SRFlow-LP: same keys, same NoneDict / opt_get semantics as the reference
(options/options.py:26-98 `parse`, :114-129 `NoneDict`/`dict_to_nonedict`; utils/util.py:167-175
`opt_get`).  Every module of the engine reads its architecture from `opt` through `opt_get`, so the
reference's yml files drive this engine unchanged."""
import os.path as osp

import yaml


class NoneDict(dict):
    """dict returning None for missing keys (reference options.py:114-116)."""

    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def opt_get(opt, keys, default=None):
    if opt is None:
        return default
    cur = opt
    for k in keys:
        cur = cur.get(k, None) if isinstance(cur, dict) else None
        if cur is None:
            return default
    return cur


def parse(opt_path, is_train=False):
    """Load a SRFlow-LP yml.  Derived fields follow the reference: `is_train`, per-dataset
    `phase`/`scale`/`data_type`, expanded paths, `path.root`, `path.results_root`,
    `network_G.scale`."""
    with open(opt_path, "r") as f:
        opt = yaml.safe_load(f)
    opt["is_train"] = is_train
    scale = opt.get("scale") if opt.get("distortion") == "sr" else None
    for phase, ds in (opt.get("datasets") or {}).items():
        ds["phase"] = phase.split("_")[0]
        if scale is not None:
            ds["scale"] = scale
        lmdb = False
        for key in ("dataroot_GT", "dataroot_LQ"):
            if ds.get(key) is not None:
                ds[key] = osp.expanduser(ds[key])
                lmdb = lmdb or ds[key].endswith("lmdb")
        ds["data_type"] = "lmdb" if lmdb else "img"
    paths = opt.setdefault("path", {})
    for key, p in list(paths.items()):
        if p and key != "strict_load" and isinstance(p, str):
            paths[key] = osp.expanduser(p)
    paths["root"] = osp.abspath(osp.join(osp.dirname(opt_path), osp.pardir))
    if not is_train:
        if not paths.get("results_root"):
            paths["results_root"] = osp.join(paths["root"], "results", str(opt.get("name")))
        paths["log"] = paths["results_root"]
    if scale is not None and "network_G" in opt:
        opt["network_G"]["scale"] = scale
    return opt


def load(opt_path, **overrides):
    """parse + dict_to_nonedict, with the two lines test.py applies (`gpu_ids=None`)."""
    opt = parse(opt_path, is_train=False)
    opt["gpu_ids"] = None
    for k, v in overrides.items():
        opt[k] = v
    return dict_to_nonedict(opt)


def derive_scale(opt, scale):
    """The 8x variant used by BASELINE config 4: the shipped 4X yml with `scale: 8`,
    `network_G.upscale: 8`, `L: 3` kept (L=4 cannot feed the shipped prior)."""
    import copy
    o = copy.deepcopy(dict(opt))
    o["scale"] = scale
    o["network_G"] = copy.deepcopy(dict(o["network_G"]))
    o["network_G"]["upscale"] = scale
    o["network_G"]["scale"] = scale
    return dict_to_nonedict(o)


DEFAULT_CONF = osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), "confs", "SRFlow-LP_DF2K_4X.yml")
