"""`create_model(opt)` (reference SRFlow-LP/code/models/__init__.py:42-55): only `model: SRFlow` is on the
inference path."""
from . import models  # noqa: F401  (registry)
from . import unet  # noqa: F401    (registers 'unet')


def create_model(opt, step=0, ops=None, **opt_kwargs):
    if opt_kwargs:
        for k, v in opt_kwargs.items():
            opt[k] = v
    model = opt['model']
    if model == 'SRFlow':
        from .SRFlow_model import SRFlowModel as M
    else:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(model))
    return M(opt, step, ops=ops)
