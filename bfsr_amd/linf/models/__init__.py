from .models import register, make, models  # noqa: F401
from . import parts, unet, linf  # noqa: F401  (populate the registry: rrdb, edsr-baseline, flow, unet, linf, linf-patch)
