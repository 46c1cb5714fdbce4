from .options import (DEFAULT_CONF, NoneDict, derive_scale, dict_to_nonedict, load, opt_get, parse)  # noqa: F401
