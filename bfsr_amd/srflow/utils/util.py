"""utils/util.py surface used on the hot path: `opt_get` (reference utils/util.py:167-175)."""
from ..options.options import opt_get  # noqa: F401
