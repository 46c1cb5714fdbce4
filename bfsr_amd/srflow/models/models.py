"""The `models` registry of this sub-project (reference: models/models.py:4-23): `models` is the name -> factory
table, `register` the decorator, `make` the factory."""
from ...registry import Registry

_registry = Registry()
models = _registry.table
register = _registry.register
make = _registry.make
