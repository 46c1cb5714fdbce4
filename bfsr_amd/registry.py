"""Name -> factory registry: the plugin surface of both reference sub-projects
(`@register(name)` / `make(spec, args=None, load_sd=False)`; LINF-LP/models/models.py:7-23 and the identical
SRFlow-LP/code/models/models.py:7-23).  One implementation, instantiated once per sub-project so the two name
spaces stay separate ('unet' means a different network in each)."""
import copy


class Registry(object):
    def __init__(self):
        self.table = {}

    def register(self, name):
        """Decorator: `@register('linf-patch')` records a class or factory function under `name`."""
        def bind(factory):
            self.table[name] = factory
            return factory
        return bind

    def make(self, model_spec, args=None, load_sd=False):
        """Instantiate `model_spec = {'name', 'args'[, 'sd']}`.  `args` (if given) overlays a deep copy of the spec's
        own kwargs; `load_sd` loads the embedded state dict (checkpoint layout of LINF-LP/train.py:234-244)."""
        kwargs = model_spec['args']
        if args is not None:
            kwargs = dict(copy.deepcopy(kwargs), **args)
        try:
            factory = self.table[model_spec['name']]
        except KeyError:
            raise KeyError("no model registered under %r (known: %s)" % (model_spec['name'], sorted(self.table)))
        instance = factory(**kwargs)
        if load_sd:
            instance.load_state_dict(model_spec['sd'])
        return instance
