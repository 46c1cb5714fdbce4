"""name -> class registry + factory: the plugin API of the reference
(SRFlow-LP/code/models/models.py:4-23, identical in LINF-LP/models/models.py:4-23)."""
import copy

models = {}


def register(name):
    def decorator(cls):
        models[name] = cls
        return cls
    return decorator


def make(model_spec, args=None, load_sd=False):
    """model_spec = {'name', 'args', 'sd'}; `args` overlays a deep copy of model_spec['args']."""
    if args is not None:
        model_args = copy.deepcopy(model_spec['args'])
        model_args.update(args)
    else:
        model_args = model_spec['args']
    model = models[model_spec['name']](**model_args)
    if load_sd:
        model.load_state_dict(model_spec['sd'])
    return model
