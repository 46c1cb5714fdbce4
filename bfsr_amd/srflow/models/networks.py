"""Generator factory (reference SRFlow-LP/code/models/networks.py:27-43,70-78): the class is looked up by
name -- `network_G.which_model_G: SRFlowNet` resolves to module `models.modules.SRFlowNet_arch`."""
import importlib


def find_model_using_name(model_name):
    modellib = importlib.import_module(__package__ + ".modules." + model_name + "_arch")
    target = model_name.replace('_Net', '').lower()
    for name, cls in modellib.__dict__.items():
        if name.lower() == target and isinstance(cls, type):
            return cls
    raise NotImplementedError("no class matching %s in %s_arch" % (model_name, model_name))


def define_Flow(opt, step, ops=None):
    opt_net = opt['network_G']
    Arch = find_model_using_name(opt_net['which_model_G'])
    return Arch(in_nc=opt_net['in_nc'], out_nc=opt_net['out_nc'], nf=opt_net['nf'], nb=opt_net['nb'],
                scale=opt['scale'], K=opt_net['flow']['K'], opt=opt, step=step, ops=ops)
