"""Architecture registry -- mirrors models/archs/__init__.py:9-46 of the reference:
`define_network(opt)` pops 'type' and instantiates the class of that name from
every `*_arch.py` module in this folder; unknown names raise ValueError."""
import importlib
import os

_folder = os.path.dirname(os.path.abspath(__file__))
_arch_modules = [importlib.import_module(f'{__name__}.{os.path.splitext(f)[0]}')
                 for f in sorted(os.listdir(_folder)) if f.endswith('_arch.py')]


def dynamic_instantiation(modules, cls_type, opt):
    cls_ = None
    for module in modules:
        cls_ = getattr(module, cls_type, None)
        if cls_ is not None:
            break
    if cls_ is None:
        raise ValueError(f'{cls_type} is not found.')
    return cls_(**opt)


def define_network(opt):
    network_type = opt.pop('type')
    return dynamic_instantiation(_arch_modules, network_type, opt)
