"""`create_model(opt)` -- mirrors models/__init__.py:22-43 of the reference: instantiate the
class named by opt['model_type'] from the *_model.py modules of this package."""
import importlib
import os

_folder = os.path.dirname(os.path.abspath(__file__))
_model_modules = [importlib.import_module(f'{__name__}.{os.path.splitext(f)[0]}')
                  for f in sorted(os.listdir(_folder)) if f.endswith('_model.py') and f != 'base_model.py']


def create_model(opt):
    model_type = opt['model_type']
    model_cls = None
    for module in _model_modules:
        model_cls = getattr(module, model_type, None)
        if model_cls is not None:
            break
    if model_cls is None:
        raise ValueError(f'Model {model_type} is not found.')
    return model_cls(opt)
