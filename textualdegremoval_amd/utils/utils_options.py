"""YAML option files -> the `opt` dict the trainer and the models consume (reference utils/utils_options.py):
key order preserved, `is_train`, per-dataset `phase`/`scale`, user-expanded paths, and the experiment layout
`<root>/experiments/<name>/{models,training_states,visualization}` (or `<root>/results/<name>` for tests).
`root` is TDR_EXPERIMENTS_ROOT if set, else the current working directory (the reference uses its checkout)."""
import os
from collections import OrderedDict

import yaml


def _ordered_loader():
    base = getattr(yaml, 'CSafeLoader', yaml.SafeLoader)

    class Loader(base):
        pass
    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                           lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    return Loader


def parse(opt_path, is_train=True):
    with open(opt_path) as f:
        opt = yaml.load(f, Loader=_ordered_loader())
    opt['is_train'] = is_train
    for phase, ds in opt['datasets'].items():
        ds['phase'] = phase.split('_')[0]              # test_1, test_2 -> test
        if 'scale' in opt:
            ds['scale'] = opt['scale']
        for key in ('dataroot_gt', 'dataroot_lq'):
            if ds.get(key) is not None:
                ds[key] = os.path.expanduser(ds[key])
    for key, val in opt['path'].items():
        if val is not None and ('resume_state' in key or 'pretrain_network' in key):
            opt['path'][key] = os.path.expanduser(val)
    root = opt['path']['root'] = os.path.abspath(os.environ.get('TDR_EXPERIMENTS_ROOT', os.getcwd()))
    if is_train:
        exp = os.path.join(root, 'experiments', opt['name'])
        opt['path'].update(experiments_root=exp, models=os.path.join(exp, 'models'),
                           training_states=os.path.join(exp, 'training_states'), log=exp,
                           visualization=os.path.join(exp, 'visualization'))
        if 'debug' in opt['name']:
            if 'val' in opt:
                opt['val']['val_freq'] = 8
            opt['logger']['print_freq'] = 1
            opt['logger']['save_checkpoint_freq'] = 8
    else:
        res = os.path.join(root, 'results', opt['name'])
        opt['path'].update(results_root=res, log=res, visualization=os.path.join(res, 'visualization'))
    return opt


def dict2str(opt, indent_level=1):
    pad = ' ' * (indent_level * 2)
    out = ['\n']
    for k, v in opt.items():
        if isinstance(v, dict):
            out.append(f'{pad}{k}:[{dict2str(v, indent_level + 1)}{pad}]\n')
        else:
            out.append(f'{pad}{k}: {v}\n')
    return ''.join(out)
