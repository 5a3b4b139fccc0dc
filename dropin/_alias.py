"""Import shims: with `<repo>/dropin` (and `<repo>`) first on sys.path, the reference trainer's own import lines

    from data import create_dataloader, create_dataset           (main_train_restoration_with_ref_input.py:10)
    from data.data_sampler import EnlargedSampler                (:11)
    from data.prefetch_dataloader import CPUPrefetcher, CUDAPrefetcher   (:12)
    from models import create_model                              (:13)
    from utils.logger import MessageLogger, get_root_logger, get_env_info, init_tb_logger, init_wandb_logger   (:14)
    from utils.utils_misc import check_resume, set_random_seed, get_time_str, make_exp_dirs, mkdir_and_rename  (:15)
    from utils.utils_dist import get_dist_info, init_dist        (:17)
    from utils.utils_options import dict2str, parse              (:18)

resolve to textualdegremoval_amd -- the script itself stays unchanged.  Each top-level name (`models`, `data`,
`utils`) and every submodule below it is registered in sys.modules as the SAME module object as its
textualdegremoval_amd counterpart (no second copy of any class)."""
import importlib
import pkgutil
import sys


def alias(top, target):
    pkg = importlib.import_module(target)
    sys.modules[top] = pkg
    for info in pkgutil.walk_packages(pkg.__path__, prefix=target + '.'):
        try:
            mod = importlib.import_module(info.name)
        except ImportError:          # optional third-party dependency of a submodule that is not installed here
            continue
        sys.modules[top + info.name[len(target):]] = mod
    return pkg
