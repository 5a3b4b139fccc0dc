"""The trainer's call sequence (main_train_restoration_with_ref_input.py:138-314) on the GPU through the drop-in
shims: option file -> datasets -> EnlargedSampler -> dataloader -> CUDAPrefetcher -> create_model ->
{update_learning_rate, feed_train_data, optimize_parameters, get_current_learning_rate, get_current_log, save,
validation} -> resume.  The loop below is this repo's own restatement of that flow (the reference checkout does not
exist on the GPU box; tests/test_dropin_shims.py runs the unchanged script itself in the build container)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import math, os, sys, time, torch
from data import create_dataloader, create_dataset
from data.data_sampler import EnlargedSampler
from data.prefetch_dataloader import CPUPrefetcher, CUDAPrefetcher
from models import create_model
from utils.logger import MessageLogger, get_root_logger
from utils.utils_misc import check_resume, set_random_seed, make_exp_dirs
from utils.utils_dist import get_dist_info
from utils.utils_options import parse

opt = parse(sys.argv[1], is_train=True)
opt['dist'] = False
opt['rank'], opt['world_size'] = get_dist_info()
set_random_seed(opt['manual_seed'] + opt['rank'])
states = sorted(int(f[:-6]) for f in os.listdir(opt['path']['training_states'])) if os.path.isdir(opt['path']['training_states']) else []
resume_state = None
if states:
    opt['path']['resume_state'] = os.path.join(opt['path']['training_states'], f'{states[-1]}.state')
    resume_state = torch.load(opt['path']['resume_state'], map_location=lambda s, l: s.cuda(torch.cuda.current_device()))
else:
    make_exp_dirs(opt)
logger = get_root_logger(log_file=os.path.join(opt['path']['log'], 'train.log'))
tr, va = opt['datasets']['train'], opt['datasets']['val']
train_set = create_dataset(tr)
sampler = EnlargedSampler(train_set, opt['world_size'], opt['rank'], tr.get('dataset_enlarge_ratio', 1))
tr['prefetch_mode'], tr['pin_memory'] = 'cuda', True
train_loader = create_dataloader(train_set, tr, num_gpu=opt['num_gpu'], dist=opt['dist'], sampler=sampler, seed=opt['manual_seed'])
val_loader = create_dataloader(create_dataset(va), va, num_gpu=opt['num_gpu'], dist=opt['dist'], sampler=None, seed=opt['manual_seed'])
total_iters = int(opt['train']['total_iter'])
if resume_state:
    check_resume(opt, resume_state['iter'])
    model = create_model(opt)
    model.resume_training(resume_state)
    epoch, it = resume_state['epoch'], resume_state['iter']
    total_iters = it + 3
else:
    model = create_model(opt)
    epoch, it = 0, 0
msg = MessageLogger(opt, it, None)
prefetcher = CUDAPrefetcher(train_loader, opt)
losses = []
while it <= total_iters:
    sampler.set_epoch(epoch)
    prefetcher.reset()
    data = prefetcher.next()
    while data is not None:
        it += 1
        if it > total_iters:
            break
        model.update_learning_rate(it, warmup_iter=opt['train'].get('warmup_iter', -1))
        model.feed_train_data({'lq': data['lq'], 'gt': data['gt'], 'ref': data['ref']})
        model.optimize_parameters(it)
        if it % opt['logger']['print_freq'] == 0:
            log = {'epoch': epoch, 'iter': it, 'lrs': model.get_current_learning_rate(), 'time': 0.0, 'data_time': 0.0}
            log.update(model.get_current_log())
            losses.append(log['l_pix'])
            msg(log)
        if it % opt['logger']['save_checkpoint_freq'] == 0:
            model.save(epoch, it)
        if it % opt['val']['val_freq'] == 0:
            psnr = model.validation(val_loader, it, None, opt['val']['save_img'], opt['val'].get('rgb2bgr', True), opt['val'].get('use_image', True))
            print('VAL', it, psnr)
        data = prefetcher.next()
    epoch += 1
model.save(epoch=-1, current_iter=-1)
print('LOSSES', ' '.join(f'{v:.6f}' for v in losses))
print('DONE', it - 1, model.optimizer_g.applied_steps())
'''


def _run(tmp_path, yml):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'dropin'), ROOT]), TDR_EXPERIMENTS_ROOT=str(tmp_path))
    drv = tmp_path / 'driver.py'
    drv.write_text(DRIVER)
    out = subprocess.run([sys.executable, str(drv), yml], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    return out.stdout + out.stderr


def test_trainer_flow_train_validate_save_resume(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    yml = os.path.join(ROOT, 'tests', 'data', 'train_nafnet_ref_synthetic_debug.yml')
    log = _run(tmp_path, yml)
    exp = tmp_path / 'experiments' / 'debug_nafnet_ref_synthetic'
    losses = [float(v) for v in log.split('LOSSES')[1].splitlines()[0].split()]
    assert len(losses) == 12 and all(0.0 < v < 1.0 for v in losses)
    assert sum(losses[-4:]) < sum(losses[:4])                  # it trains
    assert 'VAL 8' in log and 'Validation psnr:' in log
    assert 'DONE 12 12' in log
    for f in ('models/net_g_8.pth', 'models/net_g_latest.pth', 'training_states/8.state', 'train.log'):
        assert (exp / f).exists(), f
    # second launch: picks up training_states/8.state like the trainer's auto-resume (:138-158), continues at iter 9
    log2 = _run(tmp_path, yml)
    assert 'DONE 11 11' in log2                               # 8 restored steps + 3 new ones
    l2 = [float(v) for v in log2.split('LOSSES')[1].splitlines()[0].split()]
    assert len(l2) == 3 and all(0.0 < v < 1.0 for v in l2)
