"""GPU smoke test of the multi-process step: two ranks (sharing the one GPU of the test box, gloo transport) run the
captured-graph train step of bench.py -- exercises what the 8-GPU RCCL run uses except the transport itself:
process-group + hipGraph capture, deferred flat all-reduce between the two graphs, loss reduce to rank 0."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(nproc, port, wrap):
    """wrap=False: exactly `python bench.py --gpus N ...` (the script launches its own ranks); wrap=True: the driver's form,
    `python -m torch.distributed.run ... bench.py --gpus N ...`"""
    cmd = [sys.executable]
    if wrap:
        cmd += ['-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
                '--master-port', str(port)]
    cmd += [os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc), '--steps', '2', '--warmup', '1', '--width', '8', '--enc', '1,1,1,1',
            '--size', '128', '--batch', '1', '--backend', 'gloo', '--bucket-mb', '0.25',
            '--no-cpu-baseline']     # roofline leg on: every rank must run its instrumented step
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    return json.loads(line)


@pytest.mark.parametrize('wrap', [False, True])
def test_two_rank_graph_step_runs(wrap):
    """`python bench.py --gpus 2` with no launcher around it must produce a 2-rank line (and so must the driver's
    torch.distributed.run form); the gradient exchange is per bucket between the segments of the captured backward."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    r2 = _bench(2, 29541, wrap)
    assert r2['n_gpus'] == 2 and r2['config']['ranks_seen'] == 2
    assert r2['value'] > 0 and r2['final_loss'] == r2['final_loss']      # finite
    assert r2['config']['global_batch'] == 2
    assert 'buckets' in r2['config']['grad_exchange']
    assert r2['guard'] is None or r2['guard']['skipped_in_timed_region'] == 0       # (None: the default arithmetic has no step verdict)
    # the N-GPU line diagnoses itself: one record per rank, identical averaged gradients and parameters on every rank, the exchange
    # timed on the comm stream (here gloo on one GPU: the numbers are functional only)
    sd = r2['scale_diagnostics']
    assert [r['rank'] for r in sd['per_rank']] == [0, 1] and sd['all_ranks_see_world']
    assert sd['gradients_identical_across_ranks'] and sd['parameters_identical_across_ranks']
    assert all(r['buckets'] >= 1 and r['ms_per_step_local'] > 0 for r in sd['per_rank'])


def test_stage_a_step_two_ranks():
    """C4 (accelerate's DDP over the Mapper, main_train_i2t_mapping.py:662): `bench.py --arch i2t --gpus 2` -- two ranks sharing the GPU
    over gloo run the captured stage-A step with the gradient exchange of parallel.GradAllReducer between its two graphs"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--arch', 'i2t', '--clip', 'L', '--batch', '1', '--steps', '2',
           '--warmup', '0', '--backend', 'gloo']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert r['n_gpus'] == 2 and r['config']['global_batch'] == 2 and r['final_loss'] == r['final_loss']
    assert r['guard']['skipped_total'] == 0


def test_gpus_flag_must_match_the_launched_world():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '0']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(os.environ, WORLD_SIZE='2', RANK='0'))
    assert out.returncode != 0 and 'WORLD_SIZE=2' in out.stderr


def test_single_rank_rccl_collectives_under_graph_capture():
    """One rank, backend nccl (= RCCL), TDR_FORCE_COLLECTIVES=1: the bucketed all-reduces of the eager steps (comm
    stream, ReduceOp.AVG), the flat all-reduce between the two captured graphs and the loss reduce all go through
    RCCL on the real GPU -- everything of the N-GPU path except a second peer."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', '29547', os.path.join(ROOT, 'tests', '_rccl_single_rank.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, TDR_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    assert 'RCCL_SINGLE_RANK_OK' in out.stdout


def test_rccl_dry_run_one_rank():
    """`bench.py --rccl-dry-run`: the staged RCCL bring-up + broadcast + all-reduce + a timed 64 MiB bucket exchange on
    a second stream, here as a one-rank communicator on the GPU at hand (TDR_FORCE_COLLECTIVES=1); and its failure path names the stage."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(TDR_FORCE_COLLECTIVES='1', MASTER_PORT='29531')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--rccl-dry-run'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert r['rccl_dry_run'] == 'ok' and all(s['ok'] for s in r['stages']) and len(r['stages']) == 5
    # failure path: an injected fault inside the bring-up -> exit code 2 and ONE line naming the stage
    env.update(TDR_FAULT='init:0', MASTER_PORT='29532')
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--rccl-dry-run'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert bad.returncode == 2 and "FAILED at stage 'RCCL communicator bring-up" in bad.stderr, (bad.returncode, bad.stderr[-1500:])
