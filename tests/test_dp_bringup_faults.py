"""Bring-up of the RCCL data plane behind the C ABI (textualdegremoval_amd/parallel.py::_resolve_plane) under injected faults,
on CPU with gloo as the side channel and a stand-in communicator class (the sequence of side-channel collectives is what is
under test; RCCL itself cannot run without GPUs).  The first real N-GPU run happens on a node this repository never sees, so:
  * a failure that comes back as an error on ANY rank must surface as DataPlaneUnavailable on EVERY rank (strict mode), never as
    ranks parked in mismatched collectives;
  * a rank that hangs / dies inside the bring-up must take the job down within TDR_COMM_INIT_TIMEOUT seconds with a one-line
    diagnosis on stderr and exit code 3 on the surviving ranks (the reference relies on torch's collective timeout:
    utils/utils_dist.py:21-58, models/base_model.py:76-82)."""
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
import torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
from textualdegremoval_amd import parallel as PL


class FakeComm:                                  # stands in for TdrComm: same constructor / class surface
    made = 0

    def __init__(self, rank, world, uid):
        assert uid == b'x' * 128
        self.rank, self.world, self.handle = rank, world, object()
        FakeComm.made += 1

    @staticmethod
    def new_unique_id():
        return b'x' * 128

    def destroy(self):
        self.handle = None


PL.TdrComm = FakeComm
PL._comm_available = lambda: 1
out = {{'rank': rank}}
try:
    comm = PL._resolve_plane(None)
    out['comm'] = type(comm).__name__
except PL.DataPlaneUnavailable as e:
    out['unavailable'] = str(e)
print('RESULT ' + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _run(world, env, timeout=120):
    port = _free_port()
    code = WORKER.format(root=ROOT)
    procs = []
    for r in range(world):
        e = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), TDR_COMM='rccl', **env)
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=max(1.0, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            res.append((None, so, se))
            continue
        res.append((p.returncode, so, se))
    return res, time.time() - t0


def _results(res):
    import json
    out = []
    for rc, so, se in res:
        lines = [ln for ln in so.splitlines() if ln.startswith('RESULT ')]
        out.append(json.loads(lines[-1][7:]) if lines else None)
    return out


@pytest.mark.timeout(180)
def test_bring_up_succeeds_on_every_rank():
    res, _ = _run(3, {})
    assert [rc for rc, _, _ in res] == [0, 0, 0], [se[-400:] for _, _, se in res]
    assert all(r and r.get('comm') == 'FakeComm' for r in _results(res))


@pytest.mark.timeout(180)
def test_rank0_failing_before_the_broadcast_is_an_error_on_every_rank():
    res, dt = _run(3, {'TDR_FAULT': 'unique_id:0'})
    assert [rc for rc, _, _ in res] == [0, 0, 0], [se[-400:] for _, _, se in res]
    for r in _results(res):
        assert r and 'ncclGetUniqueId failed on rank 0' in r.get('unavailable', ''), r
    assert dt < 60


@pytest.mark.timeout(180)
def test_one_rank_failing_init_is_an_error_on_every_rank():
    res, dt = _run(3, {'TDR_FAULT': 'init:1'})
    assert [rc for rc, _, _ in res] == [0, 0, 0], [se[-400:] for _, _, se in res]
    for r in _results(res):
        assert r and 'ncclCommInitRank failed on some rank' in r.get('unavailable', ''), r
    assert dt < 60


@pytest.mark.timeout(180)
def test_a_rank_hanging_inside_init_takes_the_job_down_with_a_diagnosis():
    res, dt = _run(3, {'TDR_FAULT': 'init:1:hang', 'TDR_COMM_INIT_TIMEOUT': '4'}, timeout=90)
    assert dt < 60, dt
    for rank, (rc, so, se) in enumerate(res):
        assert rc == 3, (rank, rc, se[-400:])
        line = [ln for ln in se.splitlines() if ln.startswith('[tdr] rank')]
        assert len(line) == 1 and 'did not finish within 4 s' in line[0] and f'rank {rank}/3' in line[0], se[-400:]
    # the hanging rank names the call it is stuck in, its peers the agreement they wait in
    assert 'ncclCommInitRank' in res[1][2]
    assert 'agreeing on the outcome' in res[0][2]
