#!/bin/bash
# usage (GPU box, repo root): profiles/pmc_collect.sh [out.json]     (default profiles/r2/pmc_traffic.json under gpurun_out/)
# Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: the TCC block cannot hold both) of profiles/pmc_workload.py -- calibration
# copies of known size, then eager train steps of BASELINE configs[1] -- summarised into HBM bytes per launch per kernel
# (pmc_summarize.py: calibrated as MI355X_MICROARCH.md prescribes), tagged with the hash of the conv kernel source so that
# bench.py can tell a stale file from a current one.
out=${1:-gpurun_out/pmc_traffic.json}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_fetch /tmp/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python $root/profiles/pmc_workload.py > /tmp/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python $root/profiles/pmc_workload.py > /tmp/pmc_write.log 2>&1
cd "$root" && python profiles/pmc_summarize.py /tmp/pmc_fetch /tmp/pmc_write > /tmp/pmc_traffic_raw.json && python - "$out" <<'PY'
import hashlib, json, sys
d = json.load(open('/tmp/pmc_traffic_raw.json'))
sha = lambda fn: hashlib.sha256(open('textualdegremoval_amd/csrc/' + fn, 'rb').read()).hexdigest()
# whole-step HBM bytes: every kernel of the 3 profiled eager steps except the calibration copies and the fp16-window survey probes
# (absmax_bits_*: they run in surveyed eager steps only, one step in 1000 of a training run)
skip = ('rows_kernel', 'rows_scalar_kernel', 'absmax_bits')
step_total = sum((v['read_bytes_per_launch'] + v['write_bytes_per_launch']) * v['launches'] for k, v in d['kernels'].items()
                 if not any(s in k for s in skip)) / 3.0
d['meta'] = {'tdr_conv_bx3_sha256': sha('tdr_conv_bx3.hip'), 'tdr_conv_p16_sha256': sha('tdr_conv_p16.hip'),
             'step_total_bytes': step_total, 'steps_profiled': 3,
             'workload': 'profiles/pmc_workload.py (BASELINE configs[1], eager steps)', 'collected_by': 'profiles/pmc_collect.sh'}
json.dump(d, open(sys.argv[1], 'w'), indent=1)
print('wrote', sys.argv[1], len(d['kernels']), 'kernels')
PY
