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
d['meta'] = {'tdr_conv_bx3_sha256': hashlib.sha256(open('textualdegremoval_amd/csrc/tdr_conv_bx3.hip', 'rb').read()).hexdigest(),
             'workload': 'profiles/pmc_workload.py (BASELINE configs[1], eager steps)', 'collected_by': 'profiles/pmc_collect.sh'}
json.dump(d, open(sys.argv[1], 'w'), indent=1)
print('wrote', sys.argv[1], len(d['kernels']), 'kernels')
PY
