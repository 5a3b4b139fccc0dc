#!/bin/bash
# usage (GPU box, repo root): profiles/pmc_run.sh <out.txt> "<kernel substring>" "<counters...>" -- <command...>
# one rocprofv3 --pmc pass (no trace options: gpurun refuses the combination), per-kernel counter means
out=$1; pat=$2; ctrs=$3; shift 4
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tdr_pmc
rocprofv3 --pmc $ctrs --output-format csv -d /tmp/tdr_pmc -o c -- "$@" > /tmp/tdr_pmc_cmd.log 2>&1
cd "$root" && python - "$pat" > "$out" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/tdr_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'^void ', '', k); k = re.sub(r'\(.*', '', k)
        if sys.argv[1] in k:
            acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), d in acc.items():
    print(k, 'grid', g)
    for c, v in sorted(d.items()):
        print(f'   {c:34s} {sum(v) / len(v):16.1f}  (n={len(v)})')
PY
