#!/bin/bash
# usage (GPU box, repo root): profiles/pmc_clock.sh <out.txt> -- <command...>
# One rocprofv3 --pmc pass (GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES; kernels run serialised) + the dispatch timestamps of the
# same pass: per kernel the effective shader clock under its own load (GRBM_GUI_ACTIVE / duration, MI355X_MICROARCH.md "DVFS give-back") and
# the share of its CYCLES in which the matrix pipes are busy -- the clock-independent reading of an MFMA roofline fraction.
out=$1; shift 2      # (<out.txt>; the same numbers as JSON next to it: <out>.json)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tdr_pmc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/tdr_pmc -o c -- "$@" > /tmp/tdr_pmc_cmd.log 2>&1
cd "$root" && python - "$out.json" > "$out" <<'PY'
import csv, glob, re
from collections import defaultdict
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k); k = re.sub(r'^void ', '', k); return re.sub(r'\(.*', '', k)
dur = {}
for f in glob.glob('/tmp/tdr_pmc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
acc = defaultdict(lambda: defaultdict(list))
cols = None
for f in glob.glob('/tmp/tdr_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cols = cols or list(r.keys())
        key = (short(r['Kernel_Name']), r['Grid_Size'])
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            d = dur.get(r['Dispatch_Id'])
            if d is None and 'Start_Timestamp' in r:
                d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            if d:
                acc[key]['_ns'].append(d)
print('# columns of the counter file:', cols)
print('# kernel | grid | launches | avg us | GRBM_GUI_ACTIVE | SQ_BUSY_CYCLES | SQ_VALU_MFMA_BUSY_CYCLES | GUI_ACTIVE / ns | MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / XCDS)')
rows = []
for (k, g), d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    ns = m.get('_ns', 0.0)
    rows.append((ns * len(d.get('_ns', [])), k, g, len(d['GRBM_GUI_ACTIVE']), ns, m.get('GRBM_GUI_ACTIVE', 0), m.get('SQ_BUSY_CYCLES', 0), m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)))
for tot, k, g, n, ns, gui, sqb, mf in sorted(rows, reverse=True)[:70]:
    print(f'{k[:78]:78s} {g:>9s} {n:4d} {ns / 1e3:9.1f} {gui:14.0f} {sqb:14.0f} {mf:14.0f} {gui / ns if ns else 0:8.3f} {mf / 1024 / gui if gui else 0:8.4f}')
# GRBM_GUI_ACTIVE is summed over the 8 XCDs and carries a fixed launch share: the shortest dispatches (1 - 2 us fills) give it
import json, sys
tiny = sorted((gui, ns) for tot, k, g, n, ns, gui, sqb, mf in rows if 0 < ns < 3000 and n >= 20)
fixed = tiny[0][0] if tiny else 0.0
js = {'note': 'per dispatch: cycles = (GRBM_GUI_ACTIVE - fixed) / 8 XCDs; effective_ghz = cycles / duration_ns; mfma_busy_cycle_frac = '
              'SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles (kernels serialised by the counter pass)', 'fixed_gui_active': fixed, 'kernels': []}
for tot, k, g, n, ns, gui, sqb, mf in sorted(rows, reverse=True)[:70]:
    cyc = (gui - fixed) / 8.0
    if ns > 20000 and cyc > 0:
        js['kernels'].append({'kernel': k, 'grid': int(g), 'launches': n, 'avg_us': ns / 1e3, 'effective_ghz': cyc / ns,
                              'mfma_busy_cycle_frac': mf / 1024.0 / cyc})
json.dump(js, open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/tdr_pmc_clock.json', 'w'), indent=1)
PY
