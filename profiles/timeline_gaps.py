"""Step timeline from a rocprofv3 rocpd database: how much of a replayed step is device-busy, how much is idle between kernels (launch /
dependency gaps of the graph), how much runs two kernels at once (weight-gradient branch), and which kernels the idle gaps precede.
usage: python profiles/timeline_gaps.py <rocpd.db> <launches-per-step> [steps-from-the-end]"""
import re
import sqlite3
import sys
from collections import defaultdict

db, per_step = sys.argv[1], int(sys.argv[2])
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = list(c.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
rows = rows[-per_step * nsteps:]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*', '', n)[:80]


t0, t1 = rows[0][1], max(r[2] for r in rows)
busy = 0
overlap = 0
gaps = defaultdict(lambda: [0, 0.0])
cur_end = rows[0][1]
prev_name = None
ev = []
for r in rows:
    ev.append((r[1], 1))
    ev.append((r[2], -1))
    if r[1] > cur_end:
        g = gaps[(short(prev_name) if prev_name else '-', short(r[0]))]
        g[0] += 1
        g[1] += r[1] - cur_end
    if r[2] > cur_end:
        cur_end, prev_name = r[2], r[0]
ev.sort()
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        overlap += t - last
    depth += d
    last = t
span = t1 - t0
idle = span - busy
print(f'{nsteps} steps x {per_step} launches: span {span / nsteps / 1e6:.3f} ms/step, busy {busy / nsteps / 1e6:.3f}, idle {idle / nsteps / 1e6:.3f}, '
      f'two-or-more kernels at once {overlap / nsteps / 1e6:.3f} ms/step; sum of kernel durations {sum(r[2] - r[1] for r in rows) / nsteps / 1e6:.3f}')
if qcol:
    per_q = defaultdict(float)
    for r in rows:
        per_q[r[3]] += r[2] - r[1]
    print('kernel time by ' + qcol + ': ' + ', '.join(f'{q}: {v / nsteps / 1e6:.2f} ms' for q, v in sorted(per_q.items(), key=lambda kv: -kv[1])))
print('--- idle gaps by (kernel that ended last -> kernel that started), ms/step')
for (a, b), (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'  {tot / nsteps / 1e6:7.3f} ms  n/step {n / nsteps:6.1f}  avg {tot / n / 1e3:6.1f} us   {a}  ->  {b}')
hist = defaultdict(lambda: [0, 0.0])
for (a, b), (n, tot) in gaps.items():
    k = 'lt2us' if tot / n < 2e3 else ('2-5us' if tot / n < 5e3 else ('5-20us' if tot / n < 2e4 else 'gt20us'))
    hist[k][0] += n
    hist[k][1] += tot
print('--- gap size classes (by mean of the pair): ' + ', '.join(f'{k}: {v[0] / nsteps:.0f}/step {v[1] / nsteps / 1e6:.3f} ms' for k, v in hist.items()))
