"""Where the small fill / copy launches of a replayed step sit: for the LAST step of a rocprofv3 rocpd database (the kernels between the last two
adamw_kernel launches), every Fill / copyBuffer / fillBuffer launch with its size class and its neighbours in start order.
usage: python profiles/seq_fills.py <db>"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start'))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)[:60]


ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
lo, hi = ad[-2] + 1, ad[-1] + 1
step = rows[lo:hi]
print(f'{len(step)} launches in the last step')
agg = {}
for i, r in enumerate(step):
    if 'FillFunctor' in r[0] or 'copyBuffer' in r[0] or 'fillBuffer' in r[0]:
        prev = short(step[i - 1][0]) if i else '-'
        nxt = short(step[i + 1][0]) if i + 1 < len(step) else '-'
        key = (short(r[0])[:40], r[3], prev, nxt)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e3
for (n, g, p, x), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{cnt:4d} x {us / cnt:6.1f} us  threads {g:9d}  {n:40s}  after {p:45s} before {x}')
