"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel and per-(kernel,grid) totals.
usage: python profiles/summarize_rocpd.py results.db [steps]
(the spin kernel bench.py uses to pre-fill the launch queue of its roofline leg is left out)"""
import re
import sqlite3
import sys


def short(n, k=70):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    return n[:k]


def main():
    db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    tot = list(c.execute('select sum(end-start) from kernels where name not like "%spin_kernel%"'))[0][0]
    nl = list(c.execute('select count(*) from kernels where name not like "%spin_kernel%"'))[0][0]
    print(f'total kernel time {tot / 1e6 / steps:.2f} ms/step over {steps:g} steps; {nl / steps:.0f} launches/step')
    print('--- by kernel')
    for name, n, t, avg in c.execute('select name, count(*), sum(end-start), avg(end-start) from kernels where name not like "%spin_kernel%" group by name '
                                     'order by sum(end-start) desc limit 140'):
        print(f'{t / 1e6 / steps:8.2f} ms/step {100 * t / tot:5.1f}%  n/step {n / steps:7.1f}  avg {avg / 1e3:8.1f} us  {short(name)}')
    print('--- by kernel and grid (workgroups x,y,z)')
    q = ('select name, grid_x/workgroup_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start), lds_size, vgpr_count '
         'from kernels where name not like "%spin_kernel%" group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit 50')
    for r in c.execute(q):
        print(f'{r[5] / 1e6 / steps:8.2f} ms/step {100 * r[5] / tot:5.1f}% n/step {r[4] / steps:6.1f} avg {r[6] / 1e3:8.1f}us '
              f'grid({r[1]},{r[2]},{r[3]}) lds {r[7]} vgpr {r[8]}  {short(r[0], 56)}')


if __name__ == '__main__':
    main()
