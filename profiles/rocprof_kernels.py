"""per-(kernel, grid) launch statistics of the kernels whose name contains argv[2], from a rocprofv3 rocpd database"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
q = ('select name, grid_x/workgroup_x, grid_y, grid_z, count(*), avg(end-start), min(end-start) from kernels where name like ? '
     'group by name, grid_x, grid_y, grid_z order by sum(end-start) desc')
for r in c.execute(q, (f'%{sys.argv[2]}%',)):
    name = re.sub(r'\(anonymous namespace\)::', '', r[0]); name = re.sub(r'\(.*', '', name)
    print(f'n {r[4]:5d} avg {r[5] / 1e3:8.1f} us min {r[6] / 1e3:8.1f} us grid({r[1]},{r[2]},{r[3]})  {name[:70]}')
