"""print per-kernel counter means from a rocprofv3 --pmc csv dir"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'^void ', '', k); k = re.sub(r'\(.*', '', k)
        if 'conv_bx3' in k or 'wgrad_bx3' in k:
            acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), d in acc.items():
    print(k, 'grid', g)
    for c, v in sorted(d.items()):
        print(f'   {c:34s} {sum(v) / len(v):16.1f}  (n={len(v)})')
