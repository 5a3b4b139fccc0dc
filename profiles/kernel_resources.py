"""usage: python profiles/kernel_resources.py <file.hip> [extra hipcc flags]: per-kernel VGPRs / spills / scratch / occupancy (hipcc -Rpass-analysis)"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    print(f"{r.get('VGPRs','?'):>4} vgpr {r.get('AGPRs','?'):>3} agpr spill {r.get('VGPRs Spill','?'):>4} scratch {r.get('ScratchSize','?'):>5} occ {r.get('Occupancy','?')}  {r['name'][:110]}")
