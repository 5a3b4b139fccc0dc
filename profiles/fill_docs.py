"""Fill the @@PLACEHOLDER@@ numbers of DESIGN.md / README.md from the files profiles/r5_measure.sh wrote (gpurun_out/r5m or profiles/r5).
usage: python profiles/fill_docs.py <dir with the logs>"""
import json, re, sys, os
d = sys.argv[1]


def line(name):
    with open(os.path.join(d, name)) as fh:
        rows = [ln for ln in fh.read().splitlines() if ln.startswith('{')]
    return json.loads(rows[-1])


b = line('bench_default.log')
ks = open(os.path.join(d, 'rocprofv3_kernel_summary_steps.txt')).read().splitlines()
m = re.search(r'total kernel time ([\d.]+) ms/step.*; (\d+) launches/step', ks[0])
by_kernel = []
for ln in ks[2:]:
    if ln.startswith('--- by kernel and grid'):
        break
    mm = re.match(r'\s*([\d.]+) ms/step\s+[\d.]+%\s+n/step\s+([\d.]+)\s+avg\s+([\d.]+) us\s+(.*)', ln)
    if mm:
        by_kernel.append((float(mm.group(1)), mm.group(4)))
tot = lambda *pats: sum(v for v, n in by_kernel if any(p in n for p in pats))
vals = {
    'DEFAULT_MS': f"{b['ms_per_step']:.1f}", 'DEFAULT_IPS': f"{b['value']:.1f}",
    'HBM_FRAC': f"{b['roofline_step']['achieved_hbm_frac']:.2f}", 'MEAS_FRAC': f"{b['roofline_step'].get('measured_hbm_frac', 0):.2f}",
    'MEAS_GB': f"{b['roofline_step'].get('measured_hbm_bytes', 0) / 1e9:.0f}",
    'ROOF_FRAC': f"{b['roofline']['frac']:.2f}", 'ROOF_US': f"{b['roofline']['avg_launch_ms'] * 1e3:.0f}",
    'KTOT': m.group(1), 'NLAUNCH': m.group(2),
    'WG1_MS': f"{tot('wgrad1x1_', 'wgrad_reduce_kernel'):.1f}", 'CONV_MS': f"{tot('conv3x3_p16_kernel'):.1f}",
    'WG3_MS': f"{tot('wgrad3x3_p16_kernel', 'wgrad_p16_reduce_kernel'):.1f}",
    'WG1_VERDICT': 'met' if tot('wgrad1x1_', 'wgrad_reduce_kernel') <= 5.0 else 'not met',
    'FAST_MS': f"{b['fast_mode']['ms_per_step']:.1f}",
    'RESTORMER': f"{line('bench_restormer_cfg3.log')['ms_per_step']:.1f}", 'RESTORMER5': f"{line('bench_restormer_cfg5.log')['ms_per_step']:.1f}",
    'PROMPTIR': f"{line('bench_promptir_384_bs8.log')['ms_per_step']:.0f}", 'DRS': f"{line('bench_drsformer_256_bs8.log')['ms_per_step']:.0f}",
    'DRSM': f"{line('bench_drsformer_mefc_256_bs8.log')['ms_per_step']:.0f}", 'DINO': f"{line('bench_dino640.log')['ms_per_step']:.1f}",
    'I2T': f"{line('bench_i2t_step.log')['ms_per_step']:.1f}", 'TR': f"{line('bench_tr_step.log')['ms_per_step']:.1f}",
}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for fn in ('DESIGN.md', 'README.md'):
    p = os.path.join(root, fn)
    s = open(p).read()
    for k, v in vals.items():
        s = s.replace(f'@@{k}@@', v)
    left = re.findall(r'@@[A-Z0-9_]+@@', s)
    open(p, 'w').write(s)
    print(fn, 'unfilled:', left)
print(vals)
