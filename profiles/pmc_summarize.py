"""Turn the two rocprofv3 --pmc passes of profiles/pmc_workload.py into per-kernel HBM traffic (JSON on stdout)."""
import csv
import glob
import json
import os
import re
import sys

CAL_BYTES = 128 * 1024 * 1024 * 4          # bytes read == bytes written by one calibration copy


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get('Counter_Name') == counter:
                    rows.append((r['Kernel_Name'], float(r['Counter_Value'])))
    return rows


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*', '', n)


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {'units': 'bytes per launch (mean over the launches of the last profiled step set)', 'kernels': {}}

    def cal(rows, pat):
        v = [x for k, x in rows if pat in k]          # the calibration copies are the first launches of the run
        return (sum(v[:2]) / max(len(v[:2]), 1)) if v else None
    f4, f1 = cal(fetch, ' (anonymous namespace)::rows_kernel<false>'), cal(fetch, 'rows_scalar_kernel<false>')
    w4, w1 = cal(write, ' (anonymous namespace)::rows_kernel<false>'), cal(write, 'rows_scalar_kernel<false>')
    # counters are reported in KiB-like units of 1024 B? derive the factor instead of assuming: bytes / counter
    out['calibration'] = {
        'copy_bytes': CAL_BYTES,
        'FETCH_SIZE_per_copy_float4': f4, 'FETCH_SIZE_per_copy_dword': f1,
        'WRITE_SIZE_per_copy_float4': w4, 'WRITE_SIZE_per_copy_dword': w1,
        'bytes_per_FETCH_unit_float4': CAL_BYTES / f4 if f4 else None, 'bytes_per_FETCH_unit_dword': CAL_BYTES / f1 if f1 else None,
        'bytes_per_WRITE_unit_float4': CAL_BYTES / w4 if w4 else None, 'bytes_per_WRITE_unit_dword': CAL_BYTES / w1 if w1 else None,
    }
    fu = out['calibration']['bytes_per_FETCH_unit_dword'] or 0.0
    wu = out['calibration']['bytes_per_WRITE_unit_dword'] or 0.0
    agg = {}
    for k, v in fetch:
        a = agg.setdefault(short(k), [0.0, 0, 0.0, 0])
        a[0] += v; a[1] += 1
    for k, v in write:
        a = agg.setdefault(short(k), [0.0, 0, 0.0, 0])
        a[2] += v; a[3] += 1
    for k, (fs, fn, ws, wn) in sorted(agg.items(), key=lambda kv: -(kv[1][0] * fu + kv[1][2] * wu)):
        out['kernels'][k] = {'launches': fn, 'read_bytes_per_launch': fs / max(fn, 1) * fu,
                             'write_bytes_per_launch': ws / max(wn, 1) * wu}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
