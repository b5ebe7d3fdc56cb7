"""Ordered kernel list of ONE training iteration (or forward) from a rocprofv3 --kernel-trace CSV: python trace_iter.py <kernel_trace.csv> [delimiter kernel substring]
Prints every launch of the last complete iteration in order, ATen / runtime kernels marked, with durations."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
delim = sys.argv[2] if len(sys.argv) > 2 else 'sgd_momentum_kernel'
idx = [i for i, r in enumerate(rows) if delim in r['Kernel_Name']]
# an iteration ends with the LAST of a run of delimiter launches
ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] != i + 1]
a, b = ends[-2] + 1, ends[-1] + 1
tot = aten = 0.0
for r in rows[a:b]:
    n = r['Kernel_Name']
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    short = re.sub(r'\(anonymous namespace\)::', '', n)
    short = re.sub(r'^void ', '', short)[:110]
    mark = 'ATEN ' if ('at::native' in n or 'rocclr' in n) else '     '
    tot += us
    aten += us if mark.strip() else 0
    print('%s%8.1f us  %s' % (mark, us, short))
print('launches %d, kernel time %.3f ms, ATen / runtime kernels %.3f ms' % (b - a, tot / 1e3, aten / 1e3))
