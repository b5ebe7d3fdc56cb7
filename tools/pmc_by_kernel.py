#!/usr/bin/env python3
"""HBM rate per kernel instantiation from the separate FETCH_SIZE / WRITE_SIZE passes of `tools/gpu.sh <tag> pmc:<bench args>`:
   python tools/pmc_by_kernel.py gpurun_out/<tag> > profiles/r05/<name>_hbm_by_kernel.txt
Durations are those of the FETCH_SIZE pass's own kernel trace (counter collection serialises the launches, --pipeline 1); bytes are
FETCH_SIZE KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB, per launch."""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', n)


def main():
    d = sys.argv[1].rstrip('/') + '/'
    tr = collections.defaultdict(list)
    for r in csv.DictReader(open(d + 'pmc_fetch/r1_kernel_trace.csv')):
        tr[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    fe, wr = collections.defaultdict(float), collections.defaultdict(float)
    for nm, dst in (('pmc_fetch', fe), ('pmc_write', wr)):
        for r in csv.DictReader(open(d + nm + '/r1_counter_collection.csv')):
            dst[short(r['Kernel_Name'])] += float(r['Counter_Value'])
    tot = sum(sum(v) for v in tr.values())
    rows = []
    for k, v in tr.items():
        t = sum(v)
        if k.startswith('Cijk') or t / tot < 0.01:
            continue
        b = fe[k] * 1024 * 2 + wr[k] * 1024
        rows.append((t / tot, k, len(v), t / len(v) / 1e3, b / len(v) / 1e6, b / t))
    print('share of GPU time | kernel | launches | avg us | HBM MB per launch | GB/s | fraction of 8 TB/s')
    for r in sorted(rows, reverse=True):
        print('%5.1f%%  %-48s n=%4d avg %7.1f us  %7.1f MB  %5.0f GB/s (%.2f)' % (r[0] * 100, r[1][:48], r[2], r[3], r[4], r[5], r[5] / 8000))


if __name__ == '__main__':
    main()
