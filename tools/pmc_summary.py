#!/usr/bin/env python
"""Summarise the rocprofv3 passes of `tools/gpu.sh <tag> bench stats pmc` into the files kept under profiles/.

  python tools/pmc_summary.py gpurun_out/<tag> profiles/r05 [pmc_traffic.json | -]

Reads  <in>/stats_kernel_stats.csv (or <in>/stats/r1_kernel_stats.csv), <in>/pmc_fetch/r1_counter_collection.csv, <in>/pmc_write/...
Writes <out>/kernel_stats.csv (copy), <out>/pmc_hbm.csv (per kernel: launches, FETCH_SIZE, WRITE_SIZE, corrected bytes)
and profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

With <in>/pmc_mfma/r1_counter_collection.csv present (round 2: SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE in one pass) also writes <out>/pmc_mfma.csv: per kernel the average duration, the shader clock during the launch
(GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GUI / 8 / duration) and the MFMA utilisation
SQ_VALU_MFMA_BUSY_CYCLES / (GUI / 8 x 1024 SIMDs); SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x issued v_mfma_f32_32x32x16_bf16.

Corrections (MI355X_MICROARCH.md, HBM section): rocprofv3's FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so reads are doubled.  WRITE_SIZE is
uncalibrated and taken as reported.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys


def per_kernel(path):
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            e = agg.setdefault(r['Kernel_Name'], [0, 0.0])
            e[0] += 1
            e[1] += float(r['Counter_Value'])
    return agg


def short(name):
    m = re.search(r'conv3d_igemm_kernel<(\d+), (\d+), (\d+), (\d+)((?:, \d+)*)>', name)
    if m:       # <dtype, channels per block, positions per block, waves over channels, taps per step, weights direct, unrolled taps>
        rest = [int(x) for x in m.group(5).replace(',', ' ').split()]
        tps = ',tps%d' % rest[0] if rest and rest[0] != 1 else ''
        return 'conv3d_igemm_kernel<%s,%s,%s%s>' % ('bf16' if m.group(1) == '1' else 'fp32', m.group(2), m.group(3), tps)
    m = re.search(r'(conv3x3_c64_ws_kernel|conv3x3_bt_kernel|conv1x1_k64_c256_ws_kernel|conv1x1_lw_kernel|stem_pool_kernel|stem_conv_kernel)', name)
    if m:
        return m.group(1) + ('<bf16,256,256>' if m.group(1) == 'conv3x3_bt_kernel' else '<bf16>')
    m = re.search(r'([A-Za-z_0-9]+)(<[^(]*>)?\(', name)
    return m.group(1) if m else name


def mfma_summary(src, dst):
    path = os.path.join(src, 'pmc_mfma', 'r1_counter_collection.csv')
    if not os.path.exists(path):
        return
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            e = agg.setdefault(short(r['Kernel_Name']), collections.defaultdict(float))
            e[r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                e['n'] += 1
                e['ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    with open(os.path.join(dst, 'pmc_mfma.csv'), 'w') as f:
        f.write('kernel,launches,avg_duration_us,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,SQ_INSTS_VALU_MFMA_MOPS_BF16_per_launch,'
                'SQ_BUSY_CYCLES_per_launch,GRBM_GUI_ACTIVE_per_launch,shader_clock_mhz,mfma_utilisation,total_ms\n')
        for k, e in sorted(agg.items(), key=lambda kv: -kv[1]['ns']):
            n = max(e['n'], 1)
            cyc = e['GRBM_GUI_ACTIVE'] / 8.0          # summed over the 8 XCDs
            util = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0) if cyc > 0 else 0.0
            mhz = cyc / (e['ns'] * 1e-3) if e['ns'] > 0 else 0.0
            f.write('%s,%d,%.1f,%.4g,%.4g,%.4g,%.4g,%.0f,%.3f,%.3f\n' % (k, n, e['ns'] / n / 1e3, e['SQ_VALU_MFMA_BUSY_CYCLES'] / n,
                                                                        e['SQ_INSTS_VALU_MFMA_MOPS_BF16'] / n, e['SQ_BUSY_CYCLES'] / n,
                                                                        e['GRBM_GUI_ACTIVE'] / n, mhz, util, e['ns'] / 1e6))
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1]['ns'])[:6]:
        cyc = e['GRBM_GUI_ACTIVE'] / 8.0
        print('%-44s n=%4d MFMA utilisation %.3f at %.0f MHz' % (k, e['n'], e['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0) if cyc else 0,
                                                               cyc / (e['ns'] * 1e-3) if e['ns'] else 0))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    mfma_summary(src, dst)
    for cand in (os.path.join(src, 'stats_kernel_stats.csv'), os.path.join(src, 'stats', 'r1_kernel_stats.csv')):
        if os.path.exists(cand):
            shutil.copy(cand, os.path.join(dst, 'kernel_stats_pipeline1.csv'))
            break
    if not os.path.exists(os.path.join(src, 'pmc_fetch', 'r1_counter_collection.csv')):
        return          # (an MFMA-only pass, e.g. of a training workload)
    def by_short(agg):      # template variants of one tile shape (table-driven / unrolled-tap loops) are one bench bucket
        out = collections.OrderedDict()
        for name, (n, v) in agg.items():
            e = out.setdefault(short(name), [0, 0.0])
            e[0] += n
            e[1] += v
        return out
    fetch = by_short(per_kernel(os.path.join(src, 'pmc_fetch', 'r1_counter_collection.csv')))
    write = by_short(per_kernel(os.path.join(src, 'pmc_write', 'r1_counter_collection.csv')))
    kernels = {}
    rows = []
    for name, (n, fk) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
        wn, wk = write.get(name, (0, 0.0))
        rd = 2.0 * fk * 1024.0 / n
        wr = wk * 1024.0 / wn if wn else 0.0
        rows.append((short(name), n, fk / n, wk / wn if wn else 0.0, rd, wr, rd + wr))
        kernels[short(name)] = {'launches_profiled': n, 'fetch_kib_per_launch_raw': round(fk / n, 2),
                                'write_kib_per_launch_raw': round(wk / wn, 2) if wn else None,
                                'hbm_bytes_per_launch': round(rd + wr)}
    with open(os.path.join(dst, 'pmc_hbm.csv'), 'w') as f:
        f.write('kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,read_bytes_corrected_x2,'
                'write_bytes,hbm_bytes_per_launch\n')
        for r in rows:
            f.write('%s,%d,%.2f,%.2f,%.0f,%.0f,%.0f\n' % r)
    # (a third argument names another file -- or '-' for none -- when the passes are of a workload other than the bench default,
    #  whose record bench.py reads as profiles/pmc_traffic.json)
    out_json = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(dst.rstrip('/')), 'pmc_traffic.json')
    if out_json != '-':
        with open(os.path.join(src, 'bench.json')) as f:
            bench = json.loads(f.read().strip().splitlines()[-1])
        wl = bench['config']['workload']
        m = re.search(r'R-(\d+) .* 1x3x(\d+)x(\d+)x(\d+)', wl)
        batch = int(bench['config'].get('images_per_forward', 1))
        rec = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/gpu.sh pmc), python bench.py '
                         '--steps 3 --warmup 1 --pipeline 1 --no-roofline; reads x2 (gfx950 FETCH_SIZE correction), KiB -> bytes',
               'workload': {'arch': m.group(1), 'frames': int(m.group(2)), 'height': int(m.group(3)),
                            'width': int(m.group(4)), 'dtype': bench['dtype'],
                            'keyframe_dce': bench['config'].get('keyframe_dce', False), 'batch': batch},
               'kernels': kernels}
        with open(out_json, 'w') as f:
            json.dump(rec, f, indent=1, sort_keys=True)
    for r in rows[:8]:
        print('%-40s n=%4d read %8.1f MB write %8.1f MB' % (r[0], r[1], r[4] / 1e6, r[5] / 1e6))


if __name__ == '__main__':
    main()
