"""profiles/<round>_pmc_l2_valu.csv from the two extra counter passes of tools/run_pmc_extra.sh
(rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum / --pmc SQ_INSTS_VALU SQ_WAVES on `bench.py --eager`):
per hand-written kernel and grid size the average per launch.   python tools/make_pmc_extra.py r01"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r01'


def agg(pattern):
    path = max(glob.glob(os.path.join(ROOT, 'gpurun_out', pattern, '**', '*counter_collection.csv'), recursive=True),
               key=os.path.getmtime)
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[(r['Kernel_Name'].split('(')[0], int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


t, s = agg(rnd + '_pmc_tcc'), agg(rnd + '_pmc_sq')
out = os.path.join(ROOT, 'profiles', rnd + '_pmc_l2_valu.csv')
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'grid_size', 'launches', 'TCC_REQ_per_launch', 'TCC_HIT_pct', 'TCC_MISS_per_launch',
                'SQ_INSTS_VALU_per_wave', 'SQ_WAVES_per_launch'])
    for (name, grid) in sorted(t):
        if not name.startswith('k_'):
            continue
        c, v = t[(name, grid)], s.get((name, grid), {})
        n = len(c.get('TCC_REQ_sum', []))
        if not n:
            continue
        req, hit, miss = (sum(c[k]) / n for k in ('TCC_REQ_sum', 'TCC_HIT_sum', 'TCC_MISS_sum'))
        nv = max(len(v.get('SQ_WAVES', [])), 1)
        waves = sum(v.get('SQ_WAVES', [0])) / nv
        valu = sum(v.get('SQ_INSTS_VALU', [0])) / nv
        w.writerow([name, grid, n, round(req), round(100 * hit / max(req, 1), 1), round(miss),
                    round(valu / max(waves, 1)), round(waves)])
print(out)
