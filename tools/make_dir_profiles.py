"""profiles/<tag>_directional_* from the raw rocprofv3 output of tools/run_dir_measurements.sh (gpurun_out/<tag>_dir_*).
usage: python tools/make_dir_profiles.py [round_tag]   (default r06)"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src, dst = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)


line = [l for l in open(os.path.join(src, f'{tag}_directional_bench.json')) if l.startswith('{')][-1]
open(os.path.join(dst, f'{tag}_directional_bench.json'), 'w').write(line)
bench = json.loads(line)
# kernel table of the replayed band-steps: the last 40 % of the trace
rows = list(csv.DictReader(open(one(f'{tag}_dir_stats/**/*kernel_trace.csv'))))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adam')]
lo, hi = ad[len(ad) // 2], ad[-1]
nsteps = len([i for i in ad if lo < i <= hi])
sel = rows[lo + 1:hi + 1]
span = (int(sel[-1]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r['Kernel_Name'][:72]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
with open(os.path.join(dst, f'{tag}_directional_kernels.txt'), 'w') as f:
    f.write(f"# rocprofv3 --kernel-trace of: python bench.py --config directional --no-cpu-baseline --steps 50 (tools/run_dir_measurements.sh)\n")
    f.write(f"# {nsteps} graph-replayed band-steps (3 groups x 9 SH channels, 12 directions, 32 receivers); bench line of the same call: "
            f"{bench['config']['ms_per_band_step']:.3f} ms per band-step\n")
    f.write("# columns: share of kernel time, average duration, launches per band-step, kernel\n")
    f.write(f"span/band-step {span / nsteps:.0f} us (under the profiler), kernel time/band-step {tot / nsteps:.0f} us, "
            f"kernels/band-step {len(sel) / nsteps:.1f}\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
        f.write(f"{v[1] / tot * 100:5.1f}% {v[1] / v[0]:8.1f} us x{v[0] / nsteps:5.1f}  {k}\n")
# PMC summary, same columns as <tag>_pmc_hbm_bytes.csv
pmc = collections.defaultdict(dict)
for kind, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f'{tag}_dir_pmc_{kind}/**/*counter_collection.csv'))):
        if r['Counter_Name'] == ctr and (r['Kernel_Name'].startswith('k_') or r['Kernel_Name'].startswith('void k_')):
            acc[(r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k, v in acc.items():
        pmc[k][ctr] = sum(v) / len(v)
        pmc[k]['launches'] = len(v)
with open(os.path.join(dst, f'{tag}_directional_pmc_hbm_bytes.csv'), 'w', newline='') as f:
    # (csv.writer quotes kernel names that carry commas -- `k_em_bwd<9, 12>` unquoted shifted every column, and the bench line
    # read the write bytes as the traffic)
    wcsv = csv.writer(f)
    wcsv.writerow(['kernel', 'grid_size', 'launches', 'FETCH_SIZE_KB_raw', 'WRITE_SIZE_KB', 'read_MB_x2_corrected', 'write_MB',
                   'hbm_traffic_MB'])
    for (k, g), d in sorted(pmc.items()):
        fe, wr = d.get('FETCH_SIZE', 0.0), d.get('WRITE_SIZE', 0.0)
        wcsv.writerow([k, g, d['launches'], f'{fe:.1f}', f'{wr:.1f}', f'{2*fe*1024/1e6:.2f}', f'{wr*1024/1e6:.2f}',
                       f'{(2*fe+wr)*1024/1e6:.2f}'])
print(open(os.path.join(dst, f'{tag}_directional_kernels.txt')).read())
