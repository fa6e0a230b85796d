"""Rebuild profiles/ from the raw rocprofv3 output of tools/run_measurements.sh (gpurun_out/<tag>_*).
usage: python tools/make_profiles.py [round_tag] [--pre]   (default r06)"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
# --pre: the tables bench.py reads at run time only (kernel stats, PMC bytes, timeline, in-step durations) -- run on the
# GPU box between the profiled runs and the final bench run, so that the bench line is computed from the same call's tables
PRE = '--pre' in sys.argv[2:]
src = os.path.join(ROOT, 'gpurun_out')
dst = os.path.join(ROOT, 'profiles')

def one(pattern):
    # gpurun merges into gpurun_out/ without clearing it: take the NEWEST match
    return max(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)

# 1. bench line
if not PRE:
    line = [l for l in open(os.path.join(src, f'{tag}_bench_n1.json')) if l.startswith('{')][-1]
    bench = json.loads(line)
    open(os.path.join(dst, f'{tag}_bench_n1.json'), 'w').write(line)
# 2. kernel stats
stats = one(f'{tag}_stats/**/*kernel_stats.csv')
shutil.copy(stats, os.path.join(dst, f'{tag}_bench_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
total_ms = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
# 3. PMC summary: per (kernel, grid) average FETCH_SIZE / WRITE_SIZE (KB) over launches
pmc = collections.defaultdict(dict)
for kind, ctr in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f'{tag}_pmc_{kind}/**/*counter_collection.csv'))):
        if r['Counter_Name'] == ctr and r['Kernel_Name'].startswith('k_') or r['Kernel_Name'].startswith('void k_'):
            acc[(r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k, v in acc.items():
        pmc[k][ctr] = sum(v) / len(v)
        pmc[k]['launches'] = len(v)
with open(os.path.join(dst, f'{tag}_pmc_hbm_bytes.csv'), 'w', newline='') as f:
    # (csv.writer: template kernels carry commas in their names -- an unquoted `k_em_bwd<9, 12>` shifted every column)
    wcsv = csv.writer(f)
    wcsv.writerow(['kernel', 'grid_size', 'launches', 'FETCH_SIZE_KB_raw', 'WRITE_SIZE_KB', 'read_MB_x2_corrected', 'write_MB',
                   'hbm_traffic_MB'])
    for (k, g), d in sorted(pmc.items()):
        fe, wr = d.get('FETCH_SIZE', 0.0), d.get('WRITE_SIZE', 0.0)
        wcsv.writerow([k, g, d['launches'], f'{fe:.1f}', f'{wr:.1f}', f'{2*fe*1024/1e6:.2f}', f'{wr*1024/1e6:.2f}',
                       f'{(2*fe+wr)*1024/1e6:.2f}'])
# 4. one replayed step as a timeline
tl = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'timeline.py'), os.path.join(src, f'{tag}_stats'), '100'],
                    capture_output=True, text=True).stdout
open(os.path.join(dst, f'{tag}_graph_step_timeline.txt'), 'w').write(
    "one HIP-graph replay of the training step (steady state): start us, end us, duration, HW queue, kernel\n" + tl)
# 4b. per-kernel durations INSIDE the replayed steps only (the timed region of the bench): the kernel trace between the
# Adam launches of 100 consecutive steady-state replays -- no host-launched roofline steps, no isolated launches, no
# one-off dataset kernels
trace = one(f'{tag}_stats/**/*kernel_trace.csv')
trows = list(csv.DictReader(open(trace)))
trows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(trows) if r['Kernel_Name'].startswith('k_adam(')]          # (one launch per step)
lo, hi = adam[60], adam[160]
nsteps = 100
agg = collections.defaultdict(lambda: [0, 0.0])
for r in trows[lo + 1:hi + 1]:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
with open(os.path.join(dst, f'{tag}_step_kernel_durations.csv'), 'w', newline='') as f:
    wcsv = csv.writer(f)
    wcsv.writerow(['kernel', 'launches_per_step', 'avg_us_in_step', 'us_per_step'])
    for k, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        wcsv.writerow([k, f'{cnt / nsteps:.2f}', f'{us / cnt:.2f}', f'{us / nsteps:.2f}'])
step_span = (int(trows[hi]['End_Timestamp']) - int(trows[lo]['End_Timestamp'])) / 1e3 / nsteps
# 4c. HBM traffic of ONE replayed step: launches per step of every kernel (100 replayed steps of the stats run) x the PMC
# traffic per launch of the grid it is launched with INSIDE those steps (the same kernels also run on other grids in the
# one-off dataset precompute) -- what bench.py reports as step_traffic_bytes
step_grid, step_count = {}, collections.Counter()
for r in trows[lo + 1:hi + 1]:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    step_grid.setdefault(k, collections.Counter())[g] += 1
    step_count[k] += 1
per_kernel, tot_step = [], 0.0
for k, grids in step_grid.items():
    g = grids.most_common(1)[0][0]
    d = pmc.get((k, g))
    if d is None:                      # (a grid the eager PMC runs did not see: the kernel's most frequent grid there)
        cands = [(v['launches'], v) for (kk, gg), v in pmc.items() if kk == k]
        d = max(cands, key=lambda c: c[0])[1] if cands else None
    if d is None:
        continue
    mb = (2 * d.get('FETCH_SIZE', 0.0) + d.get('WRITE_SIZE', 0.0)) * 1024 / 1e6
    per_kernel.append({'kernel': k, 'launches_per_step': step_count[k] / nsteps, 'MB_per_launch': round(mb, 3),
                       'MB_per_step': round(mb * step_count[k] / nsteps, 3)})
    tot_step += mb * step_count[k] / nsteps
per_kernel.sort(key=lambda e: -e['MB_per_step'])
json.dump({'bytes_per_step': int(tot_step * 1e6), 'source': f'profiles/{tag}_step_traffic.json = sum over the launches of one '
           f'replayed step of 2 x FETCH_SIZE + WRITE_SIZE per launch (profiles/{tag}_pmc_hbm_bytes.csv, separate --pmc passes) '
           f'x launches per step (100 replayed steps of the stats run)', 'per_kernel': per_kernel},
          open(os.path.join(dst, f'{tag}_step_traffic.json'), 'w'), indent=1)
if PRE:
    sys.exit(0)
# 5. README
dom = bench.get('roofline', {})
name = dom.get('kernel', '')
drow = next((r for r in rows if r['Name'].startswith(name)), None)
with open(os.path.join(dst, 'README.md'), 'w') as f:
    f.write(f"""# profiles/ -- round {tag[1:]}
All `{tag}_*` files come from ONE `gpurun` call (`bash tools/run_measurements.sh {tag}`) on one MI355X (gfx950, ROCm 7.2);
`bench.py` is the command the driver runs (N = 1, workload = the 7-band configuration BASELINE.json's metric is quoted
on).  Rebuilt by `python tools/make_profiles.py {tag}`.  The `r01_*` ... `r05_*` files are the previous rounds', kept for comparison.

| file | command | what it holds |
|---|---|---|
| `{tag}_bench_n1.json` | `python bench.py` | the bench JSON line (graph-replay steps, roofline leg, CPU baseline with `loss_delta_vs_cpu`) |
| `{tag}_bench_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline` | per-kernel totals / averages (includes the one-off dataset front end, the 20 host-launched roofline steps and the 60 isolated launches of the roofline kernel) |
| `{tag}_pmc_hbm_bytes.csv` | `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --eager`, and the same with `--pmc WRITE_SIZE` (separate passes) | average FETCH_SIZE / WRITE_SIZE per launch of every hand-written kernel, by grid size; read bytes corrected x2 for gfx950 as MI355X_MICROARCH.md prescribes |
| `{tag}_graph_step_timeline.txt` | from the kernel trace of the stats run | every kernel of one replayed step with start/end and hardware queue |
| `{tag}_step_kernel_durations.csv` | from the kernel trace of the stats run | per kernel: launches per step, average duration and microseconds per step over 100 consecutive REPLAYED steps only (what `roofline.top` of the bench line is built from) |
| `{tag}_directional_bench.json`, `{tag}_directional_kernels.txt`, `{tag}_directional_pmc_hbm_bytes.csv` | `bash tools/run_dir_measurements.sh {tag}` (its own gpurun call) + `python tools/make_dir_profiles.py {tag}` | BASELINE.json configs[3]: the bench line of `python bench.py --config directional` (roofline block of `k_em_bwd` with PMC traffic, CPU baseline with loss and gradient deviations), kernel totals of the graph-replayed band-steps, PMC bytes per launch of its kernels |
| `{tag}_n32_kernels.txt`, `{tag}_n32_timeline.txt` | `bash tools/run_aux.sh {tag}` (`tools/run_n32_profile.sh` + `tools/timeline.py`; its own gpurun call) | kernel totals and one replayed step of the 7-band step at N = 32 (configs[4]) |
| `{tag}_recipe_bench.json`, `{tag}_recipe_timeline.txt` | `bash tools/run_aux.sh {tag}` | `bench.py --recipe reference`: the sub-band driver's own configuration (8 bands, N = 12, per-band gain networks) in one bank: bench line and one replayed step |
| `{tag}_step_traffic.json` | this script | the PMC traffic of ALL launches of one replayed step (`step_traffic_bytes` of the bench line) |
| `r06_edc_band_experiment.txt` | several gpurun calls of round 6 (`tools/run_ab.sh`, `run_probe.sh`, wall-clock stamps from a probe build) | measured negatives of round 6: the EDC term with the band's receivers inside the workgroup, the gain network on a lane of its own, the side stream's tail reordered (DESIGN.md section 4.3) |
| `r06_parity_margins.txt` | one full `pytest -m gpu` run; `tests/margins.py` | how far inside its bound every tolerance check of the GPU suite lands (worst value per test and check): the evidence behind the tightened gradient bounds (DESIGN.md section 2) |
| `r06_stamps_edc.txt`, `r06_stamps_edr.txt`, `r06_sq_bench.txt` | `tools/mid_stamps.py` on probe builds (`-DE1_TIMING`, `-DEDW_TIMING`), `tools/run_pmc_bench.sh` | the two launches of the middle from the inside: wall-clock stamps per phase of `k_edc_lin_one` (medians over the 224 workgroups) and per loop section of `k_edr_lin_wave`; SQ counters of every kernel of the bench run (vector / scalar / memory instructions per wave, wave lifetime, share of the waves' cycles with a vector instruction in flight) -- DESIGN.md section 4.0.8 |
| `r05_ab_same_box.txt`, `r05_grad_stage_probe.txt`, `r05_n32_kernels_stage1.txt` | round 5 | same-box A/B of round 5's switches; where the float32 deviation of dL/dM enters, stage by stage (DESIGN.md section 2); the N = 32 step before its polynomial passes became transforms |
| `r02_mfma_experiment.json` (round 2; not repeated since: the kernels it times did not change) | `python tools/mfma_experiment.py` (its own gpurun call) | configs[4]'s bf16 / f32 MFMA contraction against the solve path: time and deviation of H |
| `r04_graph_step_timeline_linear_v1.txt`, `_spectral_v1.txt` | as `r04_graph_step_timeline.txt`, earlier in round 4 | the replayed step after the time-domain output stage (0.527 ms) and after the EDR loss on composed spectra (0.473 ms) |

`bench.py` reads `{tag}_pmc_hbm_bytes.csv` (`roofline.traffic`), `{tag}_step_kernel_durations.csv` (`roofline.top`: in-step
durations) and `{tag}_bench_kernel_stats.csv` (whole-run averages beside them) at run time, so every fraction in the bench
line can be recomputed from this directory.  Replayed step period in the stats run: {step_span:.1f} us.

## bench line
`{bench['value']:.0f} {bench['unit']}` = {bench['ms_per_step']:.4f} ms/step on 1 GPU;
cpu_baseline {bench.get('cpu_baseline', {}).get('value', float('nan')):.1f} {bench['unit']} ({bench.get('cpu_baseline', {}).get('cores')} threads, kind {bench.get('cpu_baseline', {}).get('kind')}).

## Roofline kernel: `{name}` (the longer of the pair `k_edr_lin_wave` / `k_edc_lin_one` in the replayed step)
bench.py, in the step (HIP events around every launch during 20 host-launched steps of the timed launch sequence, raw
bracket): **{dom.get('avg_launch_us', float('nan')):.1f} us**; alone on the chip: {dom.get('isolated_us', float('nan')):.1f} us;
rocprofv3 average over {drow['Calls'] if drow else '?'} launches of the stats run: **{float(drow['AverageNs'])/1e3 if drow else float('nan'):.1f} us**
(that average covers the replayed steps, the host-launched steps, the isolated launches and the 64-receiver launches of the
one-off target precompute).
Algorithmic bytes per launch {dom.get('alg_bytes_per_launch', 0)/1e6:.2f} MB -> achieved {dom.get('achieved', 0):.0f} GB/s = {dom.get('frac', 0):.3f} of {dom.get('peak')} GB/s (in the step).

## Top kernels by total time (stats run)

| kernel | calls | total ms | avg us | % |
|---|---|---|---|---|
""")
    for r in rows[:30]:
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    f.write(f"\nTotal GPU time in the trace: {total_ms:.1f} ms.\n")
    # HBM traffic of one step: the kernels of the replayed step (timeline) x the PMC traffic of their launches (per
    # kernel the grid with most launches = the step's)
    f.write("\n## HBM traffic per step (PMC, 2 x FETCH_SIZE + WRITE_SIZE)\n\n| kernel | launches per step | MB per launch | MB per step |\n|---|---|---|---|\n")
    # the grid a kernel is launched with INSIDE the replayed steps (the stats run's trace): the PMC row of that grid is
    # the step's -- the same kernels also run on other grids in the one-off dataset precompute
    step_grid = {}
    for r in trows[lo + 1:hi + 1]:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
        step_grid.setdefault(k, collections.Counter())[g] += 1
    best = {}
    for (k, g), d in pmc.items():
        want = step_grid.get(k)
        if want is not None:
            if g == want.most_common(1)[0][0]:
                best[k] = d
        elif k not in best or d['launches'] > best[k]['launches']:
            best[k] = d
    counts = collections.Counter()
    for line in tl.splitlines():
        parts = line.split()
        if len(parts) >= 6 and parts[2].startswith('d='):
            nm = line.split('q=')[1].split(None, 1)[1].split('(')[0].replace('void ', '').strip()
            counts[nm] += 1
    tot = 0.0
    rowsout = []
    for nm, c in counts.items():
        d = best.get(nm)
        if d is None:
            continue
        mb = (2 * d.get('FETCH_SIZE', 0.0) + d.get('WRITE_SIZE', 0.0)) * 1024 / 1e6
        rowsout.append((mb * c, nm, c, mb))
        tot += mb * c
    for t_, nm, c, mb in sorted(rowsout, reverse=True):
        f.write(f"| `{nm}` | {c} | {mb:.1f} | {t_:.1f} |\n")
    f.write(f"\nSum over the step's launches: **{tot / 1e3:.2f} GB** (SURVEY section 8d bytes: {224 * 2811048 / 1e9:.2f} GB; what the linear "
            f"step has to move: {224 * 1165696 / 1e9:.2f} GB; round 5: 0.76 GB; round 4: 0.97 GB; round 1: 2.47 GB).  `{tag}_step_traffic.json` holds the same "
            f"sum built from 100 replayed steps' launch counts: {tot_step / 1e3:.2f} GB.\n")
print(open(os.path.join(dst, 'README.md')).read()[-3500:])
