# kernel totals of the graph-replayed N = 32 bank step: bash tools/run_n32_profile.sh
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/n32_stats
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/n32_stats -- python $GRAFT_REPO_ROOT/bench.py --lines-per-group 8 --no-cpu-baseline --steps 60 > $GRAFT_REPO_ROOT/gpurun_out/n32_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob, collections, os
f = max(glob.glob('gpurun_out/n32_stats/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adam(')]
sel = rows[ad[20]:ad[50]]
n = 30
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r['Kernel_Name'][:72]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"span/step {span/n:.0f} us, kernel time/step {tot/n:.0f} us, kernels/step {len(sel)/n:.1f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
    print(f"{v[1]/tot*100:5.1f}% {v[1]/v[0]:8.1f} us x{v[0]/n:5.1f}  {k}")
PY
