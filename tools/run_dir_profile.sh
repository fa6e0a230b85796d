# kernel totals of the graph-replayed directional step: bash tools/run_dir_profile.sh
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/dir_stats
timeout 300 python bench.py --config directional --no-cpu-baseline > gpurun_out/dir_bench.json 2> gpurun_out/dir_bench.err; tail -c 400 gpurun_out/dir_bench.json; echo
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dir_stats -- python $GRAFT_REPO_ROOT/bench.py --config directional --no-cpu-baseline --steps 50 > $GRAFT_REPO_ROOT/gpurun_out/dir_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/dir_stats/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 40 % of the trace = replayed steps
t0 = int(rows[int(len(rows) * 0.6)]['Start_Timestamp'])
sel = [r for r in rows if int(r['Start_Timestamp']) >= t0]
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = r['Kernel_Name'][:70]
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"span {span:.0f} us, kernel time {tot:.0f} us, kernels {len(sel)}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{v[1]/tot*100:5.1f}% {v[1]/v[0]:8.1f} us x{v[0]:5d}  {k}")
PY
