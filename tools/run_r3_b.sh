set -x
cd $GRAFT_REPO_ROOT
python tools/decay_probe.py && python tools/decay_probe.py 64 && \
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3_probe && \
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_probe -- python $GRAFT_REPO_ROOT/tools/decay_probe.py > /dev/null 2>&1 && \
python - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3_probe/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
