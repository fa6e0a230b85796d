# per-kernel averages of a probe script alone on the chip:  bash tools/run_probe.sh <tag> <script.py> [args]
set -x
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_probe
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_probe -- python $GRAFT_REPO_ROOT/"$@" > $OUT/${TAG}_probe.log 2>&1
cat $OUT/${TAG}_probe.log | tail -5
python - <<PY
import csv, glob
f = glob.glob("$OUT/${TAG}_probe/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg us {float(r['AverageNs'])/1e3:8.1f}")
PY
rm -rf $OUT/${TAG}_probe
