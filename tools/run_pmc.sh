# SQ counters per kernel of a probe script:  bash tools/run_pmc.sh <tag> "<counters>" <script.py> [args]
set -x
TAG=$1; CTR=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_pmc
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/${TAG}_pmc -- python $GRAFT_REPO_ROOT/"$@" > $OUT/${TAG}_pmc.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/${TAG}_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if not k.startswith(('void k_', 'k_')): continue
    print(k, ' '.join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(d.items())), 'n=%d' % len(next(iter(d.values()))))
PY
rm -rf $OUT/${TAG}_pmc
