# usage (one gpurun call):  BARGS="--config directional" bash tools/run_pmc_bench.sh > gpurun_out/pmc.txt
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/pmcb
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmcb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 20 --repeats 1 --warmup 5 $BARGS > $OUT/pmcb.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmcb/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
rows=[]
for k, d in acc.items():
    m={c: sum(v[len(v)//2:])/max(1,len(v[len(v)//2:])) for c,v in d.items()}
    if m.get('SQ_WAVES',0)==0: continue
    w=m['SQ_WAVES']; 
    rows.append((m['SQ_WAVE_CYCLES'], k, w, m['SQ_INSTS_VALU']/w, m['SQ_INSTS_SALU']/w, m['SQ_INSTS_VMEM_RD']/w, m['SQ_WAVE_CYCLES']/w*4/2400, m['SQ_ACTIVE_INST_VALU']/m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'], len(next(iter(d.values())))))
rows.sort(reverse=True)
print("%-44s %7s %8s %8s %7s %9s %9s %8s %5s" % ("kernel","waves","valu/w","salu/w","vmem/w","life us","valu/wcyc","waitinst","n"))
for r in rows[:40]:
    print("%-44s %7d %8.0f %8.0f %7.0f %9.1f %9.3f %8.3f %5d" % r[1:])
PY
rm -rf $OUT/pmcb
