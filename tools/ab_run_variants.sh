# usage (on the GPU box): bash tools/ab_run_variants.sh [rounds] -- runs bench.py once per library variant built by
# tools/ab_build_variants.sh (all tools/_probe/libv_*.so), `rounds` times round-robin, and restores the in-tree library
cd $GRAFT_REPO_ROOT
cp diffgfdn_amd/lib/libdiffgfdn_hip.so /tmp/lib_keep.so
R=${1:-2}
for i in $(seq $R); do
for f in tools/_probe/libv_*.so; do
  v=$(basename $f .so)
  cp $f diffgfdn_amd/lib/libdiffgfdn_hip.so
  timeout -k 10 200 python bench.py --no-cpu-baseline ${BENCH_ARGS:---steps 400} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'])" "$v"
done
done
cp /tmp/lib_keep.so diffgfdn_amd/lib/libdiffgfdn_hip.so
