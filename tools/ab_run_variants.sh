# usage (on the GPU box): bash tools/ab_run_variants.sh  -- runs bench.py once per library variant built by
# tools/ab_build_variants.sh and restores the in-tree library; edit the list in the for loop
cd $GRAFT_REPO_ROOT
cp diffgfdn_amd/lib/libdiffgfdn_hip.so /tmp/lib_keep.so
for v in base colfadj2 rpb16 base; do
  cp tools/_probe/libv_$v.so diffgfdn_amd/lib/libdiffgfdn_hip.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1000 > gpurun_out/g.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open('gpurun_out/g.json'))
print(sys.argv[1], d['value'], d['ms_per_step'])" "$v"
done
cp /tmp/lib_keep.so diffgfdn_amd/lib/libdiffgfdn_hip.so
