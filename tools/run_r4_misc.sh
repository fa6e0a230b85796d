set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "f16 or directional_graphed or f11 or f8 or f15 or svf" > $OUT/r04_misc.log 2>&1; tail -12 $OUT/r04_misc.log
