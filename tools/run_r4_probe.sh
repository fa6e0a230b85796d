# round 4: kernel test of the banded EDC entry points, the bench-shape oracle test at N = 16 and N = 32, the gradient-stage probe
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "banded" > $OUT/r04_banded.log 2>&1; tail -3 $OUT/r04_banded.log
timeout 300 python tools/grad_stage_probe.py > $OUT/r04_grad_stage_probe.txt 2>&1; tail -12 $OUT/r04_grad_stage_probe.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "bench_shape" > $OUT/r04_bench_shape.log 2>&1; grep -a "worst gradient\|passed\|failed\|Error" $OUT/r04_bench_shape.log | tail
