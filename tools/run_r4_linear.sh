# round 4: the time-domain output stage (csrc/linear.hip): kernel tests, bank tests, full-size tests, bench A/B, timeline
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear_output" > $OUT/r04_lin_kernels.log 2>&1; tail -3 $OUT/r04_lin_kernels.log
timeout 600 python -m pytest tests/test_gpu_bank.py -q -x > $OUT/r04_lin_bank.log 2>&1; tail -3 $OUT/r04_lin_bank.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "not directional" > $OUT/r04_lin_full.log 2>&1; tail -3 $OUT/r04_lin_full.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_lin.json 2> $OUT/r04_bench_lin.err; tail -c 300 $OUT/r04_bench_lin.err; python -c "import json;d=json.load(open('$OUT/r04_bench_lin.json'));print('LINEAR', d['ms_per_step'],d['value'])"
GFDN_LINEAR=0 timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_fold.json 2> $OUT/r04_bench_fold.err; python -c "import json;d=json.load(open('$OUT/r04_bench_fold.json'));print('FOLDED', d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --lines-per-group 8 > $OUT/r04_bench_lin_n32.json 2> $OUT/r04_bench_lin_n32.err; python -c "import json;d=json.load(open('$OUT/r04_bench_lin_n32.json'));print('LINEAR N32', d['ms_per_step'],d['value'])"
rm -rf $OUT/r04_trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/r04_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/r04_trace 100 > $OUT/r04_timeline.txt 2>&1; cat $OUT/r04_timeline.txt
find $OUT/r04_trace -name "*kernel_trace.csv" -size +20M -delete
