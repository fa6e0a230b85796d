# round 4: combine folded into the STFT load, edge fold by a wave, gamma in transform order: tests, bench, timeline
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear_output or folded_in or transform_order or irfft or blu or pairs" > $OUT/r04_lin2_kernels.log 2>&1; tail -3 $OUT/r04_lin2_kernels.log
timeout 600 python -m pytest tests/test_gpu_bank.py -q -x > $OUT/r04_lin2_bank.log 2>&1; tail -3 $OUT/r04_lin2_bank.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "not directional" > $OUT/r04_lin2_full.log 2>&1; grep -a "deviation\|passed\|failed" $OUT/r04_lin2_full.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_lin2.json 2> $OUT/r04_bench_lin2.err; tail -c 300 $OUT/r04_bench_lin2.err; python -c "import json;d=json.load(open('$OUT/r04_bench_lin2.json'));print('LINEAR2', d['ms_per_step'],d['value'])"
GFDN_COMBINE_IN_STFT=0 timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_lin2b.json 2> $OUT/r04_bench_lin2b.err; python -c "import json;d=json.load(open('$OUT/r04_bench_lin2b.json'));print('separate combine', d['ms_per_step'],d['value'])"
rm -rf $OUT/r04_trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/r04_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/r04_trace 100 > $OUT/r04_timeline2.txt 2>&1; cat $OUT/r04_timeline2.txt
find $OUT/r04_trace -name "*kernel_trace.csv" -size +20M -delete
