# round 4: EDR on composed spectra + EDC on the fly (csrc/edrlin.hip): kernel tests, full-size tests, bench A/B, timeline
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "spectrum or composed or on_the_fly or linear_output or transform_order or one_sweep" > $OUT/r04_spec_kernels.log 2>&1; tail -3 $OUT/r04_spec_kernels.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "composed or time_domain or bank_graph or distinct or bench_shape" > $OUT/r04_spec_full.log 2>&1; grep -a "deviation\|passed\|failed\|Error" $OUT/r04_spec_full.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_spec.json 2> $OUT/r04_bench_spec.err; tail -c 300 $OUT/r04_bench_spec.err; python -c "import json;d=json.load(open('$OUT/r04_bench_spec.json'));print('SPEC', d['ms_per_step'],d['value'])"
GFDN_SPECTRAL_EDR=0 timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_nospec.json 2> $OUT/r04_bench_nospec.err; python -c "import json;d=json.load(open('$OUT/r04_bench_nospec.json'));print('stored signals', d['ms_per_step'],d['value'])"
rm -rf $OUT/r04_trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/r04_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/r04_trace 100 > $OUT/r04_timeline3.txt 2>&1; cat $OUT/r04_timeline3.txt
find $OUT/r04_trace -name "*kernel_trace.csv" -size +20M -delete
