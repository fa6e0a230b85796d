# round 4: tiled cell order of the spectral planes: kernel tests, full-size equivalence, bench A/B, timeline
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "spectrum or composed" > $OUT/r04_tiled_kernels.log 2>&1; tail -3 $OUT/r04_tiled_kernels.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "composed or bank_graph" > $OUT/r04_tiled_full.log 2>&1; tail -3 $OUT/r04_tiled_full.log
timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/r04_bench_tiled.json 2> $OUT/r04_bench_tiled.err; tail -c 300 $OUT/r04_bench_tiled.err; python -c "import json;d=json.load(open('$OUT/r04_bench_tiled.json'));print('TILED', d['ms_per_step'],d['value'], d['roofline']['avg_launch_us'], d['roofline']['isolated_us'])"
GFDN_TILED_SPECTRA=0 timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/r04_bench_plain.json 2> $OUT/r04_bench_plain.err; python -c "import json;d=json.load(open('$OUT/r04_bench_plain.json'));print('PLAIN', d['ms_per_step'],d['value'], d['roofline']['avg_launch_us'], d['roofline']['isolated_us'])"
rm -rf $OUT/r04_trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/r04_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $OUT/r04_trace 100 > $OUT/r04_timeline4.txt 2>&1; cat $OUT/r04_timeline4.txt
find $OUT/r04_trace -name "*kernel_trace.csv" -size +20M -delete
