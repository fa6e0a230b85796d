set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "composed" > $OUT/r04_q_kernels.log 2>&1; tail -2 $OUT/r04_q_kernels.log
timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/r04_bench_q.json 2> $OUT/r04_bench_q.err; tail -c 300 $OUT/r04_bench_q.err; python -c "import json;d=json.load(open('$OUT/r04_bench_q.json'));print('BENCH', d['ms_per_step'],d['value'], d['roofline']['avg_launch_us'], d['roofline']['isolated_us'])"
