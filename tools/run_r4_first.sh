# round 4, first GPU call: the new per-band-window tests, the whole GPU suite, the default bench and --distinct-t60
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bank.py -q -x -k "banded or distinct" > $OUT/r04_new_tests.log 2>&1; tail -5 $OUT/r04_new_tests.log
timeout 1100 python -m pytest tests -q -m gpu -x > $OUT/r04_suite.log 2>&1; tail -5 $OUT/r04_suite.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/r04_bench_a.json 2> $OUT/r04_bench_a.err; tail -c 400 $OUT/r04_bench_a.err; python -c "import json;d=json.load(open('$OUT/r04_bench_a.json'));print(d['ms_per_step'],d['value'])"
timeout 300 python bench.py --no-cpu-baseline --distinct-t60 > $OUT/r04_bench_distinct.json 2> $OUT/r04_bench_distinct.err; tail -c 400 $OUT/r04_bench_distinct.err; python -c "import json;d=json.load(open('$OUT/r04_bench_distinct.json'));print(d['ms_per_step'],d['value'],d['config']['band_edc_windows'])"
