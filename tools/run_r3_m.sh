cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do GFDN_ADAM_ON_SIDE=$v timeout 300 python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adam_on_side=$v', d['ms_per_step'])"; done
timeout 600 python -m pytest tests/test_gpu_bank.py tests/test_gpu_fullsize.py -q -x -k "bank or bench_shape" 2>&1 | tail -3
