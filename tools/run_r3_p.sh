cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do GFDN_COLORLESS_LATE=$v timeout 300 python bench.py --lines-per-group 8 --no-cpu-baseline --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('late=$v', d['ms_per_step'])"; done
timeout 600 python -m pytest tests/test_gpu_bank.py tests/test_gpu_fullsize.py -q -x -k "eight or n32" 2>&1 | tail -2
