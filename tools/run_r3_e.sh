set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bank.py -x -q -k "eight_line" 2>&1 | grep -E "^E  .*assert|passed|failed" | cut -c1-400 | head -12 && \
python tools/tf8_probe.py 2>&1 | tail -7 && \
python bench.py --lines-per-group 8 --no-cpu-baseline --steps 200 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=32 ms_per_step', d['ms_per_step'], d['config']['final_loss'][:3])"
