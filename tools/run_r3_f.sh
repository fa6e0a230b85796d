set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bank.py -x -q -k "eight_line or fused_bank_step_equals" 2>&1 | tail -15 && \
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "n32" 2>&1 | grep -E "passed|failed|deviation|Error" | tail -5 && \
python bench.py --lines-per-group 8 --no-cpu-baseline --steps 200 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=32 ms_per_step', d['ms_per_step'], d['config']['final_loss'][:3])" && \
bash tools/run_n32_profile.sh 2>&1 | tail -30
