cd $GRAFT_REPO_ROOT
python bench.py --lines-per-group 8 --no-cpu-baseline --steps 300 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=32 ms_per_step', d['ms_per_step'])"
bash tools/run_n32_profile.sh 2>&1 | tail -14
