cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in 0 2 4 8; do
timeout 300 python bench.py --no-cpu-baseline --steps 400 --pipe-steps $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipe $v', d['ms_per_step'])"
done; done
