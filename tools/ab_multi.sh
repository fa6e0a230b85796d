cd $GRAFT_REPO_ROOT
for i in 1 2; do for m in 1 2 4; do
GFDN_MULTI=$m timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('multi $m', d['ms_per_step']/$m)"
done; done
