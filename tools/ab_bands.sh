cd $GRAFT_REPO_ROOT
for nb in 1 2 4 7; do for v in 0 1; do
GFDN_GG_SIDE=$v timeout 300 python bench.py --no-cpu-baseline --steps 400 --bands $nb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bands $nb side=$v', d['ms_per_step'])"
done; done
