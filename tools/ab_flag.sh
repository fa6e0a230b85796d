# same-box A/B of a bench.py flag: bash tools/ab_flag.sh "--flag" [rounds]
cd $GRAFT_REPO_ROOT
F=$1; R=${2:-3}
for i in $(seq $R); do
  for v in "" "$F"; do
    timeout 300 python bench.py --no-cpu-baseline --steps 400 $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'])"
  done
done
