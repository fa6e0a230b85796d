# same-box A/B of an environment switch: bash tools/ab_env.sh VAR [rounds]
set -e
cd $GRAFT_REPO_ROOT
VAR=$1; R=${2:-3}
for i in $(seq $R); do
  for v in 0 1; do
    env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'])"
  done
done
