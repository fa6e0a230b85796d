set -x
cd $GRAFT_REPO_ROOT
python tools/decay_probe.py 224 --stamps
