# same-box A/B of class attributes of bankstep.FusedBankStep:  bash tools/run_ab.sh <tag> "<bench args>" "<overrides 1>" "<overrides 2>" ...
# ("-" = defaults); every setting is run twice, interleaved
TAG=$1; BARGS=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/${TAG}_ab.log
for rep in 1 2; do
  for ov in "$@"; do
    if [ "$ov" = "-" ]; then ov=""; fi
    timeout -k 10 200 python tools/ab_attr.py $ov -- --no-cpu-baseline --no-extras --steps 400 $BARGS 2>/dev/null | grep ms_per_step >> $OUT/${TAG}_ab.log || exit 1
  done
done
cat $OUT/${TAG}_ab.log
