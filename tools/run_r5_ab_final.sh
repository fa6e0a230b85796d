#!/bin/bash
# round 5: every switch of the round flipped once on the FINAL code, one box, one call -> profiles/r05_ab_same_box.txt
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_ab_same_box.txt
B="--no-cpu-baseline --no-extras --steps 400"
ab() { timeout -k 10 300 python tools/ab_attr.py "$@" 2>/dev/null | tail -1; }
{
echo "# same-box A/B on the round's final code (tools/run_r5_ab_final.sh; one gpurun call, bench.py $B):"
echo "# class attributes of bankstep.FusedBankStep flipped for one process each; ms per 7-band step"
echo "## N = 16 (headline)"
for rep in 1 2; do
ab -- $B
ab scale_in_gains=False -- $B
ab scale_late=False -- $B
ab edc_one_launch=False -- $B
ab gamma_split=False -- $B
ab fused_tail=False -- $B
ab scale_late=False edc_one_launch=False gamma_split=False fused_tail=False -- $B
done
echo "## one band (--bands 1)"
ab -- --bands 1 $B
ab scale_late=False edc_one_launch=False gamma_split=False fused_tail=False -- --bands 1 $B
echo "## N = 32 (--lines-per-group 8)"
B8="--lines-per-group 8 --no-cpu-baseline --no-extras --steps 300"
for rep in 1 2; do
ab -- $B8
ab scale_in_gains=False -- $B8
ab fused_tail=False -- $B8
ab transform_polys=False -- $B8
ab transform_polys=False scale_late=False edc_one_launch=False gamma_split=False fused_tail=False -- $B8
done
echo "## directional (--config directional; ms per band-step; the bench is noisy at +-4 %)"
for rep in 1 2 3; do
timeout -k 10 200 python tools/ab_dir.py -- --steps 40 2>/dev/null | tail -1
timeout -k 10 200 python tools/ab_dir.py sub_fdn_by_transforms=False -- --steps 40 2>/dev/null | tail -1
done
} > $OUT 2>&1
cat $OUT
