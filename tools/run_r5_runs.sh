#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
for r in 0 8 4 2; do timeout -k 10 300 python tools/ab_attr.py edr_receiver_runs=$r -- --no-cpu-baseline --no-extras --steps 400 --bands 1 2>/dev/null | tail -1; done
for r in 0 4 1; do timeout -k 10 300 python tools/ab_attr.py edr_receiver_runs=$r -- --no-cpu-baseline --no-extras --steps 400 2>/dev/null | tail -1; done
for r in 0 4; do timeout -k 10 300 python tools/ab_attr.py edr_receiver_runs=$r -- --no-cpu-baseline --no-extras --steps 400 --bands 2 2>/dev/null | tail -1; done
