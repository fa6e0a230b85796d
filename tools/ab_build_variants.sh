#!/bin/bash
# Same-box A/B of kernel constants: builds whole-library variants (one sed expression each) into
# tools/_probe/libv_<name>.so; tools/ab_run_variants.sh then swaps them in on ONE GPU box and runs bench.py for each
# (box-to-box noise is 1-2 %, same-box noise 0.3 %).  Edit the `mk` lines for the variants to compare.
set -e
cd "$(dirname "$0")/../diffgfdn_amd/csrc"
mkdir -p ../../tools/_probe
rm -f ../../tools/_probe/libv_*.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
OUT=../../tools/_probe
mk() { # name file sed-expr
  sed "$3" $2 > ./_v_$1.hip
  /opt/rocm/bin/hipcc $FLAGS -c _v_$1.hip -o /tmp/v_$1.o
  rm -f _v_$1.hip
  OBJS=""
  for f in $(sed -n 's/^SRCS *:= *//p' Makefile | sed 's/\.hip//g'); do
    if [ "$f.hip" = "$2" ]; then OBJS="$OBJS /tmp/v_$1.o"; else OBJS="$OBJS $f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libv_$1.so
}
cp ../lib/libdiffgfdn_hip.so $OUT/libv_base.so
# mk <name> <file.hip> '<sed expression>' &      e.g.:
# mk p2tc16 pow2.hip 's/#define P2_TC 8/#define P2_TC 16/' &
wait
ls $OUT/libv_*.so
