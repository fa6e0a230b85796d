#!/bin/bash
# builds variants of csrc/mlp.hip into tools/_probe/ for tools/mlp_probe.py (diagnostic)
set -e
cd "$(dirname "$0")/.."
SRC=diffgfdn_amd/csrc/mlp.hip
OUT=tools/_probe
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -shared -Idiffgfdn_amd/csrc -Iinclude"
mk() { /opt/rocm/bin/hipcc $FLAGS "$1" -o "$OUT/mlp_$2.so"; }
cp $SRC $OUT/v_base.hip; mk $OUT/v_base.hip 0base
sed 's/(float)(r < 3 ? sin(arg) : cos(arg))/(r < 3 ? sinf((float)arg) : cosf((float)arg))/' $SRC > $OUT/v_ftrig.hip; mk $OUT/v_ftrig.hip 1ftrig
sed 's/if (d.stage) return MLP_T; /if (d.stage) return 64; /' $SRC > $OUT/v_t64.hip; mk $OUT/v_t64.hip 2t64
sed 's/#define MLP_STAGE_MAX 7168 /#define MLP_STAGE_MAX 1 /' $SRC > $OUT/v_nostage.hip; mk $OUT/v_nostage.hip 3nostage
