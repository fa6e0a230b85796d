#!/bin/bash
# Same-box A/B of kernel constants: builds whole-library variants (one sed expression each) into
# tools/_probe/libv_<name>.so; tools/ab_run_variants.sh then swaps them in on ONE GPU box and runs bench.py for each
# (box-to-box noise is 1-2 %, same-box noise 0.3 %).  Edit the `mk` lines for the variants to compare.
set -e
cd "$(dirname "$0")/../diffgfdn_amd/csrc"
mkdir -p ../../tools/_probe
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
OUT=../../tools/_probe
mk() { # name file sed-expr
  sed "$3" $2 > /tmp/v_$1.hip
  cp /tmp/v_$1.hip ./_v_$1.hip
  /opt/rocm/bin/hipcc $FLAGS -c _v_$1.hip -o /tmp/v_$1.o
  rm -f _v_$1.hip
  OBJS=""
  for f in solve ortho fft pow2 losses optim mlp svf; do
    if [ "$f.hip" = "$2" ]; then OBJS="$OBJS /tmp/v_$1.o"; else OBJS="$OBJS $f.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libv_$1.so
}
cp ../lib/libdiffgfdn_hip.so $OUT/libv_base.so
mk colfadj2 fft.hip 's/(tw2_elems + 4 \* CW_LDS) \* sizeof(float2), s, a);/(tw2_elems + (a.adjoint \&\& a.pair ? 8 : 4) * CW_LDS) * sizeof(float2), s, a);/' &
mk rpb16 solve.hip 's/if ((long long)ktiles \* nbands >= 768) rpb = ((B + COMPOSE_BCH - 1) \/ COMPOSE_BCH) \* COMPOSE_BCH;/if ((long long)ktiles * nbands >= 768) rpb = 16;/' &
mk edr128 losses.hip 's/const int f = blockIdx.x \* 256 + threadIdx.x;/const int f = blockIdx.x * blockDim.x + threadIdx.x;/; s/hipLaunchKernelGGL(k_edr_loss_cols, dim3(fblk, batch), dim3(256)/hipLaunchKernelGGL(k_edr_loss_cols, dim3(2 * fblk, batch), dim3(128)/' &
wait
ls $OUT/libv_*.so
