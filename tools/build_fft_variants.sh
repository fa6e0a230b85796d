#!/bin/bash
# builds variants of csrc/fft.hip into tools/_probe/ for tools/fft_probe.py (diagnostic)
set -e
cd "$(dirname "$0")/.."
SRC=diffgfdn_amd/csrc/fft.hip
OUT=tools/_probe
mkdir -p $OUT
rm -f $OUT/fft_*.so
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -shared -Idiffgfdn_amd/csrc -Iinclude"
mk() { /opt/rocm/bin/hipcc $FLAGS "$1" -o "$OUT/fft_$2.so" & }
sed 's/if (col128 \&\& rader \&\& g.L2 == 512) {/if (false \&\& col128 \&\& rader \&\& g.L2 == 512) {/' $SRC > $OUT/f_nopair.hip; mk $OUT/f_nopair.hip 0nopair
cp $SRC $OUT/f_base.hip; mk $OUT/f_base.hip 1base
wait
ls -la $OUT/fft_*.so
