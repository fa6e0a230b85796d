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
cp $SRC $OUT/f_base.hip; mk $OUT/f_base.hip 0base
sed 's/__launch_bounds__(S4K_T) void k_stft4k_power_bwd/__launch_bounds__(S4K_T, 4) void k_stft4k_power_bwd/' $SRC > $OUT/f_lb4.hip; mk $OUT/f_lb4.hip 1lb4
sed 's/__launch_bounds__(S4K_T) void k_stft4k_power_bwd/__launch_bounds__(S4K_T, 2) void k_stft4k_power_bwd/' $SRC > $OUT/f_lb2.hip; mk $OUT/f_lb2.hip 2lb2
wait
ls -la $OUT/fft_*.so
