# diagnostic build of the fused decay kernel with phase stamps (tools/decay_probe.py --stamps)
set -e
cd $(dirname $0)/../diffgfdn_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DDK_STAMPS ${DK_EXTRA} -c decay.hip -o /tmp/decay_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/decay_stamps.o fft.o -o ../lib/libdecay_stamps.so
