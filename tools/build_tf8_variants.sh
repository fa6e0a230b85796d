# diagnostic variants of csrc/blocktf8.hip (tools/tf8_probe.py --lib <path>): where does a tile's time go?
set -e
cd $(dirname $0)/../diffgfdn_amd/csrc
OBJS=$(ls *.o | grep -v blocktf8.o)
for v in NO_PHASE NO_MFMA NO_SHFL; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DT8_DIAG_$v -c blocktf8.hip -o /tmp/blocktf8_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/blocktf8_$v.o -o ../lib/libvariant_$v.so
done
