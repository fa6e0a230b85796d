#!/bin/bash
# A whole-library variant with one source rebuilt under extra compiler flags, into tools/_probe/lib_<name>.so (not shipped in
# the product; loaded by probes through diffgfdn_amd._lib.LIB_PATH).   usage: tools/build_probe_lib.sh <name> <file.hip> <flags...>
set -e
cd "$(dirname "$0")/../diffgfdn_amd/csrc"
NAME=$1; SRC=$2; shift; shift
mkdir -p ../../tools/_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c $SRC -o /tmp/pv_$NAME.o
OBJS=""
for f in $(sed -n 's/^SRCS *:= *//p' Makefile | sed 's/\.hip//g'); do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS /tmp/pv_$NAME.o"; else OBJS="$OBJS $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../../tools/_probe/lib_$NAME.so
ls -la ../../tools/_probe/lib_$NAME.so
