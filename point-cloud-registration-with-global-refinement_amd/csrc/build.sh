#!/bin/bash
# Builds libpcr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
ARCH=${PCR_ARCH:-gfx950}
FLAGS="-O3 --offload-arch=$ARCH -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable"
OUT=../libpcr_hip.so
objs=""
for f in pcr_sort pcr_cloud pcr_gicp pcr_featnn pcr_fgr pcr_api; do
  [ -f $f.hip ] || continue
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ pcr_internal.h -nt $f.o ] || [ pcr_device.h -nt $f.o ] || [ pcr_octree.h -nt $f.o ] || [ ../../include/pcr_hip.h -nt $f.o ]; then
    echo "hipcc $f.hip"
    hipcc $FLAGS -c $f.hip -o $f.o
  fi
  objs="$objs $f.o"
done
hipcc --offload-arch=$ARCH -shared -fPIC -pthread -o $OUT $objs
echo "built $OUT"
