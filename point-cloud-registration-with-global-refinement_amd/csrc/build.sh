#!/bin/bash
# Builds libpcr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
ARCH=${PCR_ARCH:-gfx950}
FLAGS="-O3 --offload-arch=$ARCH -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable $PCR_EXTRA_FLAGS"
OUT=${PCR_OUT:-../libpcr_hip.so}
objs=""
for f in pcr_sort pcr_cloud pcr_gicp pcr_featnn pcr_fgr pcr_api; do
  [ -f $f.hip ] || continue
  stale=0
  for h in $f.hip *.h ../../include/pcr_hip.h; do [ $h -nt $f.o ] && stale=1; done
  if [ ! -f $f.o ] || [ $stale = 1 ]; then
    echo "hipcc $f.hip"
    hipcc $FLAGS -c $f.hip -o $f.o
  fi
  objs="$objs $f.o"
done
hipcc --offload-arch=$ARCH -shared -fPIC -pthread -o $OUT $objs
echo "built $OUT"
