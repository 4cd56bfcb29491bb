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
    # pcr_fgr: no SLP vectorisation = no packed-FP32 instructions.  On MI355X a v_pk_mul_f32 that reads the result of a v_rsq_f32 issued ten
    # instructions earlier took a STALE value in one of its halves when other kernels kept the transcendental pipe busy (measured:
    # tools/lab/fpfh_race.py); tools/pk_trans_scan.py (a CPU test) checks that no unit holds such a pair.
    extra=""; [ $f = pcr_fgr ] && extra="-fno-slp-vectorize"
    hipcc $FLAGS $extra -c $f.hip -o $f.o
  fi
  objs="$objs $f.o"
done
hipcc --offload-arch=$ARCH -shared -fPIC -pthread -o $OUT $objs
echo "built $OUT"
