#!/bin/bash
# Diagnostic builds of ONE translation unit with extra flags into a separate library next to the product one:
#   tools/build_variant.sh <name> <unit> "<flags>"   ->  point-cloud-registration-with-global-refinement_amd/libpcr_hip_<name>.so
# (loaded through PCR_HIP_SO=<path>, tools only; the product library is untouched)
set -e
cd "$(dirname "$0")/../point-cloud-registration-with-global-refinement_amd/csrc"
NAME=$1; UNIT=$2; FLAGS=$3
extra=""; [ $UNIT = pcr_fgr ] && extra="-fno-slp-vectorize"       # (as csrc/build.sh)
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w $extra $FLAGS -c $UNIT.hip -o /tmp/${UNIT}_$NAME.o
objs=""
for f in pcr_sort pcr_cloud pcr_gicp pcr_featnn pcr_fgr pcr_api; do
  if [ $f = $UNIT ]; then objs="$objs /tmp/${UNIT}_$NAME.o"; else objs="$objs $f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libpcr_hip_$NAME.so $objs
echo "built libpcr_hip_$NAME.so"
