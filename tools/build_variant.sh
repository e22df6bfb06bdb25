#!/bin/bash
# Build a variant of libsslam_frontend.so with extra compile flags for ONE translation unit (default lines.hip), next to the product
# library: tools/build_variant.sh NAME "-DSSLAM_LSD_DRIFT=0" [unit.hip]  ->  structure-slam-pointline_amd/lib/variants/NAME.so
# Run it with SSLAM_LIB=<that path> (the harness binding honours it); used for A/B kernel measurements in one gpurun call.
set -e
root=$(cd "$(dirname "$0")/.." && pwd); P=$root/structure-slam-pointline_amd
name=$1; flags=$2; unit=${3:-lines.hip}
mkdir -p $P/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-result -Wno-pass-failed $flags -c $P/csrc/$unit -o $P/lib/variants/$name.o
objs=""; for o in $P/lib/obj/*.hip.o; do [ "$(basename $o)" = "$unit.o" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/variants/$name.so $P/lib/variants/$name.o $objs
rm -f $P/lib/variants/$name.o
echo $P/lib/variants/$name.so
