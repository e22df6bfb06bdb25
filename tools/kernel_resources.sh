#!/bin/bash
# Per-kernel register / scratch / LDS / occupancy table from the compiler's resource-usage remarks (no GPU needed).
# usage: tools/kernel_resources.sh structure-slam-pointline_amd/csrc/lines.hip
set -e
src=$(realpath "$1"); root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math $SSLAM_EXTRA_FLAGS -I"$root/include" -I"$(dirname "$src")" \
  -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/ {n=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {printf "%-60s vgpr %4s scratch %5s occ %2s lds %6s\n", n, v, s, o, $NF}' |
  c++filt | sed 's/(anonymous namespace):://; s/(.*)//'
