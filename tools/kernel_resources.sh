#!/bin/bash
# Per-kernel register / LDS / scratch use as the compiler reports it (device-only assembly; no GPU needed).
#   tools/kernel_resources.sh azk_nn.hip|azk_engine.hip [name pattern]
cd "$(dirname "$0")/../alpha-zero_amd/csrc" || exit 1
src=${1:-azk_nn.hip}
extra="-fno-slp-vectorize"; [ "$src" = azk_engine.hip ] && extra="-ffp-contract=off"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include $extra --offload-device-only -S -o /tmp/_azk_dev.s $src || exit 1
awk -v pat="${2:-.}" '
  /^  - \.agpr_count:/ {ag=$3} /\.name:/ {name=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2}
  /\.vgpr_spill_count:/ {sp=$2} /\.private_segment_fixed_size:/ {ps=$2} /\.group_segment_fixed_size:/ {ls=$2}
  /\.wavefront_size:/ { if (name ~ pat) printf "%s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s\n", name, vg, ag, sg, sp, ps, ls }' /tmp/_azk_dev.s | c++filt
