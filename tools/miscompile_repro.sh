#!/bin/bash
# Reproduces, without a GPU, the code-generation defect behind round 3's wrong result:
# compiles lbm2m_kernel<float, D3Q27, slab, BGK, 64 x 4, AX = 0> from the first failing commit (977a956) to ISA at
# -O3, at -O2 and at -O3 without hipcc's SIOptimizeVGPRLiveRange pass, and counts the register copies the backend
# dropped because it took their source for undefined ("; kill: def $vgprA killed $vgprB").  Expected: 3 / 0 / 0.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
git -C "$ROOT" archive 977a956 lettuce_amd/csrc include | tar -x -C "$W"
cd "$W/lettuce_amd/csrc"
printf '#include "twostep_masked.hpp"\ntemplate __global__ void lt::lbm2m_kernel<float, lt::D3Q27, 1, 1, 64, 4, 0>(const lt::KParams<float>, const int);\n' > one.hip
for v in "-O3" "-O2" "-O3 -mllvm -amdgpu-opt-vgpr-liverange=false"; do
  /opt/rocm/bin/hipcc $v -std=c++17 --offload-arch=gfx950 -I../../include -S --cuda-device-only one.hip -o one.s 2>/dev/null
  echo "$v: $(grep -c 'kill: def \$vgpr[0-9]* killed \$vgpr[0-9]* ' one.s) dropped copies"
  grep -n -B6 -A1 'kill: def \$vgpr[0-9]* killed \$vgpr[0-9]* ' one.s | cut -c1-110 | head -30
done
