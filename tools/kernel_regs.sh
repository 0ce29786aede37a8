#!/bin/bash
# usage: tools/kernel_regs.sh [extra -D flags]   prints registers / LDS / spills of every kernel (device-only compile to assembly)
set -e
out=${TMPDIR:-/tmp}/rt_kernels_regs.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 --offload-device-only -S "$@" \
    -o "$out" "$(dirname "$0")/../cudaraytracer_amd/csrc/kernels/rt_kernels.hip"
awk '/^; Kernel info|\.amdhsa_kernel /{name=$2} /\.amdhsa_kernel /{k=$2} /; NumVgprs:/{v=$3} /; ScratchSize:/{s=$3} /; Occupancy:/{printf "%-90s vgprs %s scratch %s occupancy %s\n", k, v, s, $3}' "$out"
echo "assembly: $out"
