#!/bin/bash
# Register / LDS / spill usage of the kernels in one .hip file whose name matches $2 (compiler remarks; no GPU needed).
#   tools/kres.sh libllsm2_amd/csrc/kernels.hip k_harm_speech
f=${1:-libllsm2_amd/csrc/kernels.hip}; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -Iinclude -Ilibllsm2_amd/csrc -c "$f" -o /dev/null --cuda-device-only -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -A 12 "Function Name: .*$pat" | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|SGPRs:"
