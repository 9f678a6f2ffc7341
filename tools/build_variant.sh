#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...]  ->  build/variants/var_NAME.so (+ resource usage of the path kernels)  for A/B runs via HIJIKI_HIP_LIB
set -e
mkdir -p build/variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wall -Wno-unused-function "$@" \
  hijiki_amd/csrc/api/context.hip hijiki_amd/csrc/api/scene_upload.hip hijiki_amd/csrc/api/scene_relayout.hip hijiki_amd/csrc/api/render.hip hijiki_amd/csrc/api/comm.hip \
  hijiki_amd/csrc/api/lbvh_build.hip hijiki_amd/csrc/api/tree_vote.hip hijiki_amd/csrc/host/blockgen.cpp hijiki_amd/csrc/api/light_grid.cpp -ldl -o build/variants/var_$name.so \
  -Rpass-analysis=kernel-resource-usage 2> build/variants/var_$name.txt
grep -A11 "k_path_wavefrontILb1ELb1ELb0" build/variants/var_$name.txt | grep -E "VGPRs|Scratch|Occupancy|LDS" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
