#!/bin/bash
# build_base.sh [REV] : build libhijiki_hip.so of a git revision (default HEAD) into build/variants/var_base.so for same-box A/B runs
set -e
rev=${1:-HEAD}
d=gpurun_out/basebuild; rm -rf $d; mkdir -p $d
git archive $rev hijiki_amd/csrc include | tar -x -C $d
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -w hijiki_amd/csrc/hj_api.hip hijiki_amd/csrc/host/blockgen.cpp -ldl -o var_base.so)
cp $d/var_base.so build/variants/var_base.so; rm -rf $d
echo "built build/variants/var_base.so from $rev"
