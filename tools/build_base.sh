#!/bin/bash
# build_base.sh [REV] : build libhijiki_hip.so of a git revision (default HEAD) into build/variants/var_base.so for same-box A/B runs
set -e
rev=${1:-HEAD}
d=gpurun_out/basebuild; rm -rf $d; mkdir -p $d
git archive $rev hijiki_amd/csrc include | tar -x -C $d
src="hijiki_amd/csrc/hj_api.hip"                                   # (revisions before the split of round 4)
[ -d $d/hijiki_amd/csrc/api ] && src=$(cd $d && ls hijiki_amd/csrc/api/*.hip | tr '\n' ' ')
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -w $src hijiki_amd/csrc/host/blockgen.cpp -ldl -o var_base.so)
mkdir -p build/variants; cp $d/var_base.so build/variants/var_base.so; rm -rf $d
echo "built build/variants/var_base.so from $rev"
