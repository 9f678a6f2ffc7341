# Build of the three native pieces (no cmake needed):
#   hijiki_amd/lib/libhijiki_host.so  — C++ host: Scene / compile / BVH / block generator   (g++)
#   hijiki_amd/lib/libhijiki_hip.so   — HIP kernels + C ABI for gfx950                      (hipcc)
#   oracle/_build/libhj_oracle.so     — CPU restatement, test infrastructure only           (gcc)
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX ?= g++
CC ?= gcc
ARCH ?= gfx950

FP_STRICT = -ffp-contract=off -fno-fast-math
HOST_SRC = hijiki_amd/csrc/host/scene.cpp hijiki_amd/csrc/host/tree_opt.cpp hijiki_amd/csrc/host/synth.cpp hijiki_amd/csrc/host/blockgen.cpp \
           hijiki_amd/csrc/host/obj_loader.cpp hijiki_amd/csrc/host/image_io.cpp hijiki_amd/csrc/host/host_api.cpp
HOST_HDR = hijiki_amd/csrc/host/scene.hpp hijiki_amd/csrc/host/blockgen.hpp include/hijiki_hip.h include/hijiki_host.h
# libhijiki_hip.so: its translation units (hijiki_amd/csrc/api/hj_internal.h lists them); render.hip, scene_relayout.hip,
# lbvh_build.hip and tree_vote.hip hold device code.  The register / scratch / LDS report of the path kernels: hijiki_amd/lib/resource_usage.txt.
HIP_UNITS = context scene_upload scene_relayout render comm lbvh_build tree_vote
HIP_OBJ = $(HIP_UNITS:%=build/obj/%.o) build/obj/blockgen.o build/obj/light_grid.o
HIP_HDR = $(wildcard hijiki_amd/csrc/kernels/*.h) hijiki_amd/csrc/api/hj_internal.h hijiki_amd/csrc/api/hj_tuning.h hijiki_amd/csrc/api/light_grid.hpp hijiki_amd/csrc/api/scene_relayout.hpp hijiki_amd/csrc/api/tree_vote.hpp include/hijiki_hip.h hijiki_amd/csrc/host/blockgen.hpp
HIP_FLAGS = --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC $(FP_STRICT) -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden \
            -Wall -Wno-unused-function $(HIP_EXTRA)

all: host hip oracle cli
host: hijiki_amd/lib/libhijiki_host.so
hip: hijiki_amd/lib/libhijiki_hip.so
oracle: oracle/_build/libhj_oracle.so
cli: hijiki_amd/bin/hijiki-hip

hijiki_amd/lib/libhijiki_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p hijiki_amd/lib
	$(CXX) -std=c++17 -O2 -g0 -fPIC -shared -Wall -Wextra $(FP_STRICT) -fvisibility=hidden \
	  -DHJ_BUILDING -pthread -o $@ $(HOST_SRC)

build/obj/render.o: hijiki_amd/csrc/api/render.hip $(HIP_HDR)
	@mkdir -p build/obj hijiki_amd/lib
	$(HIPCC) $(HIP_FLAGS) -c $< -o $@ -Rpass-analysis=kernel-resource-usage 2> hijiki_amd/lib/resource_usage.txt \
	  || (cat hijiki_amd/lib/resource_usage.txt; false)

build/obj/%.o: hijiki_amd/csrc/api/%.hip $(HIP_HDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIP_FLAGS) -c $< -o $@

build/obj/light_grid.o: hijiki_amd/csrc/api/light_grid.cpp hijiki_amd/csrc/api/light_grid.hpp hijiki_amd/csrc/api/hj_tuning.h include/hijiki_hip.h
	@mkdir -p build/obj
	$(CXX) -std=c++17 -O2 -fPIC -pthread -Wall -Wextra $(FP_STRICT) -fvisibility=hidden -c $< -o $@

build/obj/blockgen.o: hijiki_amd/csrc/host/blockgen.cpp hijiki_amd/csrc/host/blockgen.hpp include/hijiki_hip.h
	@mkdir -p build/obj
	$(CXX) -std=c++17 -O2 -fPIC $(FP_STRICT) -fvisibility=hidden -c $< -o $@

hijiki_amd/lib/libhijiki_hip.so: $(HIP_OBJ)
	@mkdir -p hijiki_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIP_OBJ) -ldl -o $@
	@strings $@ | grep -q 'amdgcn-amd-amdhsa--$(ARCH)' || (echo 'ERROR: no $(ARCH) code object in $@'; rm -f $@; false)
	@python3 tools/build_stamp.py

oracle/_build/libhj_oracle.so: oracle/hj_oracle.c include/hijiki_hip.h
	@mkdir -p oracle/_build
	$(CC) -std=c11 -O2 -fPIC -shared -Wall -Wextra $(FP_STRICT) -mfma -fvisibility=hidden -o $@ oracle/hj_oracle.c -lm -lpthread

CLI_SRC = hijiki_amd/csrc/cli/main.cpp hijiki_amd/csrc/host/scene.cpp hijiki_amd/csrc/host/tree_opt.cpp hijiki_amd/csrc/host/synth.cpp \
          hijiki_amd/csrc/host/obj_loader.cpp hijiki_amd/csrc/host/image_io.cpp
hijiki_amd/bin/hijiki-hip: $(CLI_SRC) $(HOST_HDR) hijiki_amd/lib/libhijiki_hip.so
	@mkdir -p hijiki_amd/bin
	$(CXX) -std=c++17 -O2 -pthread -Wall -Wextra $(FP_STRICT) -o $@ $(CLI_SRC) -Lhijiki_amd/lib -lhijiki_hip \
	  -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,$(ROCM)/lib -L$(ROCM)/lib -lamdhip64

clean:
	rm -rf hijiki_amd/lib hijiki_amd/bin oracle/_build build/obj

.PHONY: all host hip oracle cli clean
