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
HOST_SRC = hijiki_amd/csrc/host/scene.cpp hijiki_amd/csrc/host/synth.cpp hijiki_amd/csrc/host/blockgen.cpp \
           hijiki_amd/csrc/host/obj_loader.cpp hijiki_amd/csrc/host/image_io.cpp hijiki_amd/csrc/host/host_api.cpp
HOST_HDR = hijiki_amd/csrc/host/scene.hpp hijiki_amd/csrc/host/blockgen.hpp include/hijiki_hip.h include/hijiki_host.h
HIP_SRC = hijiki_amd/csrc/hj_api.hip hijiki_amd/csrc/host/blockgen.cpp
HIP_HDR = $(wildcard hijiki_amd/csrc/kernels/*.h) $(wildcard hijiki_amd/csrc/*.h) include/hijiki_hip.h hijiki_amd/csrc/host/blockgen.hpp

all: host hip oracle cli
host: hijiki_amd/lib/libhijiki_host.so
hip: hijiki_amd/lib/libhijiki_hip.so
oracle: oracle/_build/libhj_oracle.so
cli: hijiki_amd/bin/hijiki-hip

hijiki_amd/lib/libhijiki_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p hijiki_amd/lib
	$(CXX) -std=c++17 -O2 -g0 -fPIC -shared -Wall -Wextra $(FP_STRICT) -fvisibility=hidden \
	  -DHJ_BUILDING -pthread -o $@ $(HOST_SRC)

hijiki_amd/lib/libhijiki_hip.so: $(HIP_SRC) $(HIP_HDR)
	@mkdir -p hijiki_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -shared $(FP_STRICT) -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden \
	  -Wall -Wno-unused-function hijiki_amd/csrc/hj_api.hip hijiki_amd/csrc/host/blockgen.cpp -ldl -o $@ \
	  -Rpass-analysis=kernel-resource-usage 2> hijiki_amd/lib/resource_usage.txt || (cat hijiki_amd/lib/resource_usage.txt; false)
	@strings $@ | grep -q 'amdgcn-amd-amdhsa--$(ARCH)' || (echo 'ERROR: no $(ARCH) code object in $@'; rm -f $@; false)

oracle/_build/libhj_oracle.so: oracle/hj_oracle.c include/hijiki_hip.h
	@mkdir -p oracle/_build
	$(CC) -std=c11 -O2 -fPIC -shared -Wall -Wextra $(FP_STRICT) -mfma -fvisibility=hidden -o $@ oracle/hj_oracle.c -lm -lpthread

CLI_SRC = hijiki_amd/csrc/cli/main.cpp hijiki_amd/csrc/host/scene.cpp hijiki_amd/csrc/host/synth.cpp \
          hijiki_amd/csrc/host/obj_loader.cpp hijiki_amd/csrc/host/image_io.cpp
hijiki_amd/bin/hijiki-hip: $(CLI_SRC) $(HOST_HDR) hijiki_amd/lib/libhijiki_hip.so
	@mkdir -p hijiki_amd/bin
	$(CXX) -std=c++17 -O2 -pthread -Wall -Wextra $(FP_STRICT) -o $@ $(CLI_SRC) -Lhijiki_amd/lib -lhijiki_hip \
	  -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,$(ROCM)/lib -L$(ROCM)/lib -lamdhip64

clean:
	rm -rf hijiki_amd/lib hijiki_amd/bin oracle/_build

.PHONY: all host hip oracle cli clean
