# Build everything in-tree (the .so files travel to the GPU box with the
# snapshot).  `make` = host scene lib + HIP lib + CPU oracle.
#   ezrt_amd/lib/libezrt_scene.so   host C++ scene build (no HIP dependency)
#   ezrt_amd/lib/libezrt_hip.so     hand-written gfx950 kernels + the C ABI
#   oracle/libezrt_oracle.so        CPU oracle (test infrastructure)
#   oracle/_ref/libezrt_ref_p*.so   the reference's own host code compiled from /root/reference (when present)
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

LIBDIR = ezrt_amd/lib
HOST_SRC = ezrt_amd/csrc/host/scene.cpp ezrt_amd/csrc/host/hdr.cpp ezrt_amd/csrc/host/p2_query.cpp ezrt_amd/csrc/host/host_c_api.cpp
HIP_SRC = ezrt_amd/csrc/hip/ezrt_hip.hip ezrt_amd/csrc/hip/ezrt_scene_build.hip ezrt_amd/csrc/hip/ezrt_launch.hip ezrt_amd/csrc/hip/ezrt_lbvh.hip ezrt_amd/csrc/hip/ezrt_sahbvh.hip ezrt_amd/csrc/hip/ezrt_mgpu.hip
HIP_DEPS = $(wildcard ezrt_amd/csrc/hip/*.h) $(wildcard ezrt_amd/csrc/hip/*.hip) $(wildcard include/*)

# -ffp-contract=off everywhere: the trace's discrete decisions must be
# bit-identical between x86 and gfx950 (include/ezrt_detmath.h).
HOST_FLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno -Wall -Wextra -Iinclude
HIP_FLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math \
            -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-function -Iinclude -Iezrt_amd/csrc/hip

all: host hip oracle examples

host: $(LIBDIR)/libezrt_scene.so
hip: $(LIBDIR)/libezrt_hip.so
oracle:
	$(MAKE) -C oracle
	@if [ -d /root/reference ]; then python3 oracle/ref_recipe/build_ref.py; fi

$(LIBDIR)/libezrt_scene.so: $(HOST_SRC) include/ezrt_scene.hpp include/ezrt_scene_c.h include/ezrt_detmath.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOST_FLAGS) -shared -o $@ $(HOST_SRC)

# one object per translation unit (build/ is git-ignored; `make -j4 hip` compiles them in parallel, and a change to the
# trace kernels does not recompile the rocPRIM-heavy builders)
HIP_OBJ = $(patsubst ezrt_amd/csrc/hip/%.hip,build/hip/%.o,$(HIP_SRC))
build/hip/%.o: ezrt_amd/csrc/hip/%.hip $(wildcard ezrt_amd/csrc/hip/*.h) $(wildcard include/*)
	@mkdir -p build/hip
	$(HIPCC) $(HIP_FLAGS) -c -o $@ $<
$(LIBDIR)/libezrt_hip.so: $(HIP_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(HIP_OBJ) -ldl -pthread

# Consumers of the public headers outside the libraries: chapter 5's main() ported onto the C ABI + the
# C++ host API (g++ only: the boundary needs no HIP header), and a C11 layout check of the by-value struct.
EXDIR = examples/bin
examples: $(EXDIR)/p5_main_port $(EXDIR)/abi_layout_check
$(EXDIR)/p5_main_port: examples/p5_main_port.cpp $(LIBDIR)/libezrt_scene.so $(LIBDIR)/libezrt_hip.so include/ezrt.h include/ezrt_scene.hpp
	@mkdir -p $(EXDIR)
	$(CXX) -O2 -std=c++17 -Wall -Wextra -Iinclude -o $@ examples/p5_main_port.cpp -L$(LIBDIR) -lezrt_scene -lezrt_hip \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM)/lib -Wl,-rpath-link,$(ROCM)/lib
$(EXDIR)/abi_layout_check: examples/abi_layout_check.c $(LIBDIR)/libezrt_scene.so $(LIBDIR)/libezrt_hip.so $(wildcard include/*.h)
	@mkdir -p $(EXDIR)
	$(CC) -O1 -std=c11 -Wall -Wextra -pedantic -Iinclude -o $@ examples/abi_layout_check.c -L$(LIBDIR) -lezrt_scene -lezrt_hip \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,$(ROCM)/lib -Wl,-rpath-link,$(ROCM)/lib

clean:
	rm -rf $(EXDIR) build/hip
	rm -f $(LIBDIR)/*.so
	$(MAKE) -C oracle clean

.PHONY: all host hip oracle examples clean negctl

# negative control of tests/test_gpu_prune.py (tools/gpu_negctl.sh): the HIP library with a pruning margin that is
# negative on purpose; never loaded by the product (EZRT_HIP_LIB selects it for that one run)
negctl:
	@mkdir -p build_ab
	$(HIPCC) $(HIP_FLAGS) -DEZRT_PRUNE_NEGATIVE_CONTROL=1 -shared -o build_ab/libezrt_hip_negctl.so $(HIP_SRC) -ldl -pthread
