# Build everything in-tree:
#   rust-path-tracer_amd/lib/librpt_hip.so   HIP kernels + C ABI (gfx950, hipcc)
#   rust-path-tracer_amd/lib/librpt_host.so  host-side dispatch mirror (g++)
#   oracle/liboracle.so, oracle/liboracle_libm.so   CPU restatement (test infrastructure)
# -ffp-contract=off everywhere: the parity contract forbids implicit FMA fusion.

PKG      := rust-path-tracer_amd
LIBDIR   := $(PKG)/lib
CSRC     := $(PKG)/csrc
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++

HOST_SRCS := $(CSRC)/host/host_api.cpp $(CSRC)/host/glb_scene.cpp $(CSRC)/host/bvh_build.cpp \
             $(CSRC)/host/light_table.cpp $(CSRC)/host/bluenoise.cpp $(CSRC)/host/image_io.cpp $(CSRC)/host/textures.cpp $(CSRC)/host/obj_scene.cpp $(CSRC)/host/jpeg_decode.cpp
HIP_SRCS  := $(CSRC)/rpt_hip.hip $(CSRC)/rpt_traverse.hip $(CSRC)/rpt_comm.hip $(CSRC)/rpt_lights.hip $(CSRC)/rpt_bvh.hip
HIP_OBJS  := $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS))
HIP_DEPS  := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hip) $(wildcard include/rpt/*.h)

CXXFLAGS_COMMON := -std=c++20 -O2 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unknown-pragmas
# -mfma only makes the EXPLICIT __builtin_fma of rpt_math.h a single instruction;
# implicit contraction stays off.
ORACLE_FLAGS := $(CXXFLAGS_COMMON) -mfma -msse4.1 -pthread
HIPFLAGS := --offload-arch=gfx950 -std=c++20 -O3 -fPIC -ffp-contract=off -fno-fast-math \
            -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
            -Wall -Wno-unused-function -fno-slp-vectorize
# -fno-slp-vectorize: the packed f32 operations the SLP vectorizer forms issue in the time of two plain ones on gfx950 and cost v_mov
# shuffles and registers (profiles/r03_slp.txt)

all: host oracle hip fake_rccl

host: $(LIBDIR)/librpt_host.so
oracle: oracle/liboracle.so oracle/liboracle_libm.so
hip: $(LIBDIR)/librpt_hip.so

$(LIBDIR)/librpt_host.so: $(HOST_SRCS) $(CSRC)/host/host_internal.h $(CSRC)/rpt_math.h include/rpt/rpt_host.h include/rpt/rpt.h include/rpt/shared_structs.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS_COMMON) -shared -o $@ $(HOST_SRCS) -lz -ldl -pthread

oracle/liboracle.so: oracle/rpt_oracle.cpp oracle/bvh_oracle.cpp $(CSRC)/rpt_math.h $(CSRC)/rpt_math_consts.h include/rpt/shared_structs.h
	$(CXX) $(ORACLE_FLAGS) -shared -o $@ oracle/rpt_oracle.cpp oracle/bvh_oracle.cpp

oracle/liboracle_libm.so: oracle/rpt_oracle.cpp oracle/bvh_oracle.cpp $(CSRC)/rpt_math.h $(CSRC)/rpt_math_consts.h include/rpt/shared_structs.h
	$(CXX) $(ORACLE_FLAGS) -DORACLE_USE_LIBM -shared -o $@ oracle/rpt_oracle.cpp oracle/bvh_oracle.cpp -lm

# the fingerprint of the device-side sources is compiled in (rpt_build_fingerprint): bench.py reports PMC-derived figures only for the
# sources the LOADED library was built from, and says so when the library is older than the source tree
# one object per translation unit, built in PARALLEL (the recipe below re-invokes make with -j): the walk kernels (rpt_traverse.hip) and the
# shade / sky / set-up kernels (rpt_hip.hip) each take about 20 s, one after the other they were 50 s
build/%.o: $(CSRC)/%.hip $(HIP_DEPS) tools/source_fingerprint.py Makefile
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DRPT_BUILD_FINGERPRINT=\"$$(python3 tools/source_fingerprint.py)\" -c -o $@ $<

$(LIBDIR)/librpt_hip.so: $(HIP_SRCS) $(HIP_DEPS) tools/source_fingerprint.py Makefile
	@mkdir -p $(LIBDIR)
	$(MAKE) -j5 $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(HIP_OBJS) -ldl

# test infrastructure: a stand-in for RCCL's point-to-point calls over shared memory, so that N PROCESSES on a one-GPU test box run
# the product's gather unchanged (tests/test_gpu_multiprocess.py; selected with RPT_RCCL_LIBRARY, never linked by the product)
fake_rccl: tests/fake_rccl/librccl_fake.so
tests/fake_rccl/librccl_fake.so: tests/fake_rccl/fake_rccl.cpp
	$(HIPCC) -x c++ -std=c++17 -O2 -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -o $@ $< -L/opt/rocm/lib -lamdhip64 -lrt -pthread

clean:
	rm -f $(LIBDIR)/*.so oracle/*.so tests/fake_rccl/*.so

.PHONY: all host oracle hip fake_rccl clean
