# rtcore-mi355x — builds the product library, the CPU oracle (test infrastructure) and the C++ host.
#   make            -> raytracer_amd/lib/librtcore.so, oracle/liboracle.so, raytracer_amd/bin/rt_host
#   make lib|oracle|host|clean
HIPCC     ?= /opt/rocm/bin/hipcc
ARCH      ?= gfx950
# strict IEEE f32 on the device: no contraction, correctly rounded div/sqrt, denormals kept.
# -fno-slp-vectorize: the SLP vectoriser pairs f32 operations into v_pk_* and pays for it with v_mov shuffles and 4 - 9 more
# live registers in the traversal kernels (same results bit for bit; cfg 3 batches 0.2445 -> 0.2256 ms per step, PT-16 2.54 -> 2.15:
# profiles/r3/ab_slp_vectorize.txt)
HIPFLAGS  ?= --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
             -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function $(EXTRA)
CSRC      := raytracer_amd/csrc
LIB       := raytracer_amd/lib/librtcore.so
OBJS      := build/rt_api.o build/rt_build.o build/rt_trace.o build/rt_scene.o

all: lib oracle host

lib: $(LIB)

build/%.o: $(CSRC)/%.hip $(CSRC)/rt_internal.h $(CSRC)/rt_device.h include/rtcore.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/rt_scene.o: $(CSRC)/rt_scene.cpp include/rtcore.h
	@mkdir -p build
	$(HIPCC) -std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -Wall -x c++ -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p raytracer_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

host: raytracer_amd/bin/rt_host

raytracer_amd/bin/rt_host: src/main.cpp include/rtcore.h $(LIB)
	@mkdir -p raytracer_amd/bin
	g++ -std=c++17 -O2 -Wall -Wextra -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include src/main.cpp -o $@ -Lraytracer_amd/lib -lrtcore \
	    -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib

clean:
	rm -rf build raytracer_amd/lib/librtcore.so raytracer_amd/bin
	$(MAKE) -C oracle clean

.PHONY: all lib oracle host clean
