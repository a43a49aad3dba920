# Builds the GPU library, the C host layer around it, the command-line renderer and the test oracle.
#   make            -> raytracer_amd/librtgpu.so, raytracer_amd/librthost.so, raytracer_amd/raytrace
#   make oracle     -> oracle/liboracle.so (+ oracle/_ref/ref_dump when /root/reference is present)
HIPCC   ?= hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := raytracer_amd
# -ffp-contract=off: the trace path must round exactly like the reference's non-FMA binary32 code.
# -fno-slp-vectorize: packed fp32 (v_pk_*) issues at half rate and needs register shuffles; plain ops are faster here.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC
CFLAGS   := -O2 -std=gnu99 -ffp-contract=off -fPIC -Wall
HOSTSRC  := $(PKG)/host/scene.c $(PKG)/host/bitmap.c $(PKG)/host/raytrace.c $(PKG)/host/error.c

ROCM    ?= /opt/rocm
all: $(PKG)/librtgpu.so $(PKG)/librthost.so $(PKG)/librtmulti.so $(PKG)/raytrace

$(PKG)/librtgpu.so: $(PKG)/csrc/rtgpu.hip $(wildcard $(PKG)/csrc/*.h) include/rtgpu.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $<

$(PKG)/librthost.so: $(HOSTSRC) $(PKG)/host/rt_host.h include/rtgpu.h $(PKG)/librtgpu.so
	$(CC) $(CFLAGS) -shared -o $@ $(HOSTSRC) -L$(PKG) -lrtgpu -Wl,-rpath,'$$ORIGIN' -lm

# the N-GPU host (one process, RCCL): its own library, loaded by the command line only for --devices
$(PKG)/librtmulti.so: $(PKG)/host/multigpu.c $(PKG)/host/rt_host.h include/rtgpu.h $(PKG)/librthost.so
	$(CC) $(CFLAGS) -I$(ROCM)/include -shared -o $@ $< -L$(PKG) -lrthost -lrtgpu -L$(ROCM)/lib -lrccl -lamdhip64 -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib -lm -lpthread

$(PKG)/raytrace: $(PKG)/host/main.c $(PKG)/librthost.so
	$(CC) $(CFLAGS) -o $@ $< -L$(PKG) -lrthost -lrtgpu -Wl,-rpath,'$$ORIGIN' -lm -ldl

oracle:
	$(MAKE) -C oracle oracle
	if [ -d /root/reference/src ]; then $(MAKE) -C oracle ref dropin; fi

clean:
	rm -f $(PKG)/librtgpu.so $(PKG)/librthost.so $(PKG)/librtmulti.so $(PKG)/raytrace

.PHONY: all oracle clean
