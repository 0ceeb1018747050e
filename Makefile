# Builds every native artefact in-tree (the built files travel to the GPU box with the
# tree; they are git-ignored).
#
#   make            -> host library, device library (gfx950), CLI, oracle checkers
#   make product    -> only what ships: librtx_host.so, librtx_amd.so, raytracer_amd
#   make checkers   -> oracle/librt_oracle.so and, when /root/reference exists, oracle/_ref/* (the compiled
#                      reference, its dumper, and the reference-side bridge's test program)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
LIBDIR  := raytracer_amd/lib
HOSTSRC := raytracer_amd/host
DEVSRC  := raytracer_amd/csrc

# -ffp-contract=off everywhere: hit selection must round exactly like the reference
# binary, which contains no fused multiply-adds.
HOSTFLAGS := -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wextra -Iinclude
# -fno-strict-aliasing: the kernels view the same LDS rows and global records through several
# types (integer keys, 16-byte pieces, doubles); type-based alias analysis is not to reason about that.
DEVFLAGS  := -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -fno-strict-aliasing \
             -munsafe-fp-atomics -fPIC -Wall -Iinclude

.PHONY: all product checkers clean
all: product checkers
product: $(LIBDIR)/librtx_host.so $(LIBDIR)/librtx_amd.so $(LIBDIR)/raytracer_amd

$(LIBDIR)/librtx_host.so: $(HOSTSRC)/host_scene.cpp $(HOSTSRC)/json_min.hpp $(HOSTSRC)/ply_reader.hpp include/rtx_host.h include/rtx_amd.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOSTSRC)/host_scene.cpp

# the BVH builder (rocPRIM templates, slow to compile) is its own object
$(LIBDIR)/rtx_bvh_build.o: $(DEVSRC)/rtx_bvh_build.hip include/rtx_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(DEVFLAGS) -c -o $@ $(DEVSRC)/rtx_bvh_build.hip

$(LIBDIR)/rtx_core.o: $(DEVSRC)/rtx_core.hip $(wildcard $(DEVSRC)/*.h) include/rtx_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(DEVFLAGS) -c -o $@ $(DEVSRC)/rtx_core.hip

$(LIBDIR)/librtx_amd.so: $(LIBDIR)/rtx_core.o $(LIBDIR)/rtx_bvh_build.o
	$(HIPCC) -shared -fPIC -o $@ $(LIBDIR)/rtx_core.o $(LIBDIR)/rtx_bvh_build.o

$(LIBDIR)/raytracer_amd: $(HOSTSRC)/main.cpp $(LIBDIR)/librtx_host.so $(LIBDIR)/librtx_amd.so
	$(CXX) $(HOSTFLAGS) -pthread -o $@ $(HOSTSRC)/main.cpp -L$(LIBDIR) -lrtx_host -lrtx_amd \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib

checkers: product
	$(MAKE) -C oracle oracle ref bridge

clean:
	rm -rf $(LIBDIR)
	$(MAKE) -C oracle clean
