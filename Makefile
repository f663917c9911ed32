# Build libleopard_amd.so (gfx950) in-tree, the oracle helpers, and the CPU kernel-logic emulator build.
HIPCC ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
CSRC := leopard_amd/csrc
HDRS := $(wildcard $(CSRC)/*.h) include/leopard_amd.h
LIB := leopard_amd/libleopard_amd.so
EMULIB := tools/hipemu/libleopard_amd_emu.so

all: $(LIB)

$(LIB): $(CSRC)/capi.hip $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
	    -Wno-unused-value -o $@ $(CSRC)/capi.hip

emu: $(EMULIB)

$(EMULIB): $(CSRC)/capi.hip $(HDRS) tools/hipemu/hipemu.cpp tools/hipemu/hipemu.h
	$(HOSTCXX) -x c++ -DLMI_EMU -O1 -ffp-contract=off -std=c++17 -fPIC -shared -Itools/hipemu -I$(CSRC) -Wno-unused-value \
	    -o $@ $(CSRC)/capi.hip tools/hipemu/hipemu.cpp

clean:
	rm -f $(LIB) $(EMULIB)

.PHONY: all emu clean
