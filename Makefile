# Build of the MI355X-native V-PCC reconstruction path.
#   make            -> product library  tmc2-rs_amd/libvpcc_recon.so (hipcc, gfx950)
#                      + test oracle     oracle/libvpcc_oracle.so     (gcc, plain C)
#   make product / make oracle / make clean        (objects under build/, `make -j` works)
#   make lds-dma    -> tmc2-rs_amd/libvpcc_recon_ldsdma.so: the library with the EXPERIMENTAL tile kernel of round 4
#                      (tools/experiments/vpcc_tiles_lds_dma.hip: attribute tiles staged in LDS by LDS-DMA; correct, 13 % fewer
#                      plane reads, 7 % slower — DESIGN.md 4.1.2).  Loaded only by tools/ab_prebuilt.sh.
#   make diag       -> tmc2-rs_amd/libvpcc_recon_diag.so: the same objects, with the tile kernel compiled
#                      -DVPCC_DIAGNOSTIC (run-time ablation switches and in-kernel stamps).  Loaded only when
#                      a tools/ script sets VPCC_DIAG_LIB=1; tests, bench.py and the product never use it.
# Built artefacts are git-ignored but travel to the GPU box with the gpurun snapshot.

HIPCC      ?= /opt/rocm/bin/hipcc
CC         ?= gcc
ARCH       ?= gfx950
PROJ       := tmc2-rs_amd
CSRC       := $(PROJ)/csrc
OBJ        := build/obj

# -amdgpu-atomic-optimizer-strategy=None: the optimizer turns the one-lane ticket fetch-add of the tile kernel into
# "aggregate over the wave, then read the result back at once" — a vmcnt(0) right behind the atomic, which is
# exactly the round trip the kernel issues a step ahead to hide.  No kernel here has a many-lane atomic to gain.
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
              -mllvm -amdgpu-atomic-optimizer-strategy=None -Iinclude -I$(CSRC) $(EXTRA)
PRODUCT_SO := $(PROJ)/libvpcc_recon.so
DIAG_SO    := $(PROJ)/libvpcc_recon_diag.so
HIP_SRC    := $(wildcard $(CSRC)/*.hip)
CPP_SRC    := $(wildcard $(CSRC)/*.cpp)
PRODUCT_OBJ := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.hip.o,$(HIP_SRC)) $(patsubst $(CSRC)/%.cpp,$(OBJ)/%.cpp.o,$(CPP_SRC))
DIAG_OBJ   := $(filter-out $(OBJ)/vpcc_tiles.hip.o,$(PRODUCT_OBJ)) $(OBJ)/vpcc_tiles.diag.o
LDSDMA_SO  := $(PROJ)/libvpcc_recon_ldsdma.so
LDSDMA_OBJ := $(filter-out $(OBJ)/vpcc_tiles.hip.o $(OBJ)/vpcc_host.cpp.o,$(PRODUCT_OBJ)) $(OBJ)/vpcc_tiles.ldsdma.o $(OBJ)/vpcc_host.ldsdma.o
PRODUCT_HDR := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.h) include/vpcc_recon.h

ORACLE_SO  := oracle/libvpcc_oracle.so
ORACLE_SRC := oracle/vpcc_oracle.c oracle/vpcc_smoothing_spec.c
ORACLE_HDR := oracle/vpcc_oracle.h oracle/vpcc_smoothing_spec.h include/vpcc_recon.h

all: product oracle
product: $(PRODUCT_SO)
diag: $(DIAG_SO)
lds-dma: $(LDSDMA_SO)
oracle: $(ORACLE_SO)

$(OBJ):
	mkdir -p $(OBJ)

# EXTRA changes the objects: remember it
$(OBJ)/flags.$(shell echo '$(HIPFLAGS)' | md5sum | cut -c1-12): | $(OBJ)
	rm -f $(OBJ)/flags.* $(OBJ)/*.o
	touch $@
FLAGS_STAMP := $(OBJ)/flags.$(shell echo '$(HIPFLAGS)' | md5sum | cut -c1-12)

$(OBJ)/%.hip.o: $(CSRC)/%.hip $(PRODUCT_HDR) $(FLAGS_STAMP)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(OBJ)/%.cpp.o: $(CSRC)/%.cpp $(PRODUCT_HDR) $(FLAGS_STAMP)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(OBJ)/vpcc_tiles.diag.o: $(CSRC)/vpcc_tiles.hip $(PRODUCT_HDR) $(FLAGS_STAMP)
	$(HIPCC) $(HIPFLAGS) -DVPCC_DIAGNOSTIC -c -o $@ $<

$(OBJ)/vpcc_tiles.ldsdma.o: tools/experiments/vpcc_tiles_lds_dma.hip $(PRODUCT_HDR) $(FLAGS_STAMP)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(OBJ)/vpcc_host.ldsdma.o: $(CSRC)/vpcc_host.cpp $(PRODUCT_HDR) $(FLAGS_STAMP)
	$(HIPCC) $(HIPFLAGS) -DVPCC_LDS_STAGED_ATTRIBUTES -c -o $@ $<
$(LDSDMA_SO): $(LDSDMA_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(LDSDMA_OBJ) -lpthread

$(PRODUCT_SO): $(PRODUCT_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(PRODUCT_OBJ) -lpthread
$(DIAG_SO): $(DIAG_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(DIAG_OBJ) -lpthread

# -ffp-contract=off: the reference's f64 colour maths is never fused (rustc does not contract)
$(ORACLE_SO): $(ORACLE_SRC) $(ORACLE_HDR)
	$(CC) -O2 -std=c99 -fPIC -shared -ffp-contract=off -Wall -Wextra -o $@ $(ORACLE_SRC) -lm

clean:
	rm -rf $(PRODUCT_SO) $(DIAG_SO) $(LDSDMA_SO) $(ORACLE_SO) build/obj

.PHONY: all product diag lds-dma oracle clean
