# Build of the MI355X-native V-PCC reconstruction path.
#   make            -> product library  tmc2-rs_amd/libvpcc_recon.so (hipcc, gfx950)
#                      + test oracle     oracle/libvpcc_oracle.so     (gcc, plain C)
#   make product / make oracle / make clean
# Built artefacts are git-ignored but travel to the GPU box with the gpurun snapshot.

HIPCC      ?= /opt/rocm/bin/hipcc
CC         ?= gcc
ARCH       ?= gfx950
PROJ       := tmc2-rs_amd
CSRC       := $(PROJ)/csrc

HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
              -Iinclude -I$(CSRC) $(EXTRA)
PRODUCT_SO := $(PROJ)/libvpcc_recon.so
PRODUCT_SRC := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
PRODUCT_HDR := $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.h) include/vpcc_recon.h

ORACLE_SO  := oracle/libvpcc_oracle.so
ORACLE_SRC := oracle/vpcc_oracle.c oracle/vpcc_smoothing_spec.c
ORACLE_HDR := oracle/vpcc_oracle.h oracle/vpcc_smoothing_spec.h include/vpcc_recon.h

all: product oracle
product: $(PRODUCT_SO)
oracle: $(ORACLE_SO)

$(PRODUCT_SO): $(PRODUCT_SRC) $(PRODUCT_HDR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(PRODUCT_SRC) -lpthread

# -ffp-contract=off: the reference's f64 colour maths is never fused (rustc does not contract)
$(ORACLE_SO): $(ORACLE_SRC) $(ORACLE_HDR)
	$(CC) -O2 -std=c99 -fPIC -shared -ffp-contract=off -Wall -Wextra -o $@ $(ORACLE_SRC) -lm

clean:
	rm -f $(PRODUCT_SO) $(ORACLE_SO) $(PROJ)/*.o

.PHONY: all product oracle clean
