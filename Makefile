# Builds the C-ABI library (hand-written HIP for gfx950) and nothing else.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mpntrackseg_amd/csrc
SRCS := $(CSRC)/gemm.hip $(CSRC)/gemm_tn.hip $(CSRC)/edge_chain.hip $(CSRC)/graph_prep.hip $(CSRC)/segment.hip $(CSRC)/mpn.hip $(CSRC)/backward.hip $(CSRC)/loss.hip $(CSRC)/attention.hip $(CSRC)/graph_build.hip $(CSRC)/tracker.hip
OBJS := $(SRCS:.hip=.o)
LIB := $(CSRC)/libmpnhip.so
# EXTRA=-DMPNHIP_CHAIN_TS builds the fused chain kernels with per-phase cycle stamps (tools/chain_stamps.py)
CXXFLAGS := -O3 -fPIC -std=c++17 --offload-arch=$(ARCH) -Wall -Wno-unused-function $(EXTRA)

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/plan.h $(CSRC)/edge_chain.h include/mpnhip.h
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
