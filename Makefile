# Builds the C-ABI library (hand-written HIP for gfx950) and nothing else.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mpntrackseg_amd/csrc
SRCS := $(CSRC)/gemm.hip $(CSRC)/gemm_bf16.hip $(CSRC)/gemm_tn.hip $(CSRC)/wgrad_panel.hip $(CSRC)/wgrad_rows16.hip $(CSRC)/edge_chain.hip $(CSRC)/edge_chain_bf16.hip $(CSRC)/edge_chain_bf16_bwd.hip $(CSRC)/graph_prep.hip $(CSRC)/segment.hip $(CSRC)/node_chain.hip $(CSRC)/persist32.hip $(CSRC)/mpn.hip $(CSRC)/backward.hip $(CSRC)/loss.hip $(CSRC)/bn_dropout.hip $(CSRC)/attention.hip $(CSRC)/graph_build.hip $(CSRC)/tracker.hip
OBJS := $(SRCS:.hip=.o)
LIB := $(CSRC)/libmpnhip.so
# EXTRA=-DMPNHIP_CHAIN_TS builds the fused chain kernels with per-phase cycle stamps (tools/chain_stamps.py)
CXXFLAGS := -O3 -fPIC -std=c++17 --offload-arch=$(ARCH) -Wall -Wno-unused-function $(EXTRA)

# torch.ops.mpnhip.* : a C++ shim over the C ABI registering the operators with the PyTorch dispatcher (TORCH_LIBRARY); plain g++
# against the torch headers -- no kernel in it, nothing hipified
TORCH_DIR := $(shell python3 -c "import os, torch; print(os.path.dirname(torch.__file__))" 2>/dev/null)
TORCH_LIB := $(CSRC)/libmpnhip_torch.so
TORCH_ABI := $(shell python3 -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))" 2>/dev/null)

all: $(LIB) $(TORCH_LIB)

$(TORCH_LIB): $(CSRC)/torch_ops.cpp include/mpnhip.h $(LIB)
	g++ -O2 -fPIC -std=c++17 -shared -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=$(TORCH_ABI) \
	    -I$(TORCH_DIR)/include -I$(TORCH_DIR)/include/torch/csrc/api/include -I/opt/rocm/include \
	    $< -o $@ -L$(CSRC) -lmpnhip -L$(TORCH_DIR)/lib -lc10 -lc10_hip -ltorch_cpu -ltorch_hip -ltorch -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(TORCH_DIR)/lib

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/plan.h $(CSRC)/edge_chain.h $(CSRC)/edge_chain_bf16_common.h $(CSRC)/row_stage.h include/mpnhip.h
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# stand-alone micro-benchmarks / diagnostics (tools/micro/*.hip); outputs go to build/micro/ (git-ignored)
MICRO := $(patsubst tools/micro/%.hip,build/micro/%,$(wildcard tools/micro/*.hip))
micro: $(MICRO)
build/micro/%: tools/micro/%.hip
	@mkdir -p build/micro
	$(HIPCC) -O3 --offload-arch=$(ARCH) $< -o $@

clean:
	rm -f $(OBJS) $(LIB) $(TORCH_LIB)
	rm -rf build/micro

.PHONY: all clean micro
