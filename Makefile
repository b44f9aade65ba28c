# Build of the MI355X ring engine (gfx950 only) and of the CPU oracle used by the tests.
HIPCC ?= hipcc
ARCH  ?= gfx950
PKG   := matrix-fhe-lattigo_amd
CSRC  := $(PKG)/csrc
LIB   := $(PKG)/lib/libringhip.so
SRCS  := $(CSRC)/engine.hip $(CSRC)/ntt3n.hip $(CSRC)/bext.hip $(CSRC)/rescale.hip $(CSRC)/keyswitch.hip $(CSRC)/kshard.hip $(CSRC)/automorphism.hip
HDRS  := $(wildcard $(CSRC)/*.hip.hpp) $(wildcard $(CSRC)/*.inc) $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)

ROCM ?= /opt/rocm
all: $(LIB) oracle tests/cpp/test_ring_cpp tests/cpp/test_sharded_host tests/cpp/libabort_trace.so

$(CSRC)/ntt_tile_asm.inc: tools/gen_tile_asm.py
	python3 tools/gen_tile_asm.py $@

$(LIB): $(SRCS) $(HDRS) $(CSRC)/ntt_tile_asm.inc
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -mllvm -pragma-unroll-threshold=131072 -fPIC -shared -Iinclude $(SRCS) -o $@

tests/cpp/test_ring_cpp: tests/cpp/test_ring_cpp.cpp tests/cpp/golden_vectors.inc include/ringhip.hpp include/ringhip.h $(LIB)
	g++ -O2 -std=c++17 -Iinclude $< -L$(PKG)/lib -lringhip -Wl,-rpath,'$$ORIGIN/../../$(PKG)/lib' -o $@

# a compiled host with threads as ranks and a plain-C all-gather callback: the limb-sharded key switch through the C ABI alone
tests/cpp/test_sharded_host: tests/cpp/test_sharded_host.cpp include/ringhip.h $(LIB)
	g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I$(ROCM)/include $< -L$(PKG)/lib -lringhip -L$(ROCM)/lib -lamdhip64 -lpthread -Wl,-rpath,'$$ORIGIN/../../$(PKG)/lib' -Wl,-rpath,$(ROCM)/lib -o $@

# test infrastructure: native call stack on SIGABRT (tests/conftest.py loads it when it is there)
tests/cpp/libabort_trace.so: tests/cpp/abort_trace.c
	gcc -O1 -g -fPIC -shared $< -o $@

oracle: oracle/libring_oracle.so
oracle/libring_oracle.so: oracle/ring_oracle.c oracle/ring_oracle.h include/ringhip_ops.h
	gcc -O3 -march=x86-64-v3 -fno-fast-math -ffp-contract=off -fPIC -shared -pthread oracle/ring_oracle.c -o $@

clean:
	rm -f $(LIB) oracle/libring_oracle.so

.PHONY: all oracle clean
