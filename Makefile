# Build of the MI355X curve-number generator.
#   make            everything that builds without a GPU (hipcc cross-compiles gfx950)
#   make gpu        libgcn10_gpu.so   (HIP kernels + C ABI, include/gcn10_gpu.h)
#   make host       libgcn10_host.so  (C99 host library, include/gcn10_host.h)
#   make cli        bin/gcn10         (drop-in for the reference's src/ program)
#   make oracle     oracle/libcn_oracle.so (test infrastructure only)
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := gcn10_amd
CSRC    := $(PKG)/csrc
HOSTSRC := $(wildcard $(CSRC)/host/*.c)
HOSTLIBSRC := $(filter-out $(CSRC)/host/main.c,$(HOSTSRC))

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall
# -ffp-contract=off: the fp64 index arithmetic must match the reference's
# non-FMA build bit for bit (src/CMakeLists.txt:68 has no -march).
CFLAGS  := -std=c99 -D_GNU_SOURCE -O2 -g -Wall -Wextra -fPIC -ffp-contract=off -Iinclude -pthread

all: gpu host cli oracle

gpu: $(PKG)/libgcn10_gpu.so
host: $(PKG)/libgcn10_host.so
oracle:
	$(MAKE) -C oracle

GPUSRC := $(wildcard $(CSRC)/*.hip)
$(PKG)/libgcn10_gpu.so: $(GPUSRC) $(wildcard $(CSRC)/*.hpp) include/gcn10_gpu.h
	$(HIPCC) $(HIPFLAGS) -I$(CSRC) -shared -o $@ $(GPUSRC)

$(PKG)/libgcn10_host.so: $(HOSTLIBSRC) $(wildcard $(CSRC)/host/*.h) include/gcn10_host.h include/gcn10_gpu.h
	$(CC) $(CFLAGS) -shared -o $@ $(HOSTLIBSRC) -lm -lz -ldl

ifneq ($(wildcard $(CSRC)/host/main.c),)
cli: bin/gcn10
bin/gcn10: $(CSRC)/host/main.c $(PKG)/libgcn10_host.so
	mkdir -p bin
	$(CC) $(CFLAGS) -o $@ $(CSRC)/host/main.c -L$(PKG) -lgcn10_host -Wl,-rpath,'$$ORIGIN/../$(PKG)' -lm -lz -ldl
else
cli:
endif

# host library under AddressSanitizer + UBSan, and the host test files run against it
asan-host:
	$(CC) -std=c99 -D_GNU_SOURCE -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined \
	    -fno-sanitize-recover=undefined -fPIC -ffp-contract=off -Iinclude -pthread -shared \
	    -o /tmp/libgcn10_host_asan.so $(HOSTLIBSRC) -lm -lz -ldl
	LD_PRELOAD="$$($(CC) -print-file-name=libasan.so) $$($(CC) -print-file-name=libubsan.so)" \
	    ASAN_OPTIONS=detect_leaks=0 GCN10_HOST_LIB=/tmp/libgcn10_host_asan.so \
	    python -m pytest tests/test_host.py tests/test_host_io.py tests/test_host_fuzz.py -q

# the threaded host program (block workers, strip hand-over, sink pool) under ThreadSanitizer, against the
# asynchronous host stub of the GPU library (tests/stub_gpu: test infrastructure, never a fallback);
# 3 workers x 8 blocks through gcn10_run in three sink modes, rasters compared with the oracle
tsan-host: oracle host
	python3 tests/tsan_host.py profiles/r03/tsan_host.log

clean:
	rm -f $(PKG)/*.so bin/gcn10
	$(MAKE) -C oracle clean
.PHONY: all gpu host cli oracle clean asan-host tsan-host
