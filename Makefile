# Build of the MI355X-native D2Q9-BGK path.
#   make            liblbm_hip.so (HIP, gfx950) + d2q9-bgk / d2q9-bgk.exe (thin C host) + oracle
#   make check      the reference's `make check` contract (reference Makefile:16-19,26-27)
PKG      = opencl-lattice-boltzmann_amd
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
HIPFLAGS = -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -I/opt/rocm/include
LIB      = $(PKG)/liblbm_hip.so
EXE      = d2q9-bgk

FINAL_STATE_FILE=./final_state.dat
AV_VELS_FILE=./av_vels.dat
REF_FINAL_STATE_FILE=check/128x128.final_state.dat
REF_AV_VELS_FILE=check/128x128.av_vels.dat

all: $(LIB) $(EXE) $(EXE).exe oracle

# lbm_version() carries a digest of the device + host sources the library was built from: a committed profile
# (profiles/traffic.json) names the build it measured, and bench.py refuses its counters for any other build
CSRC   = $(PKG)/csrc/d2q9_kernels.h $(PKG)/csrc/deep_instances.h $(PKG)/csrc/halo_exchange.h $(PKG)/csrc/lbm_hip.cpp $(PKG)/csrc/lbm_deep.cpp
SRC_ID = $(shell cat $(CSRC) | sha256sum | cut -c1-12)

# Two translation units: the deep window kernels get the compiler's max-ILP scheduling strategy (csrc/deep_instances.h says
# why), everything else the default one.  The objects are build products next to the library (*.o is git-ignored).
$(PKG)/csrc/lbm_hip.o: $(CSRC) include/lbm.h
	$(HIPCC) $(HIPFLAGS) -DLBM_SRC_ID=\"$(SRC_ID)\" -c $(PKG)/csrc/lbm_hip.cpp -o $@

$(PKG)/csrc/lbm_deep.o: $(PKG)/csrc/lbm_deep.cpp $(PKG)/csrc/deep_instances.h $(PKG)/csrc/d2q9_kernels.h
	$(HIPCC) $(HIPFLAGS) -mllvm -amdgpu-sched-strategy=max-ilp -c $(PKG)/csrc/lbm_deep.cpp -o $@

$(LIB): $(PKG)/csrc/lbm_hip.o $(PKG)/csrc/lbm_deep.o
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -Wl,-z,defs $^ -o $@ -ldl   # -z defs: a launch of a deep kernel instance that LBM_DEEP_INSTANCES lacks fails HERE

$(EXE): $(PKG)/host/d2q9-bgk.c include/lbm.h $(LIB)
	$(CC) -std=c99 -O2 -Wall -D_GNU_SOURCE -Iinclude $(PKG)/host/d2q9-bgk.c -o $@ -L$(PKG) -llbm_hip -lm -lpthread -Wl,-rpath,'$$ORIGIN/$(PKG)'

$(EXE).exe: $(EXE)
	cp $(EXE) $(EXE).exe

oracle:
	$(MAKE) -C oracle

# golden files are kept gzip-compressed under tests/golden/check; unpack them where the
# reference keeps them (check/<size>.<name>.dat)
check/%.dat: tests/golden/check/%.dat.gz
	gzip -dc $< > $@

check: $(REF_AV_VELS_FILE) $(REF_FINAL_STATE_FILE)
	python check/check.py --ref-av-vels-file=$(REF_AV_VELS_FILE) --ref-final-state-file=$(REF_FINAL_STATE_FILE) --av-vels-file=$(AV_VELS_FILE) --final-state-file=$(FINAL_STATE_FILE)

# Sanitizer builds of everything that runs on the CPU (the reference had only a dead -DDEBUG target,
# Makefile_old:30-31).  The C host is linked against a stand-in for the HIP library (tests/cpu/lbm_stub.c: the ABI of
# include/lbm.h without numerics) so that its parsers and writers run under AddressSanitizer + UBSan on a CPU-only box;
# the oracle's serial drivers get the same flags.  tests/test_host_cli.py and tests/test_sanitizers.py run them.
SAN = -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
asan: $(EXE)-asan oracle-asan

$(EXE)-asan: $(PKG)/host/d2q9-bgk.c tests/cpu/lbm_stub.c include/lbm.h
	$(CC) -std=c99 $(SAN) -Wall -Wextra -D_GNU_SOURCE -Iinclude $(PKG)/host/d2q9-bgk.c tests/cpu/lbm_stub.c -o $@ -lm -lpthread

oracle-asan:
	$(MAKE) -C oracle asan

# stand-alone measurement programs (not part of the product): tools/barrier_probe (grid barriers against launch boundaries,
# profiles/r03_persistent_kernel_negative.txt), tools/deep_probe (the harness the deep window kernel was built in)
probes: tools/barrier_probe tools/deep_probe tools/resident_probe
tools/%: tools/%.cpp
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -I/opt/rocm/include $< -o $@

clean:
	rm -f $(LIB) $(PKG)/csrc/*.o $(EXE) $(EXE).exe $(EXE)-asan
	$(MAKE) -C oracle clean

.PHONY: all check clean oracle asan oracle-asan probes
