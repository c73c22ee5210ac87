# Builds libmcmc_hip.so (gfx950 only) in-tree, the host drivers, and the CPU
# oracle (test infrastructure).  CMakeLists.txt builds the same targets.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = mcmc_amd/csrc
OBJ = build/obj
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude \
           -DMCMC_BUILD -Wall -Wno-unused-function
LIB = mcmc_amd/libmcmc_hip.so

OBJS = $(OBJ)/chain.o $(OBJ)/chain_kind_0.o $(OBJ)/chain_kind_1.o $(OBJ)/chain_kind_2.o $(OBJ)/chain_kind_3.o \
       $(OBJ)/xsection.o $(OBJ)/sab_mode_1.o $(OBJ)/sab_mode_2.o $(OBJ)/sab_mode_3.o $(OBJ)/sab_mode_4.o \
       $(OBJ)/mvn.o $(OBJ)/capi.o $(OBJ)/multi.o $(OBJ)/errors.o
HDRS = $(wildcard $(CSRC)/*.hpp) include/mcmc_hip.h

all: $(LIB) drivers oracle probes

$(OBJ):
	mkdir -p $(OBJ)

# the chain kernels are built without SLP vectorisation: v_pk_* f32 ops issue at half rate on gfx950
# and the packing costs extra v_mov (measured 907 -> 967 G proposals/s); the S_AB kernel keeps it
$(OBJ)/chain_kind_%.o: $(CSRC)/chain_kind.hip $(HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -DMCMC_KIND=$* -c $< -o $@

# one translation unit per mode of the S_AB kernel
$(OBJ)/sab_mode_%.o: $(CSRC)/sab_mode.hip $(HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DMCMC_SAB_MODE=$* -c $< -o $@

$(OBJ)/mvn.o: $(CSRC)/mvn.hip $(HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -c $< -o $@

$(OBJ)/%.o: $(CSRC)/%.hip $(HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/%.o: $(CSRC)/%.cpp $(HDRS) | $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl -lpthread

# The same library with Philox4x32-7 (MCMC_PHILOX_ROUNDS=7): an optional performance build, not part of `all`
OBJ7 = build/obj7
LIB7 = mcmc_amd/libmcmc_hip_philox7.so
OBJS7 = $(patsubst $(OBJ)/%,$(OBJ7)/%,$(OBJS))
$(OBJ7):
	mkdir -p $(OBJ7)
$(OBJ7)/chain_kind_%.o: $(CSRC)/chain_kind.hip $(HDRS) | $(OBJ7)
	$(HIPCC) $(HIPFLAGS) -DMCMC_PHILOX_ROUNDS=7 -fno-slp-vectorize -DMCMC_KIND=$* -c $< -o $@
$(OBJ7)/sab_mode_%.o: $(CSRC)/sab_mode.hip $(HDRS) | $(OBJ7)
	$(HIPCC) $(HIPFLAGS) -DMCMC_PHILOX_ROUNDS=7 -DMCMC_SAB_MODE=$* -c $< -o $@
$(OBJ7)/mvn.o: $(CSRC)/mvn.hip $(HDRS) | $(OBJ7)
	$(HIPCC) $(HIPFLAGS) -DMCMC_PHILOX_ROUNDS=7 -fno-slp-vectorize -c $< -o $@
$(OBJ7)/%.o: $(CSRC)/%.hip $(HDRS) | $(OBJ7)
	$(HIPCC) $(HIPFLAGS) -DMCMC_PHILOX_ROUNDS=7 -c $< -o $@
$(OBJ7)/%.o: $(CSRC)/%.cpp $(HDRS) | $(OBJ7)
	$(HIPCC) $(HIPFLAGS) -DMCMC_PHILOX_ROUNDS=7 -c $< -o $@
$(LIB7): $(OBJS7)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS7) -ldl -lpthread
philox7: $(LIB7)

# Diagnostic builds of the S_AB kernel with shader-clock stamps around its phases (tools/gpu_sab_phases*.sh); the
# libraries go to tools/bin (they travel to the GPU box; delete them after use: 15 MB each)
PROF = build/prof
sabprof: $(LIB)
	mkdir -p $(PROF) tools/bin
	$(HIPCC) $(HIPFLAGS) -DMCMC_SAB_MODE=3 -DMCMC_SAB_PROFILE=1 -c $(CSRC)/sab_mode.hip -o $(PROF)/sab_mode_3.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o tools/bin/libmcmc_hip_sabprof.so $(filter-out $(OBJ)/sab_mode_3.o,$(OBJS)) $(PROF)/sab_mode_3.o
	for v in 1 2 3 4; do \
	  $(HIPCC) $(HIPFLAGS) -DMCMC_SAB_MODE=4 -DMCMC_SAB_PROFILE=$$v -c $(CSRC)/sab_mode.hip -o $(PROF)/sab_mode_4_$$v.o && \
	  $(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o tools/bin/libprof4_$$v.so $(filter-out $(OBJ)/sab_mode_4.o,$(OBJS)) $(PROF)/sab_mode_4_$$v.o || exit 1; \
	done

drivers: $(LIB)
	@if [ -f mcmc_amd/host/stepsize_main.cpp ]; then \
	  g++ -O3 -std=c++17 -Iinclude -Imcmc_amd/host -o mcmc_amd/host/stepsize mcmc_amd/host/stepsize_main.cpp \
	      -Lmcmc_amd -lmcmc_hip -Wl,-rpath,'$$ORIGIN/..' ; fi
	@if [ -f mcmc_amd/host/xsection_main.cpp ]; then \
	  g++ -O3 -std=c++17 -Iinclude -Imcmc_amd/host -o mcmc_amd/host/xsection mcmc_amd/host/xsection_main.cpp \
	      -Lmcmc_amd -lmcmc_hip -Wl,-rpath,'$$ORIGIN/..' ; fi

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB) $(LIB7) mcmc_amd/host/stepsize mcmc_amd/host/xsection
	$(MAKE) -C oracle clean
.PHONY: sabprof all drivers oracle probes philox7 clean

# device probe of include/mcmc_distributions.h (tests/test_distributions_header.py)
probes: tools/bin/dist_probe
tools/bin/dist_probe: tools/dist_probe.hip include/mcmc_distributions.h
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Iinclude -o $@ $<

# VALU issue-rate microbenchmark (calibrates the cycle model of tools/isa_mix.py)
ubench: tools/bin/ubench tools/bin/ubench2 tools/bin/ubench3 tools/bin/ubench_cal tools/bin/ubench_mfma tools/bin/ubench_store tools/bin/ubench_store_valu
tools/bin/%: tools/%.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -o $@ $<

# A/B builds of the lag kernel (tools/lag49_probe.py with MCMC_HIP_LIB=tools/bin/libmcmc_hip_lagab.so): only the r^2
# Woods-Saxon chain kernels are rebuilt, with LAGAB_FLAGS (e.g. -DMCMC_LAG_PHASES: prologue / epilogue cycles through tools/dbg/lag_phases.py)
LAGAB_FLAGS ?= -DMCMC_LAG_PHASES
lagab: $(LIB)
	mkdir -p build/lagab tools/bin
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -DMCMC_KIND=1 $(LAGAB_FLAGS) -c $(CSRC)/chain_kind.hip -o build/lagab/chain_kind_1.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o tools/bin/libmcmc_hip_lagab.so $(filter-out $(OBJ)/chain_kind_1.o,$(OBJS)) build/lagab/chain_kind_1.o -ldl -lpthread
.PHONY: lagab
