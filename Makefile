# Builds the product library (gfx950 only) and the CPU oracle (test infrastructure).
HIPCC ?= hipcc
ARCH ?= gfx950
# -ffp-contract=off: the arithmetic contract forbids fused multiply-add,
# so correspondence indices are bit-identical to the oracle's.
HIPFLAGS = --offload-arch=$(ARCH) -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc -Wall -Wno-unused-result
CSRC = pgslam_amd/csrc
LIB = pgslam_amd/lib/libpgicp.so
OBJS = $(CSRC)/kernels.o $(CSRC)/pgicp_api.o $(CSRC)/pgicp_comm.o

all: $(LIB) oracle

$(CSRC)/kernels.o: $(CSRC)/kernels.hip $(wildcard $(CSRC)/k_*.inc) $(CSRC)/kernels.hpp $(CSRC)/device_types.hpp $(CSRC)/icp_math.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/pgicp_api.o: $(CSRC)/pgicp_api.cpp $(wildcard $(CSRC)/api_*.inc) $(CSRC)/kernels.hpp $(CSRC)/device_types.hpp $(CSRC)/icp_math.hpp include/pgicp.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/pgicp_comm.o: $(CSRC)/pgicp_comm.cpp include/pgicp.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(OBJS)
	@mkdir -p pgslam_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl -pthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
