// TEST INFRASTRUCTURE ONLY (tests/hostsim): the device-only hooks of include/gpk_testhooks.h (hardware probes: MFMA issue
// loops, LDS banking, CU placement, kernel timings) do not exist in the host simulation; the shared-library form of it
// still has to export every declared symbol, because the ctypes binding resolves the whole table when it loads.
#include "../../include/gpk_testhooks.h"
#include "gpk_internal.h"

#define HOSTSIM_ABSENT(sig)                                                             \
  extern "C" int sig { return gpk_set_err(-1, "not available in the host simulation"); }

HOSTSIM_ABSENT(gpk_bench_clock_under_load(int, int, int, int, double*, double*))
HOSTSIM_ABSENT(gpk_bench_chain_step(int, int, int, double*, double*))
HOSTSIM_ABSENT(gpk_bench_se_build(int, int, int, int, int, double*))
HOSTSIM_ABSENT(gpk_bench_gemm_nt(int, int, int, int, int, int, double*))
HOSTSIM_ABSENT(gpk_bench_lds_bank_probe(int, int))
HOSTSIM_ABSENT(gpk_bench_leaf(int, int, double*))
HOSTSIM_ABSENT(gpk_bench_leaf_contended(int, double*, double*))
HOSTSIM_ABSENT(gpk_bench_mfma_pinned(int, int, double*))
HOSTSIM_ABSENT(gpk_bench_streams(int, int, int, int, int, int, int, double*))
