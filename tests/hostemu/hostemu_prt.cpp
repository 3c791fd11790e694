// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): the sources of libprt, unmodified, compiled for the
// host against the stand-in runtime.  Built by tests/hostemu/__init__.py into tests/hostemu/_build/ (git-ignored).
#include "hip/hip_runtime.h"
// The product must never run on this build: pyrate_amd/_lib.py checks prt_abi_version() of whatever it loads (also through
// its PRT_LIBRARY switch for A/B builds) -- the host build answers 1000 + the sources' version, which no package accepts.
#define prt_abi_version prt_abi_version_of_the_sources
#include "../../pyrate_amd/csrc/prt.hip"
#undef prt_abi_version
extern "C" int32_t prt_abi_version(void) { return 1000 + prt_abi_version_of_the_sources(); }
// the dynamic LDS of the crystal march (`extern __shared__ double park_lds[]`): room for the largest launch
// (PRT_PARK_LDS_LEVELS levels x PRT_GENERAL_BLOCK threads x (9 doubles + 1 byte))
thread_local double park_lds[(PRT_PARK_LDS_LEVELS * PRT_GENERAL_BLOCK * (9 * 8 + 1) + 7) / 8 + 64];
