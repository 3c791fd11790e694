// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): the sources of libprt, unmodified, compiled for the
// host against the stand-in runtime.  Built by tests/hostemu/build.py into tests/hostemu/_build/ (git-ignored).
#include "hip/hip_runtime.h"
#include "../../pyrate_amd/csrc/prt.hip"
// the dynamic LDS of the crystal march (`extern __shared__ double park_lds[]`): room for the largest launch
// (PRT_PARK_LDS_LEVELS levels x PRT_GENERAL_BLOCK threads x (9 doubles + 1 byte))
thread_local double park_lds[(PRT_PARK_LDS_LEVELS * PRT_GENERAL_BLOCK * (9 * 8 + 1) + 7) / 8 + 64];
