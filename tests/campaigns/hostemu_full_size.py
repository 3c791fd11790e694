"""One-off campaign (not collected by pytest): the FULL-SIZE bodies of the `-m gpu` suite -- BASELINE's 1e7-ray double
Gauss over 12 surfaces (hit points on their surfaces, |k| = n, masks), the reversed trace, 1e6 rays through the crystal
doublets (wave-equation residual of every solution) -- on the HOST build of the kernels' sources WITH AddressSanitizer
and UBSan (tests/hostemu): every index computation of the marches at the sizes bench.py runs them, on exact-size arrays.

    LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) PRT_HOSTEMU_LIBRARY=tests/hostemu/_build/libprt_hostemu_san.so \
        ASAN_OPTIONS=detect_leaks=0 python tests/campaigns/hostemu_full_size.py        (about 20 minutes on 1 core)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch                                 # noqa: E402
import hostemu                               # noqa: E402,F401
from hostemu import adapter                  # noqa: E402
from pyrate_amd import engine                # noqa: E402

engine.DeviceSystem = adapter.HostDeviceSystem          # this process only: the bodies below ask the engine for these two
engine.to_device_rays = adapter.to_device_rays
torch.cuda.synchronize = lambda *a, **k: None
HOST = torch.device("cpu")

import test_gpu_parity as P                  # noqa: E402

print("library:", os.environ.get("PRT_HOSTEMU_LIBRARY") or hostemu.build(), flush=True)
for (fn, kw) in ((P.test_full_size_properties, {}), (P.test_double_gauss_trace_is_reversible_at_full_size, {}),
                 (P.test_double_gauss_trace_scales_exactly_with_powers_of_two, {}),
                 (P.test_crystal_solutions_satisfy_the_wave_equation_at_full_size, {"kind": "uniaxial"}),
                 (P.test_crystal_solutions_satisfy_the_wave_equation_at_full_size, {"kind": "biaxial"}),
                 (P.test_sharded_crystal_trace_reassembles_to_the_whole_bundle, {})):
    t0 = time.time()
    fn(gpu_device=HOST, **kw)
    print("%s %s: passed, %.0f s" % (fn.__name__, kw or "", time.time() - t0), flush=True)
print("done")
