#!/usr/bin/env python
"""A/B of libprt builds (scratch/variants/libprt_<name>.so, same ABI) against the in-tree library on the bench
configurations, the way bench.py runs them: uniform first segment (prt_trace_ex), arena-placed arrays -- the SAME
arrays for every build, builds interleaved, path and image mode.

    python benchmarks/ab_builds.py [doublegauss asphere xypoly aniso] > profiles/<tag>_ab_builds.json
"""
import ctypes
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from benchmarks import workloads as bench
from pyrate_amd import engine, _lib

dev = torch.device("cuda", 0)
configs = [a for a in sys.argv[1:] if not a.startswith("-")] or ["doublegauss", "asphere", "xypoly", "aniso"]
paths = sorted(glob.glob(os.path.join(ROOT, "scratch", "variants", "libprt_*.so")))
out = {}
for config in configs:
    wl = bench.make_workload(config, 1_000_000 if config == "aniso" else 10_000_000, dev)
    sysd = engine.DeviceSystem(wl["records"], 0)
    builds = [("in-tree", sysd.lib, sysd._h)]
    for path in paths:
        lib = ctypes.CDLL(os.path.abspath(path))
        for fn in ("prt_system_create", "prt_trace_ex", "prt_system_destroy"):
            (res, args) = _lib.PROTOTYPES[fn]
            getattr(lib, fn).restype = res
            getattr(lib, fn).argtypes = args
        h = ctypes.c_void_p()
        assert lib.prt_system_create(sysd._table, sysd.n_surfaces, 0, ctypes.byref(h)) == 0
        builds.append((os.path.basename(path)[7:-3], lib, h))
    n = wl["n_local"]
    for (mname, mode) in (("path", _lib.MODE_PATH), ("image", _lib.MODE_IMAGE)):
        bufs = sysd.alloc_outputs(n, mode, packed_flags=sysd.all_isotropic)

        def timed(lib, h, iters):
            ms = ctypes.c_double()
            a = sysd._trace_args(wl["x0"], wl["k0"], bufs, wl["e0"], uniform=wl["uniform"])
            a.timed_iters = iters
            a.ms_avg = ctypes.pointer(ms)
            rc = lib.prt_trace_ex(h, ctypes.byref(a))
            assert rc == 0, rc
            return ms.value
        timed(builds[0][1], builds[0][2], 100)          # clocks up before anything is compared
        for rep in range(5):                            # builds interleaved: drifts hit all of them alike
            for (name, lib, h) in builds:
                out.setdefault("%s_%s_ms" % (config, mname), {}).setdefault(name, []).append(round(timed(lib, h, 20), 4))
        # do the builds agree on the last surface's hit points?
        ref = None
        for (name, lib, h) in builds:
            timed(lib, h, 1)
            torch.cuda.synchronize()
            cur = sysd.views(bufs).x_hit[-1].clone()
            if ref is None:
                ref = cur
            else:
                d = (torch.nan_to_num(cur) - torch.nan_to_num(ref)).abs().max().item()
                out.setdefault("%s_%s_max_abs_difference_of_last_hit_points_to_in_tree" % (config, mname), {})[name] = d
        del bufs
print(json.dumps(out))
