#!/usr/bin/env python
"""A/B of libprt builds (scratch/variants/libprt_<name>.so) on the crystal march of BASELINE configs[3]:
anisotropic doublet, 1e6 rays -> 4e6 at the image, uniaxial and biaxial crystals, path and image mode;
same arrays for every build (concatenated layout with the engine's ray pitch, inputs as arrays)."""
import ctypes
import glob
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from pyrate_amd import engine, systems, _lib

dev = torch.device("cuda", 0)
c = systems.CALCITE_TILTED
cases = {"uniaxial": (systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
                      systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))),
         "biaxial": (np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2]), np.diag([1.62 ** 2, 1.66 ** 2, 1.70 ** 2]))}
(o, k) = systems.collimated_bundle(1000000, 11.43, -5.0)
e0 = np.ascontiguousarray(np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T)
(x0, k0, e0d) = [engine.to_device_rays(a, dev, pitched=False) for a in (o, k, e0)]
n = o.shape[1]
st = engine._stream_handle(dev)
P = engine._ptr
paths = [None] + sorted(glob.glob(os.path.join(ROOT, "scratch", "variants", "libprt_*.so")))
out = {}
for (tag, (e1, e2)) in cases.items():
    sysd = engine.DeviceSystem(systems.aniso_doublet_records(e1, e2), 0)
    builds = []
    for path in paths:
        if path is None:
            builds.append(("in-tree", sysd.lib, sysd._h))
            continue
        lib = ctypes.CDLL(os.path.abspath(path))
        for fn in ("prt_system_create", "prt_trace_timed", "prt_system_destroy"):
            (res, args) = _lib.PROTOTYPES[fn]
            getattr(lib, fn).restype = res
            getattr(lib, fn).argtypes = args
        h = ctypes.c_void_p()
        assert lib.prt_system_create(sysd._table, sysd.n_surfaces, 0, ctypes.byref(h)) == 0
        builds.append((os.path.basename(path)[7:-3], lib, h))
    for (mname, mode) in (("path", _lib.MODE_PATH), ("image", _lib.MODE_IMAGE)):
        bufs = sysd.alloc_outputs(n, mode)

        def timed(lib, h, iters):
            ms = ctypes.c_double()
            rc = lib.prt_trace_timed(h, n, 0, P(x0), P(k0), P(e0d), None, mode, bufs["pitch"], P(bufs["x_hit"]),
                                     P(bufs["k_out"]), P(bufs["valid"]), P(bufs["valid_out"]), st, iters,
                                     ctypes.byref(ms))
            assert rc == 0, rc
            return ms.value
        timed(builds[0][1], builds[0][2], 100)          # clocks up before anything is compared
        for rep in range(5):                            # builds interleaved: drifts hit all of them alike
            for (name, lib, h) in builds:
                out.setdefault("%s_%s_%s" % (tag, mname, name), []).append(round(timed(lib, h, 20), 4))
print(json.dumps(out))
