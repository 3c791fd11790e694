"""A/B of libprt builds on the crystal doublet (k_trace_general), same arrays, one process"""
import ctypes, glob, math, os, sys, torch
import numpy as np
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
c = systems.CALCITE_TILTED
for (tag, e1, e2) in (("uniaxial", systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]), systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))),
                      ("biaxial", np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2]), np.diag([1.62 ** 2, 1.66 ** 2, 1.70 ** 2]))):
    sysd = engine.DeviceSystem(systems.aniso_doublet_records(e1, e2), 0)
    (o, k) = systems.collimated_bundle(int(1e6), 11.43, -5.0)
    e0 = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T.copy()
    (x0, k0, e0d) = [engine.to_device_rays(a, dev) for a in (o, k, e0)]
    n = x0.shape[1]
    bufs = sysd.alloc_outputs(n, _lib.MODE_PATH)
    libs = {"in-tree": (sysd.lib, sysd._h)}
    for path in sorted(glob.glob("scratch/variants/libprt_*.so")):
        lib = ctypes.CDLL(os.path.abspath(path))
        for name in ("prt_system_create", "prt_trace_timed", "prt_system_destroy"):
            (res, args) = _lib.PROTOTYPES[name]
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
        h = ctypes.c_void_p()
        assert lib.prt_system_create(sysd._table, sysd.n_surfaces, 0, ctypes.byref(h)) == 0
        libs[os.path.basename(path)[7:-3]] = (lib, h)
    st = engine._stream_handle(dev)
    P = engine._ptr
    def timed(lib, h, iters):
        ms = ctypes.c_double()
        rc = lib.prt_trace_timed(h, n, 0, P(x0), P(k0), P(e0d), None, _lib.MODE_PATH, 0, P(bufs["x_hit"]),
                                 P(bufs["k_out"]), P(bufs["valid"]), P(bufs["valid_out"]), st, iters, ctypes.byref(ms))
        assert rc == 0, rc
        return ms.value
    timed(*libs["in-tree"], 20)
    for rep in range(3):
        print(tag + " " + "  ".join("%s %.4f" % (name, (timed(lib, h, 2), timed(lib, h, 20))[1]) for (name, (lib, h)) in libs.items()), flush=True)
