"""config 4 (anisotropic doublet, 1e6 rays): engine trace time, kernel variants via environment"""
import sys, time, math, torch, numpy as np
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
c = systems.CALCITE_TILTED
eps1 = systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"])
eps2 = systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))
for (label, recs) in (("uniaxial", systems.aniso_doublet_records(eps1, eps2)),
                      ("isoeps", systems.aniso_doublet_records()),
                      ("biaxial", systems.aniso_doublet_records(
                          np.array([[2.43, 0.03, -0.05], [0.03, 2.55, 0.04], [-0.05, 0.04, 2.78]]),
                          np.eye(3) * 1.6727 ** 2)),
                      ("biaxial2", systems.aniso_doublet_records(
                          np.array([[2.43, 0.03, -0.05], [0.03, 2.55, 0.04], [-0.05, 0.04, 2.78]]),
                          np.array([[2.75, -0.02, 0.03], [-0.02, 2.60, 0.05], [0.03, 0.05, 2.90]])))):
    sysd = engine.DeviceSystem(recs, 0)
    (o, k) = systems.collimated_bundle(10**6, 11.43, -5.0)
    e0 = np.cross(k, np.array([1., 0, 0]), axisa=0, axisb=0).T.copy()
    (x0, k0, e0d) = [engine.to_device_rays(a, dev, pitched=False) for a in (o, k, e0)]
    n = x0.shape[1]
    for mode in (_lib.MODE_PATH, _lib.MODE_IMAGE):
        bufs = sysd.alloc_outputs(n, mode)
        ms = sysd.trace_timed(x0, k0, bufs, 30, e0d)
        ms = sysd.trace_timed(x0, k0, bufs, 50, e0d)
        print("%-9s mode %d  %.4f ms" % (label, mode, ms))
