"""trace time of the freeform shapes at 1e6 rays (engine level, path mode)"""
import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import systems_zoo as zoo
from pyrate_amd import engine, systems, _lib
from pyrate_amd.surface_table import flatten_sequence
api = zoo.mirror_api()
dev = torch.device("cuda", 0)
cases = {"zernike fringe (25 terms)": api.build_simple_optical_system(zoo.zernike_builduplist("Fringe")),
         "asphere+zernike combination mirror": zoo.zernike_combination_system(api),
         "grid sag 25x21": zoo.gridsag_system(api),
         "xy polynomial (6 terms)": api.build_simple_optical_system(zoo.xypoly_builduplist())}
(o, k, e0) = systems.double_gauss_bundle(10**6, rpup=7.0, z0=-3.0, field_deg=1.0)
(x0, k0, e0d) = [engine.to_device_rays(a, dev) for a in (o, k, e0)]
for (label, (s, seq)) in cases.items():
    (recs, _) = flatten_sequence(s, seq, 0.5876e-3)
    sysd = engine.DeviceSystem(recs, 0)
    bufs = sysd.alloc_outputs(x0.shape[1], _lib.MODE_PATH)
    sysd.trace_timed(x0, k0, bufs, 10, e0d)
    ms = sysd.trace_timed(x0, k0, bufs, 30, e0d)
    v = sysd.views(bufs)
    print("%-38s S=%d  %.3f ms  %.2e ray-surface-ops/s  valid at image %.3f" %
          (label, len(recs), ms, x0.shape[1] * len(recs) / ms * 1e3, float(v.valid[-1].float().mean())))
