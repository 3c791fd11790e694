"""device memory over 3000 merit evaluations with a changing system (table create / destroy, lazy
bundles, compaction scratch, moments workspace): must stay flat"""
import sys, torch
sys.path.insert(0, '.')
import numpy as np
from pyrate_amd.builders import build_simple_optical_system
from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.ray import RayBundle
(s, seq) = build_simple_optical_system([
    ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
    ({"shape": "Conic"}, {"decz": 5.0}, 1.5168, "front", {}),
    ({"shape": "Asphere", "curv": -1. / 50., "cc": -1., "coefficients": [0.0, 0.0, 0.0]}, {"decz": 20.0}, None, "back", {}),
    ({"shape": "Conic"}, {"decz": 100.0}, None, "image", {})])
osa = OpticalSystemAnalysis(s, seq)
(o, k, e0) = osa.collimated_bundle(2000, {"startz": -5., "radius": 11.43}, wave=0.5876e-3)
b = RayBundle(x0=o, k0=k, Efield0=e0, wave=0.5876e-3)
params = s.elements["stdelem"].surfaces["back"].shape.params
def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20
f0 = None
for it in range(3001):
    params["curv"].set_value(-1. / 50. + 1e-7 * it)
    rp = s.seqtrace(b, seq)[0]
    x = rp.raybundles[-1].x
    (m, _) = s.image_moments(b, seq)
    if it in (100, 1000, 2000, 3000):
        f = free()
        if f0 is None: f0 = f
        print("iteration %4d: free device memory %.1f MiB (%.1f vs iteration 100), torch allocated %.1f MiB"
              % (it, f, f - f0, torch.cuda.memory_allocated() / 2 ** 20))
