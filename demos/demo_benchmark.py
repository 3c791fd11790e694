#!/usr/bin/env python
"""The reference's own benchmark scenario (demos/demo_benchmark.py:47-85): the 8-surface
n=1.7/1.5 system, a divergent bundle of 1e5 rays, wall-clock around seqtrace only, and the
reference's "ray-surface-operations per second" formula."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import time

import torch

from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.globalconstants import degree, standard_wavelength
from pyrate_amd.raytracer.ray import RayBundle
from pyrate_amd.sampling2d import raster


def main(nrays=100000):
    (s, seq) = build_rotationally_symmetric_optical_system(
        [(-5.922, 0, 2.0, 1.7, "surf1", {}),
         (-3.160, 0, 3.0, None, "surf2", {}),
         (15.884, 0, 5.0, 1.7, "surf3", {}),
         (-12.756, 0, 3.0, None, "surf4", {}),
         (0, 0, 3.0, None, "stop", {"is_stop": True}),
         (3.125, 0, 2.0, 1.5, "surf5", {}),
         (1.479, 0, 3.0, None, "surf6", {}),
         (0, 0, 19.0, None, "surf7", {})])
    osa = OpticalSystemAnalysis(s, seq, name="Analysis")
    (x0, k0, e0) = osa.divergent_bundle(nrays, {"radius": 10. * degree, "raster": raster.RectGrid()})
    bundle = RayBundle(x0=x0, k0=k0, Efield0=e0, wave=standard_wavelength)
    s.seqtrace(bundle, seq)                       # warm-up (table upload, allocator)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    raypath = s.seqtrace(bundle, seq)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    nsurf = len(s.elements["stdelem"].surfaces)
    print("benchmark : %.6f s for tracing %d rays through %d surfaces." % (t2 - t1, x0.shape[1], nsurf))
    print("That is %d ray-surface-operations per second" % int(round(x0.shape[1] * nsurf / (t2 - t1))))
    print("rays reaching the last surface: %d" % raypath[0].raybundles[-1].x.shape[2])


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000)
