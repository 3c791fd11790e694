#!/usr/bin/env python
"""The optimisation loop of the reference's demos/demo_asphere.py:61-100: Nelder-Mead on
curvature, conic constant, A4, A6 of the aspheric back surface, merit = sum of squared image
heights of a 121-ray collimated bundle.  Every merit evaluation is one OpticalSystem.seqtrace on
the GPU (the surface table is re-flattened from the changed FloatVariables each call).  The
reference drives this through its Optimizer/ScipyBackend classes (out of scope, SURVEY.md
section 2 #14); here scipy.optimize.minimize is called directly."""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np
from scipy.optimize import minimize

from pyrate_amd.builders import build_simple_optical_system
from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.ray import RayBundle

wavelength = 0.5876e-3


def main(maxiter=400, fast=False):
    (s, sysseq) = build_simple_optical_system([
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic"}, {"decz": 5.0}, 1.5168, "front", {}),
        ({"shape": "Asphere", "curv": -1. / 50., "cc": -1., "coefficients": [0.0, 0.0, 0.0]},
         {"decz": 20.0}, None, "back", {}),
        ({"shape": "Conic"}, {"decz": 100.0}, None, "image", {})])
    osa = OpticalSystemAnalysis(s, sysseq, name="Analysis")
    (o, k, e0) = osa.collimated_bundle(121, {"startz": -5., "radius": 11.43}, wave=wavelength)
    bundle = RayBundle(x0=o, k0=k, Efield0=e0, wave=wavelength)
    params = s.elements["stdelem"].surfaces["back"].shape.params
    names = ["curv", "cc", "A4", "A6"]
    ncalls = [0]

    def merit(v):
        for (name, val) in zip(names, v):
            params[name].set_value(val)
        ncalls[0] += 1
        if fast:
            # the same number from the moments the trace launch reduces itself: sum (x^2 + y^2) about
            # the axis = S2x + S2y, arrived rays = count (OpticalSystem.image_moments)
            (m, _) = s.image_moments(bundle, sysseq)
            return float(m[4] + m[5]) + 1e6 * (o.shape[1] - m[0])
        x = s.seqtrace(bundle, sysseq)[0].raybundles[-1].x[-1]
        return float(np.sum(x[0] ** 2 + x[1] ** 2)) + 1e6 * (o.shape[1] - x.shape[1])   # lost rays are penalised

    v0 = np.array([params[n]() for n in names])
    m0 = merit(v0)
    t0 = time.perf_counter()
    res = minimize(merit, v0, method="Nelder-Mead", options={"maxiter": maxiter, "xatol": 1e-12, "fatol": 1e-12})
    dt = time.perf_counter() - t0
    print("merit %.6e -> %.6e after %d traces (%.2f ms per merit evaluation%s)"
          % (m0, res.fun, ncalls[0], dt / max(ncalls[0] - 1, 1) * 1e3,
             ", image-plane moments from the trace launch" if fast else ""))
    print("optimised back surface: " + ", ".join("%s=%.6g" % (n, v) for (n, v) in zip(names, res.x)))
    return (m0, res.fun)


if __name__ == "__main__":
    main(fast="--fast" in sys.argv)
