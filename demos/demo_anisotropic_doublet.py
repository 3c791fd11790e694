#!/usr/bin/env python
"""BASELINE config 4: the doublet of demos/demo_anisotropic_doublet.py:55-121 with
AnisotropicMaterial crystals: (i) the demo's eps = n^2 I tensors, (ii) a birefringent
calcite-like uniaxial crystal.  Every crystal interface doubles the rays; splitup=True forks
the ray paths (4 paths after two crystals)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import math

import numpy as np

from pyrate_amd import systems
from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
from pyrate_amd.sampling2d import raster
from pyrate_amd.builders import build_simple_optical_system

wavelength = 0.5876e-3


def build(eps1, eps2):
    """the cemented doublet of the reference demo (:55-121); {"eps": tensor} = AnisotropicMaterial"""
    def lens_aperture():
        return {"type": "CircularAperture", "maxradius": 12.7}
    return build_simple_optical_system([
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic", "curv": 1. / 62.8, "aperture": lens_aperture()}, {"decz": -1.048}, {"eps": eps1}, "front", {}),
        ({"shape": "Conic", "curv": -1. / 45.7, "aperture": lens_aperture()}, {"decz": 4.0}, {"eps": eps2}, "cement", {}),
        ({"shape": "Conic", "curv": -1. / 128.2, "aperture": lens_aperture()}, {"decz": 2.5}, None, "rear", {}),
        ({"shape": "Conic"}, {"decz": 97.2}, None, "image", {})])


def main(nrays=11):
    c = systems.CALCITE_TILTED
    cases = {
        "eps = n^2 I (demo)": (1.5168 ** 2 * np.eye(3), 1.6727 ** 2 * np.eye(3)),
        "uniaxial crystals": (systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
                              systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))),
    }
    for (name, (e1, e2)) in cases.items():
        (s, sysseq) = build(e1, e2)
        osa = OpticalSystemAnalysis(s, sysseq, name="Analysis")
        osa.aim(nrays, {"radius": 11.43, "startz": -5., "raster": raster.MeridionalFan()},
                bundletype="collimated", wave=wavelength)
        paths = osa.trace(splitup=True)[0]
        print("%s: %d ray paths" % (name, len(paths)))
        for (i, rp) in enumerate(paths):
            img = rp.raybundles[-1]
            print("   path %d: %d rays, RMS spot %.5f mm" % (i, img.x.shape[2],
                                                          RayBundleAnalysis(img).get_rms_spot_size_centroid()))
        stacked = osa.trace(splitup=False)[0][0].raybundles[-1]
        print("   splitup=False: one bundle with %d rays (2 doublings)" % stacked.x.shape[2])


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 11)
