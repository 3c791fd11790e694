#!/usr/bin/env python
"""Reflection inside a crystal (reference: demos/demo_anisotropic_mirror.py): a uniaxial slab whose
tilted rear face is a mirror.  Entering the slab splits every ray in two, the mirror splits each of
them again (the two backward modes); with ``splitup=True`` every branch comes back as its own
RayPath, otherwise as one path with stacked bundles."""
import math
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np

from pyrate_amd.builders import raytrace
from pyrate_amd.raytracer.aperture import CircularAperture
from pyrate_amd.raytracer.globalconstants import degree
from pyrate_amd.raytracer.localcoordinates import LocalCoordinates
from pyrate_amd.raytracer.material.material_anisotropic import AnisotropicMaterial
from pyrate_amd.raytracer.optical_element import OpticalElement
from pyrate_amd.raytracer.optical_system import OpticalSystem
from pyrate_amd.raytracer.surface import Surface
from pyrate_amd.raytracer.surface_shape import Conic
from pyrate_amd.sampling2d import raster


def build(no=1.5, neo=1.8):
    s = OpticalSystem.p(name="os")
    lc0 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="stop", decz=1.0), refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="surf1", decz=10.0), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="surf2", decz=5.0, tiltx=10 * degree), refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="image", decz=-5.0, tiltx=-10 * degree), refname=lc2.name)
    elem = OpticalElement.p(lc0, name="crystalelem")
    elem.addMaterial("crystal", AnisotropicMaterial.p(lc1, np.diag([no, no, neo])))
    elem.addSurface("stop", Surface.p(lc0), (None, None))
    elem.addSurface("front", Surface.p(lc1, shape=Conic.p(lc1, curv=0), aperture=CircularAperture.p(lc1, maxradius=10.0)),
                    (None, "crystal"))
    elem.addSurface("rear", Surface.p(lc2, shape=Conic.p(lc2, curv=0), aperture=CircularAperture.p(lc3, maxradius=10.0)),
                    ("crystal", "crystal"))
    elem.addSurface("image", Surface.p(lc3), ("crystal", None))
    s.addElement("crystalelem", elem)
    return (s, [("crystalelem", [("stop", {}), ("front", {}), ("rear", {"is_mirror": True}), ("image", {})])])


def main(nrays=10):
    (s, seq) = build()
    rays = {"radius": 20 * degree, "startz": -5., "raster": raster.MeridionalFan()}
    forks = raytrace(s, seq, nrays, dict(rays), bundletype="divergent", traceoptions={"splitup": True},
                     wave=0.5876e-3)[0]
    stacked = raytrace(s, seq, nrays, dict(rays), bundletype="divergent", wave=0.5876e-3)[0][0]
    print("crystal mirror: %d ray paths with splitup, %d rays each; one stacked path with %d rays"
          % (len(forks), forks[0].raybundles[-1].num_rays, stacked.raybundles[-1].num_rays))
    for (i, rp) in enumerate(forks):
        y = rp.raybundles[-1].x[-1, 1, :]
        print("  branch %d: image heights %.4f .. %.4f mm" % (i, float(np.nanmin(y)), float(np.nanmax(y))))
    return (forks, stacked)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
