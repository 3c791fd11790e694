#!/usr/bin/env python
"""Dispersing prism (reference: demos/demo_prism.py): two plane faces tilted by +-30 degrees, Conrady
ModelGlass, a meridional fan at 23 degrees traced at a red and a blue wavelength through the
``raytrace`` convenience; prints where the two colours land on the image plane."""
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
from pyrate_amd.raytracer.material.material_isotropic import ModelGlass
from pyrate_amd.raytracer.optical_element import OpticalElement
from pyrate_amd.raytracer.optical_system import OpticalSystem
from pyrate_amd.raytracer.surface import Surface
from pyrate_amd.raytracer.surface_shape import Conic
from pyrate_amd.sampling2d import raster


def build():
    s = OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="stop", decz=0.0), refname=s.rootcoordinatesystem.name)
    lcc = s.addLocalCoordinateSystem(LocalCoordinates.p(name="prismcenter", decz=50.0), refname=lc0.name)
    lc1 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="surf1", decz=-10.0, tiltx=30. * degree), refname=lcc.name)
    lc2 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="surf2", decz=10.0, tiltx=-30. * degree), refname=lcc.name)
    lc3 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="image", decz=50.0), refname=lcc.name)
    elem = OpticalElement.p(lc0, name="prism")
    elem.addMaterial("glass", ModelGlass.p(lc1))
    elem.addSurface("stop", Surface.p(lc0), (None, None))
    elem.addSurface("surf1", Surface.p(lc1, shape=Conic.p(lc1, curv=0), aperture=CircularAperture.p(lc1, maxradius=20.0)),
                    (None, "glass"))
    elem.addSurface("surf2", Surface.p(lc2, shape=Conic.p(lc2, curv=0), aperture=CircularAperture.p(lc2, maxradius=20.0)),
                    ("glass", None))
    elem.addSurface("image", Surface.p(lc3), (None, None))
    s.addElement("prism", elem)
    return (s, [("prism", [("stop", {"is_stop": True}), ("surf1", {}), ("surf2", {}), ("image", {})])])


def main(nrays=20):
    (s, seq) = build()
    raysdict = {"radius": 5.0, "startz": -5., "starty": -20., "anglex": 23 * degree, "raster": raster.MeridionalFan()}
    out = {}
    for (name, wave) in (("red", 0.700e-3), ("blue", 0.470e-3)):
        rp = raytrace(s, seq, nrays, raysdict, wave=wave)[0][0]
        img = rp.raybundles[-1]
        y = img.x[-1, 1, :]
        print("prism, %-4s (%.0f nm): %d rays on the image plane, mean height %.4f mm, fan width %.4f mm"
              % (name, wave * 1e6, img.num_rays, float(np.mean(y)), float(y.max() - y.min())))
        out[name] = float(np.mean(y))
    print("dispersion: blue lands %.4f mm from red" % (out["blue"] - out["red"]))
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
