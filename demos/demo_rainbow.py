#!/usr/bin/env python
"""Light path of the primary rainbow in a water droplet (reference: demos/demo_rainbow.py): a ray fan enters a sphere
of water, is reflected once inside and leaves through the entry surface again -- refraction, internal reflection
(the same medium on both sides of the mirror step) and refraction at ONE spherical surface visited twice, traced at
a red and a blue wavelength with a Conrady model of water.  The two hemispheres are explicit sag surfaces, like in
the reference, so only rays that enter AND leave through the front hemisphere get through: the fan of the reference
(12 degrees off the axis, 0.9 R above it at the stop) enters 0.55-0.7 R from the axis of the drop and comes back
24-29 degrees from the antisolar direction -- inside the bow, whose rays (0.86 R, 42 degrees) leave just behind the
equator.  Prints the scattering angles of both colours."""
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
from pyrate_amd.raytracer.surface_shape import Asphere

RADIUS = 0.1            # mm: a drizzle droplet
# water: n_d = 1.3330, n_F = 1.3371, n_C = 1.3312 (Conrady n0 + A / lambda + B / lambda^3.5 through these three)
WATER_D_F_C = (1.3330, 1.3371, 1.3312)

# frame name -> (axial position relative to the droplet's centre, curvature of the sphere as seen from there,
#                clear radius); "stop" and "image" are planes on the entry side
FRAMES = (("front", -RADIUS, 1. / RADIUS, RADIUS), ("rear", RADIUS, -1. / RADIUS, RADIUS))


def conrady_through(nd, nF, nC, ld=0.5876e-3, lF=0.4861e-3, lC=0.6563e-3):
    """(n0, A, B) of n = n0 + A / lambda + B / lambda^3.5 through three indices"""
    m = np.array([[1.0, 1.0 / l, l ** -3.5] for l in (ld, lF, lC)])
    return tuple(float(v) for v in np.linalg.solve(m, np.array([nd, nF, nC])))


def build():
    s = OpticalSystem.p(name="droplet")
    stop = s.addLocalCoordinateSystem(LocalCoordinates.p(name="stop", decz=0.0), refname=s.rootcoordinatesystem.name)
    centre = s.addLocalCoordinateSystem(LocalCoordinates.p(name="centre", decz=2. * RADIUS), refname=stop.name)
    elem = OpticalElement.p(stop, name="droplet")
    elem.addMaterial("water", ModelGlass.p(stop, conrady_through(*WATER_D_F_C), name="water"))
    wide = 7. * RADIUS
    elem.addSurface("stop", Surface.p(stop, aperture=CircularAperture.p(stop, maxradius=wide)), (None, None))
    for (name, z, curv, clear) in FRAMES:
        lc = s.addLocalCoordinateSystem(LocalCoordinates.p(name=name, decz=z), refname=centre.name)
        surf = Surface.p(lc, shape=Asphere.p(lc, curv=curv), aperture=CircularAperture.p(lc, maxradius=clear))
        if name == "front":
            elem.addSurface("enter", surf, (None, "water"))
            elem.addSurface("leave", surf, ("water", None))         # the same surface, visited again on the way out
        else:
            elem.addSurface("rear", surf, ("water", "water"))
    img = s.addLocalCoordinateSystem(LocalCoordinates.p(name="image", decz=-2. * RADIUS), refname=centre.name)
    elem.addSurface("image", Surface.p(img, aperture=CircularAperture.p(img, maxradius=wide)), (None, None))
    s.addElement("droplet", elem)
    seq = [("droplet", [("stop", {"is_stop": True}), ("enter", {}), ("rear", {"is_mirror": True}), ("leave", {}),
                        ("image", {})])]
    return (s, seq)


def main(nrays=11):
    from pyrate_amd.sampling2d import raster
    (s, seq) = build()
    fan = {"radius": 0.05 * RADIUS, "starty": 0.9 * RADIUS, "anglex": -12. * degree, "raster": raster.MeridionalFan()}
    bow = {}
    for (colour, wave) in (("red", 0.700e-3), ("blue", 0.470e-3)):
        path = raytrace(s, seq, nrays, fan, wave=wave)[0][0]
        k_in = np.real(path.raybundles[0].k[0][:, 0])
        out = path.raybundles[-1]
        k = np.real(out.k[-1])
        # Behind ONE reflection the wave vector of the reference's convention (k2 = -k_inplane + xi n,
        # material_isotropic.py:201-236) points against the direction of travel: the rays travel along -k, the
        # antisolar direction is -k_in, so the scattering angle is the angle between k and k_in.
        cosang = (k_in @ k) / (np.linalg.norm(k_in) * np.linalg.norm(k, axis=0))
        angle = np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0)))
        bow[colour] = float(np.mean(angle)) if out.num_rays else float("nan")
        print("droplet, %-4s (%.0f nm): %d of %d rays come back, %.2f ... %.2f degrees from the antisolar direction"
              % (colour, wave * 1e6, out.num_rays, nrays, float(np.min(angle)), float(np.max(angle))))
    print("mean scattering angle: red %.2f, blue %.2f degrees (difference %.2f)" % (bow["red"], bow["blue"], bow["red"] - bow["blue"]))
    return bow


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 11)
