#!/usr/bin/env python
"""Head-up display prism (reference: demos/demo_hud.py; design: US patent 5 701 202, Takahashi; Chen and Herkommer,
Opt. Express 24, 26999 (2016)): a plastic free-form prism whose three optical faces are biconic surfaces in frames
that are tilted and decentred against the OBJECT frame (not chained) -- one face is used twice, in transmission and
in (total internal) reflection.  Dummy planes in front of and behind every face mark the patent's coordinate
breaks.  Three collimated fans (0, +-15 degrees) are traced; prints where they land on the image plane."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np

from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.aperture import CircularAperture
from pyrate_amd.raytracer.globalconstants import degree, standard_wavelength
from pyrate_amd.raytracer.localcoordinates import LocalCoordinates
from pyrate_amd.raytracer.material.material_isotropic import ConstantIndexGlass
from pyrate_amd.raytracer.optical_element import OpticalElement
from pyrate_amd.raytracer.optical_system import OpticalSystem
from pyrate_amd.raytracer.ray import RayBundle
from pyrate_amd.raytracer.surface import Surface
from pyrate_amd.raytracer.surface_shape import Biconic, Conic

N_PLASTIC = 1.492
CLEAR = 40.0

# the patent's frames, all relative to the object frame: name -> (decy, decz, tiltx in degrees)
FRAMES = {
    "d1": (0.0, 30.002, 0.0), "s1": (-24.028, 26.360, 14.7), "d1p": (0.0, 30.002, 1.066),
    "d2": (-0.251, 43.485, 1.066), "s2": (19.109, 33.339, -36.660), "d2p": (-0.251, 43.485, -38.376),
    "d3": (-11.858, 28.827, -38.376), "s3": (-24.028, 26.360, 14.7), "d3p": (-11.858, 28.827, 55.019),
    "d4": (-23.067, 36.667, 55.019), "s4": (-35.215, 18.817, 47.770), "d4p": (-23.067, 36.667, 50.668),
    "image": (-30.892, 43.083, 50.668),
}
# the optical faces: biconic parameters (radii in mm; the even coefficient pairs (A, B) start at the fourth order)
FACES = {
    "s1": dict(ry=-108.187, rx=-73.105, ccy=0.0, ccx=0.0, coefficients=[(0., 0.), (5.542e-7, -0.08), (8.176e-11, -1.379)]),
    "s2": dict(ry=-69.871, rx=-60.374, ccy=-0.1368, ccx=-0.123,
               coefficients=[(0., 0.), (-7.233e-11, 29.075), (-4.529e-12, -2.085)]),
}
# the light path: surface key, (medium before, medium behind), reflects?
PATH = (("object", (None, None), False), ("d1", (None, None), False), ("s1", (None, "plastic"), False),
        ("d1p", ("plastic", "plastic"), False), ("d2", ("plastic", "plastic"), False),
        ("s2", ("plastic", "plastic"), True), ("d2p", ("plastic", "plastic"), False),
        ("d3", ("plastic", "plastic"), False), ("s3", ("plastic", "plastic"), True),
        ("d3p", ("plastic", "plastic"), False), ("d4", ("plastic", "plastic"), False),
        ("s4", ("plastic", None), False), ("d4p", (None, None), False), ("image", (None, None), False))


def build(api=None):
    """``api``: a namespace with the classes to build from (tests build the same prescription from the reference's
    own classes to generate golden vectors); default: this package's"""
    import types
    if api is None:
        api = types.SimpleNamespace(OpticalSystem=OpticalSystem, LocalCoordinates=LocalCoordinates,
                                    OpticalElement=OpticalElement, Surface=Surface, CircularAperture=CircularAperture,
                                    ConstantIndexGlass=ConstantIndexGlass, Biconic=Biconic, Conic=Conic)
    (OpticalSystem_, LocalCoordinates_, OpticalElement_, Surface_) = (api.OpticalSystem, api.LocalCoordinates,
                                                                     api.OpticalElement, api.Surface)
    s = OpticalSystem_.p(name="hud")
    obj = s.addLocalCoordinateSystem(LocalCoordinates_.p(name="object", decz=0.0), refname=s.rootcoordinatesystem.name)
    frames = {"object": obj}
    for (name, (decy, decz, tilt_deg)) in FRAMES.items():
        frames[name] = s.addLocalCoordinateSystem(
            LocalCoordinates_.p(name=name, decy=decy, decz=decz, tiltx=tilt_deg * degree, tiltThenDecenter=False),
            refname=obj.name)

    def face(name):
        lc = frames[name]
        if name == "s4":
            shape = api.Conic.p(lc, curv=1. / 77.772)
        else:
            f = FACES["s1" if name == "s3" else name]            # the first face again, now from the inside
            shape = api.Biconic.p(lc, curvy=1. / f["ry"], curvx=1. / f["rx"], ccy=f["ccy"], ccx=f["ccx"],
                                  coefficients=f["coefficients"])
        return Surface_.p(lc, shape=shape, aperture=api.CircularAperture.p(lc, maxradius=CLEAR))
    elem = OpticalElement_.p(obj, name="hud")
    elem.addMaterial("plastic", api.ConstantIndexGlass.p(obj, N_PLASTIC))
    sequence = []
    for (key, media, reflects) in PATH:
        surf = face(key) if key in ("s1", "s2", "s3", "s4") else Surface_.p(frames[key])
        elem.addSurface(key, surf, media)
        options = {"is_mirror": True} if reflects else {}
        if key == "object":
            options["is_stop"] = True
        sequence.append((key, options))
    s.addElement("hud", elem)
    return (s, [("hud", sequence)])


def main(nrays=9):
    from pyrate_amd.sampling2d import raster
    (s, seq) = build()
    osa = OpticalSystemAnalysis(s, seq, name="hud analysis")
    out = {}
    for field_deg in (0.0, 15.0, -15.0):
        (o, k, e) = osa.collimated_bundle(nrays, {"radius": 2.0, "raster": raster.MeridionalFan(),
                                                  "anglex": field_deg * degree}, wave=standard_wavelength)
        path = s.seqtrace(RayBundle(x0=o, k0=k, Efield0=e, wave=standard_wavelength), seq)[0]
        img = path.raybundles[-1]
        y = img.x[-1, 1, :]
        print("hud, field %+5.1f deg: %d of %d rays on the image plane, at y = %.3f ... %.3f mm (global)"
              % (field_deg, img.num_rays, o.shape[1], float(y.min()) if img.num_rays else float("nan"),
                 float(y.max()) if img.num_rays else float("nan")))
        out[field_deg] = img.num_rays
    return out


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 9)
