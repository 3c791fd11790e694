#!/usr/bin/env python
"""All-reflective system (reference: demos/demo_mirrors.py): a three-mirror anastigmat of tilted, decentred spherical
mirrors with an intermediate image, followed by an off-axis paraboloid used far from its vertex -- every frame hangs
on the one before it, every deflection is a reflection in air.  Three collimated fields (0, +-0.5 degrees) are traced
and their footprints on the three image planes printed.  (The reference aims its fields through the stop with its
paraxial machinery, which is outside the HIP engine's scope; here the fields are collimated bundles.)"""
import math
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np

from pyrate_amd.builders import build_simple_optical_system, raytrace
from pyrate_amd.raytracer.globalconstants import degree, standard_wavelength

PI = math.pi
# (key, shape, frame relative to the previous surface, options, clear radius or None)
PRESCRIPTION = (
    ("object", {}, {"decz": 0.0}, {}, None),
    ("m1", {"curv": -0.01}, {"decz": 50.0, "tiltx": -PI / 8}, {"is_mirror": True}, None),
    ("m2", {"curv": 0.01}, {"decz": -50.0, "decy": -20.0, "tiltx": PI / 16}, {"is_mirror": True, "is_stop": True}, None),
    ("m3", {"curv": -0.006}, {"decz": 50.0, "decy": -30.0, "tiltx": 3 * PI / 32}, {"is_mirror": True}, None),
    ("image1", {}, {"decz": -50.0, "decy": -15.0, "tiltx": -PI / 16}, {}, None),
    ("oapara", {"curv": 0.01, "cc": -1.0}, {"decz": -100.0, "decy": -35.0}, {"is_mirror": True}, None),
    ("image2", {}, {"decz": 52.8, "tiltx": PI / 32}, {}, 20.0),
    ("image3", {}, {"decz": 5.0}, {}, 20.0),
)
IMAGES = ("image1", "image2", "image3")


def build(builder=None):
    """``builder``: the ``build_simple_optical_system`` to use (tests pass the reference's own to generate golden
    vectors); default: this package's"""
    rows = []
    for (key, shape, frame, options, clear) in PRESCRIPTION:
        spec = dict(shape, shape="Conic")
        if clear is not None:
            spec["aperture"] = {"type": "CircularAperture", "maxradius": clear}
        rows.append((spec, frame, None, key, options))           # None: air behind every surface
    return (builder or build_simple_optical_system)(rows, name="TMA")


def main(nrays=300):
    from pyrate_amd.sampling2d import raster
    (s, seq) = build()
    keys = [k for (k, _) in seq[0][1]]
    out = {}
    for field_deg in (0.0, 0.5, -0.5):
        bundle = {"radius": 2.0, "startz": -5.0, "anglex": field_deg * degree, "raster": raster.RectGrid()}
        path = raytrace(s, seq, nrays, bundle, wave=standard_wavelength)[0][0]
        line = []
        for name in IMAGES:
            b = path.raybundles[keys.index(name) + 2]          # bundle 0 is the initial one (twice), then one per surface
            x = b.x[0]
            c = x.mean(axis=1)
            rms = float(np.sqrt(np.mean(np.sum((x - c[:, None]) ** 2, axis=0))))
            line.append("%s: %d rays, rms %.4f mm" % (name, b.num_rays, rms))
            out[(field_deg, name)] = (b.num_rays, rms)
        print("mirrors, field %+.1f deg | %s" % (field_deg, " | ".join(line)))
    return out


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 300)
