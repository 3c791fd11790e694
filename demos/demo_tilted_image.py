#!/usr/bin/env python
"""Tilted image plane (reference: demos/demo_tilted_image.py): a plano-convex singlet built by the rotationally
symmetric builder, whose image frame is tilted by 10 degrees AFTER construction -- ``tiltx.set_value`` on the frame
of a surface and ``update()``, the way a tolerancing or optimisation loop moves things.  Three collimated fans (0,
+-1 degree) are traced before and after the tilt; prints where the fans cross the image plane.  (The reference goes
on to its paraxial XYUV matrices, which are outside this engine's scope.)"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np

from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.globalconstants import degree
from pyrate_amd.raytracer.ray import RayBundle

TILT = 10. * degree
PUPIL_RADIUS = 5.0
# (radius of curvature, conic constant, distance to the surface before, medium behind, key, options)
SINGLET = ((0, 0, 0., None, "object", {}),
           (100., 0, 5, 1.5, "lens1front", {"is_stop": True}),
           (0., 0, 5, None, "lens1rear", {}),
           (0, 0, 196.228, None, "image", {}))
FIELDS = (0.0, 1.0, -1.0)


def footprints(s, seq, nrays):
    """per field: (mean, spread) of the local y coordinate of the fan on the image surface"""
    from pyrate_amd.sampling2d.raster import MeridionalFan
    osa = OpticalSystemAnalysis(s, seq, name="fans")
    frame = s.elements["stdelem"].surfaces["image"].rootcoordinatesystem
    out = []
    for field in FIELDS:
        (o, k, e) = osa.collimated_bundle(nrays, {"radius": PUPIL_RADIUS, "raster": MeridionalFan(), "anglex": field * degree})
        last = s.seqtrace(RayBundle(o, k, e), seq)[0].raybundles[-1]
        y = frame.returnGlobalToLocalPoints(np.real(last.x[-1]))[1]
        out.append((float(np.mean(y)), float(np.max(y) - np.min(y)), int(last.num_rays)))
    return out


def main(nrays=11):
    (s, seq) = build_rotationally_symmetric_optical_system(list(SINGLET), name="os")
    before = footprints(s, seq, nrays)
    image = s.elements["stdelem"].surfaces["image"]
    image.rootcoordinatesystem.tiltx.set_value(TILT)
    image.rootcoordinatesystem.update()
    after = footprints(s, seq, nrays)
    for (field, b, a) in zip(FIELDS, before, after):
        print("tilted image, field %+.0f deg: fan centre %.4f -> %.4f mm, fan width %.4f -> %.4f mm on the image surface (%d rays)"
              % (field, b[0], a[0], b[1], a[1], a[2]))
    return (before, after)


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 11)
