#!/usr/bin/env python
"""Ordinary and extraordinary rays in a uniaxial plate (reference: demos/demo_anisotropic_ord_eo.py): a plane-parallel
crystal slab, optic axis along the plate normal, lit by a divergent meridional fan.  With ``splitup=True`` the
trace forks at the entrance face into two ray paths -- one per polarisation mode --, without it the doubled rays
travel in one path (``[sol2, sol3]`` stacking).  Prints where the two modes of every ray leave the plate and how far
apart they land on the image plane."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np

from pyrate_amd.builders import build_simple_optical_system, raytrace
from pyrate_amd.raytracer.globalconstants import degree, standard_wavelength

# the reference's tensor: diag(1.5, 1.5, 1.8) -- axis along z (the plate normal), n_o^2 = 1.5, n_e^2 = 1.8
EPS = np.diag([1.5, 1.5, 1.8])
PLATE = (
    ({"shape": "Conic"}, {"decz": 1.0}, None, "stop", {"is_stop": True}),
    ({"shape": "Conic", "aperture": {"type": "CircularAperture", "maxradius": 10.0}}, {"decz": 10.0}, {"eps": EPS}, "front", {}),
    ({"shape": "Conic", "aperture": {"type": "CircularAperture", "maxradius": 10.0}}, {"decz": 5.0}, None, "rear", {}),
    ({"shape": "Conic"}, {"decz": 10.0}, None, "image", {}),
)


def build():
    return build_simple_optical_system(list(PLATE), name="plate")


def main(nrays=10):
    from pyrate_amd.sampling2d import raster
    (s, seq) = build()
    fan = {"opticalsystem": s, "startz": -5.0, "radius": 20 * degree, "raster": raster.MeridionalFan()}
    forks = raytrace(s, seq, nrays, fan, bundletype="divergent", traceoptions={"splitup": True},
                     wave=standard_wavelength)[0]
    assert len(forks) == 2 and not forks[0].containsSplitted()
    y = [np.real(p.raybundles[-1].x[-1])[1] for p in forks]
    n = min(len(y[0]), len(y[1]))
    print("plate, splitup: %d ray paths, %d / %d rays on the image plane" % (len(forks), len(y[0]), len(y[1])))
    gap = np.abs(y[0][:n] - y[1][:n])
    print("   the two modes of a ray land %.4f ... %.4f mm apart (0 on the axis: both modes see n_o there)"
          % (float(gap.min()), float(gap.max())))
    one = raytrace(s, seq, nrays, fan, bundletype="divergent", traceoptions={"splitup": False},
                   wave=standard_wavelength)[0]
    last = one[0].raybundles[-1]
    print("plate, one path: %d ray path, %d rays on the image plane (%d went in), splitted: %s"
          % (len(one), last.num_rays, forks[0].raybundles[0].num_rays, one[0].containsSplitted()))
    return {"paths": len(forks), "gap_max": float(gap.max()), "rays_one_path": int(last.num_rays),
            "rays_in": int(forks[0].raybundles[0].num_rays)}


if __name__ == "__main__":
    main(int(float(sys.argv[1])) if len(sys.argv) > 1 else 10)
