#!/usr/bin/env python
"""Trace a WinLens SPD prescription (reference: demos/demo_spd.py): SPDParser builds the system --
glasses from the file's own GlassIndex rows (Conrady model through the d, F, C indices), or from a
refractiveindex.info database when its path is given --, bundles start at the object plane and aim
at the entrance pupil of the file's paraxial summary, like the reference demo's ``bundle``.

    python demos/demo_spd.py file.spd [nrays] [database path]"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import numpy as np

from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
from pyrate_amd.raytracer.io.spd import SPDParser
from pyrate_amd.raytracer.ray import RayBundle
from pyrate_amd.sampling2d.raster import RectGrid


def field_bundle(psys, fpy, nrays, rpup, wave):
    """rays from the object point at relative field height fpy through the entrance pupil"""
    (px, py) = RectGrid().getGrid(nrays)
    pupil = np.vstack((rpup * px, rpup * py, np.full(px.shape, psys.entpup)))
    o = np.vstack((np.zeros_like(px), np.full(px.shape, -fpy * psys.field_size_obj()),
                   np.full(px.shape, psys.obj_dist())))
    k = pupil - o
    k = k / np.sqrt(np.sum(k * k, axis=0))
    e0 = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T
    return RayBundle(o, k, e0, wave=wave)


def main(filename, nrays=2000, db_path=None):
    sp = SPDParser(filename, name=os.path.basename(filename))
    options = None
    if db_path:
        from pyrate_amd.raytracer.material.material_glasscat import GlassCatalog
        options = {"gcat": GlassCatalog(db_path), "db_path": db_path}
    (s, seq) = sp.create_optical_system(options=options)
    psys = sp.psys
    print("efl %.4f mm, object at %.3f mm, paraxial image at %.4f mm behind the last surface, "
          "entrance pupil at %.4f mm (radius %.4f mm)"
          % (psys.efl, psys.obj_dist(), psys.img_dist(), psys.entpup, psys.entpup_rad))
    results = []
    waves = [w * 1e-6 for w in psys.spd.wavelengths_nm[:3]] or [0.5876e-3]
    for wave in waves:
        for fpy in (0.0, 0.7, 1.0):
            rp = s.seqtrace(field_bundle(psys, fpy, nrays, psys.entpup_rad, wave), seq)[0]
            img = rp.raybundles[-1]
            ra = RayBundleAnalysis(img)
            c = ra.get_centroid_position()
            rms = ra.get_rms_spot_size_centroid()
            print("wave %.1f nm  field %.1f : %5d rays at the image, centroid y %.4f mm (paraxial %.4f), RMS spot %.5f mm"
                  % (wave * 1e6, fpy, img.num_rays, c[1], -fpy * psys.field_size_img(), rms))
            results.append((fpy, c[1], rms))
    return results


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    main(sys.argv[1], int(float(sys.argv[2])) if len(sys.argv) > 2 else 2000,
         sys.argv[3] if len(sys.argv) > 3 else None)
