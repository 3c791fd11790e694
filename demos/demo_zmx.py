#!/usr/bin/env python
"""Trace a Zemax prescription (reference: demos/demo_zmx.py): ZMXParser builds the system, the file's
own field points and pupil definition give the initial bundles (create_initial_bundle ->
OpticalSystemAnalysis.aim), every field is traced on the GPU and its spot is reported.

    python demos/demo_zmx.py [file.zmx] [nrays] [GLASS=index ...]

Without arguments: tests/golden/lenssystem.ZMX (the reference's own test file) with BK7 = 1.5168."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
from pyrate_amd.raytracer.io.zmx import ZMXParser
from pyrate_amd.raytracer.localcoordinates import LocalCoordinates
from pyrate_amd.raytracer.material.material_isotropic import ConstantIndexGlass


def main(filename=None, nrays=1000, glasses=None):
    if filename is None:
        filename = os.path.join(_ROOT, "tests", "golden", "lenssystem.ZMX")
        glasses = {"BK7": 1.5168} if glasses is None else glasses
    zp = ZMXParser(filename, name="zmx")
    lc = LocalCoordinates.p(name="glasses")
    matdict = {name: ConstantIndexGlass.p(lc, n, name=name) for (name, n) in (glasses or {}).items()}
    (s, seq) = zp.create_optical_system(matdict)
    if s is None:
        raise SystemExit("the file names glasses; give their indices as NAME=index arguments")
    field = zp.read_field()
    wave = field["wavelengths"][0][0] if field.get("wavelengths") else 0.5876e-3
    osa = OpticalSystemAnalysis(s, seq, name="zmx analysis")
    results = []
    for bundle_dict in zp.create_initial_bundle():
        osa.aim(nrays, dict(bundle_dict), bundletype="collimated", wave=wave)
        rp = osa.trace()[0][0]
        img = rp.raybundles[-1]
        (xy, rms) = osa.get_spot(rp)
        c = RayBundleAnalysis(img).get_centroid_position()
        print("field %-40s: %5d / %5d rays at the image, centroid (%.4f, %.4f, %.4f) mm, RMS spot %.5f mm"
              % (bundle_dict, xy.shape[1], osa.initial_bundles[0].num_rays, c[0], c[1], c[2], rms))
        results.append((xy.shape[1], rms))
    return results


if __name__ == "__main__":
    args = sys.argv[1:]
    gl = {a.split("=")[0]: float(a.split("=")[1]) for a in args if "=" in a}
    pos = [a for a in args if "=" not in a]
    main(pos[0] if pos else None, int(float(pos[1])) if len(pos) > 1 else 1000, gl or None)
