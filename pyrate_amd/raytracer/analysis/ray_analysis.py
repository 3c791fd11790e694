"""``RayBundleAnalysis`` with the reference's method names and formulas
(raytracer/analysis/ray_analysis.py:33-166).  Centroid, RMS spot size, centroid direction
and RMS angular size are two-stage device reductions (prt_bundle_moments): only seven
doubles per call cross PCIe, however many rays the bundle has."""
import math

import numpy as np

from ... import engine
from ..globalconstants import numerical_tolerance


class RayBundleAnalysis(object):
    kind = "rayanalysis"

    def __init__(self, raybundle, name=""):
        self.raybundle = raybundle
        self.name = name

    def _points(self):
        return self.raybundle.x_dev(-1)

    def get_centroid_position(self):
        """arithmetic mean position of all rays at the end of the bundle (:44-58)"""
        (cnt, s1, _) = engine.bundle_moments(self._points())
        return 1.0 / (cnt + numerical_tolerance) * s1

    def get_rms_spot_size(self, reference_pos):
        """RMS deviation of the ray positions from reference_pos (:60-77)"""
        ref = np.asarray(reference_pos, dtype=float).reshape(3)
        (cnt, _, s2) = engine.bundle_moments(self._points(), ref=ref)
        return math.sqrt(float(np.sum(s2)) / (cnt - 1 + numerical_tolerance))

    def get_rms_spot_size_centroid(self):
        return self.get_rms_spot_size(self.get_centroid_position())

    def get_centroid_direction(self):
        """normalised mean of the unit ray directions (:88-102)"""
        (_, com, _) = engine.bundle_moments(self.raybundle.direction_dev(-1), mode=1)
        return com / np.sqrt(np.sum(com ** 2))

    def get_rms_angluar_size(self, ref_direction):
        """arcsin of the RMS of |d x ref| (:104-125; spelling as in the reference)"""
        ref = np.asarray(ref_direction, dtype=float).reshape(3)
        (cnt, _, s2) = engine.bundle_moments(self.raybundle.direction_dev(-1), ref=ref, mode=2)
        return math.asin(math.sqrt(float(np.sum(s2)) / cnt))

    def get_rms_angluar_size_centroid(self):
        return self.get_rms_angluar_size(self.get_centroid_direction())

    def _point_range(self, first, last):
        rb = self.raybundle
        rb._ensure()
        npts = len(rb._x)
        stop = npts if last is None else (last if last >= 0 else npts + last)
        return list(range(first, stop))

    def get_arc_length(self, first=0, last=None):
        """per-ray arc length over the stored points first..last (:136-147), device kernel
        (prt_path_sums); returned as a NumPy array like the reference"""
        idx = self._point_range(first, last)
        if len(idx) < 2:
            return np.zeros(self.raybundle.num_rays)
        return engine.path_sums([self.raybundle._x[p] for p in idx], mode=0).cpu().numpy()

    def get_phase_difference(self, first=0, last=None):
        """per-ray sum of x.k differences between consecutive stored points (:149-163)"""
        idx = self._point_range(first, last)
        if len(idx) < 2:
            return np.zeros(self.raybundle.num_rays)
        rb = self.raybundle
        return engine.path_sums([rb._x[p] for p in idx], [rb._k[p] for p in idx], mode=1).cpu().numpy()


class RayPathAnalysis(object):
    """Optical path quantities along a whole RayPath (reference :169-213): per-ray sums of the
    bundles' arc lengths / phase differences.  Like in the reference this needs every bundle of
    the path to hold the same rays (no ray lost on the way); the sums stay on the device until
    the end."""
    kind = "raypathanalysis"

    def __init__(self, raypath, name=""):
        self.raypath = raypath
        self.name = name

    def _sum(self, first, last, mode):
        bundles = self.raypath.raybundles
        total = None
        for rb in bundles[first:last]:
            rb._ensure()
            if len(rb._x) < 2:
                continue
            if mode == 0:
                part = engine.path_sums(list(rb._x), mode=0)
            else:
                part = engine.path_sums(list(rb._x), list(rb._k), mode=1)
            if total is not None and part.shape != total.shape:
                raise ValueError("operands could not be broadcast together with shapes %s %s"
                                 % (tuple(total.shape), tuple(part.shape)))
            total = part if total is None else total + part
        if total is None:
            return np.zeros(bundles[0].num_rays)
        return total.cpu().numpy()

    def get_arc_length(self, first=0, last=None):
        return self._sum(first, last, 0)

    def get_phase_difference(self, first=0, last=None):
        return self._sum(first, last, 1)

    def get_relative_phase_difference(self, first=0, last=None, referenceray=None, wavelength=None):
        """phase difference relative to a chief ray, optionally in units of the wavelength"""
        out = self.get_phase_difference(first=first, last=last)
        if referenceray is not None:
            out = out - out[referenceray]
        if wavelength is not None:
            out = out / wavelength
        return out
