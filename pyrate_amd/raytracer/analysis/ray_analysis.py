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

    def _directions_source(self):
        rb = self.raybundle
        rb._ensure()
        if rb._dir is not None:
            return rb._dir                      # explicit Poynting directions (anisotropic media)
        if not rb._dir_from_k:
            return None                         # user bundle: E decides; use the host formula
        return rb.k_dev(-1)

    def get_centroid_direction(self):
        """normalised mean of the unit ray directions (:88-102)"""
        src = self._directions_source()
        if src is None:
            d = self.raybundle.returnKtoD()[-1]
            com = np.sum(d, axis=1)
        else:
            (_, com, _) = engine.bundle_moments(src, mode=1)
        return com / np.sqrt(np.sum(com ** 2))

    def get_rms_angluar_size(self, ref_direction):
        """arcsin of the RMS of |d x ref| (:104-125; spelling as in the reference)"""
        ref = np.asarray(ref_direction, dtype=float).reshape(3)
        src = self._directions_source()
        if src is None:
            d = self.raybundle.returnKtoD()[-1]
            cr = np.cross(d, ref, axisa=0).T
            return math.asin(math.sqrt(np.sum(cr ** 2) / d.shape[1]))
        (cnt, _, s2) = engine.bundle_moments(src, ref=ref, mode=2)
        return math.asin(math.sqrt(float(np.sum(s2)) / cnt))

    def get_rms_angluar_size_centroid(self):
        return self.get_rms_angluar_size(self.get_centroid_direction())

    def get_arc_length(self, first=0, last=None):
        """per-ray arc length over the stored points (:136-147); evaluated from the NumPy
        views (a per-ray output array, not a reduction)"""
        last_no = 0 if last is None else last
        x = self.raybundle.x
        delta_s = np.sqrt(np.sum((x[first + 1:last] - x[first:-1 + last_no]) ** 2, axis=1))
        return np.sum(delta_s, axis=0)

    def get_phase_difference(self, first=0, last=None):
        """(:149-163)"""
        last_no = 0 if last is None else last
        x = self.raybundle.x
        k_real = np.real(self.raybundle.k)
        dph = x[first + 1:last] * k_real[first + 1:last] - x[first:-1 + last_no] * k_real[first:-1 + last_no]
        return np.sum(np.sum(dph, axis=1), axis=0)
