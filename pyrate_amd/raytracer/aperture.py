"""Aperture descriptors (reference: raytracer/aperture.py:34-154).  The predicates
themselves are evaluated per ray inside the HIP propagate step; these classes only
carry the parameters and the frame, with the reference's constructor signatures."""
import math

import numpy as np

from .variables import Named


class BaseAperture(Named):
    """does not limit the beam"""
    kind = "aperture"

    def __init__(self, lc, annotations=None, name=""):
        Named.__init__(self, name)
        self.lc = lc
        self.annotations = dict(annotations or {"typicaldimension": 1e16})

    @classmethod
    def p(cls, lc, name="", *_):
        return cls(lc, {"typicaldimension": 1e16}, name=name)

    def get_typical_dimension(self):
        return self.annotations["typicaldimension"]

    # host-side form of the predicate for callers that hold NumPy points in the aperture frame
    # (aperture.py:71-88); the trace itself evaluates it inside the kernels (aperture_ok)
    def get_boolean_function(self):
        return lambda x, y: np.ones_like(x, dtype=bool)

    def are_points_in_aperture(self, x_intersection, y_intersection):
        return self.get_boolean_function()(np.asarray(x_intersection), np.asarray(y_intersection))


class CircularAperture(BaseAperture):
    kind = "aperture_Circular"

    @classmethod
    def p(cls, lc, maxradius=1.0, minradius=0.0, name="", *_):
        return cls(lc, {"maxradius": maxradius, "minradius": minradius,
                        "typicaldimension": maxradius}, name=name)

    def get_boolean_function(self):
        (rmin, rmax) = (self.annotations["minradius"], self.annotations["maxradius"])
        return lambda x, y: (x ** 2 + y ** 2 >= rmin ** 2) * (x ** 2 + y ** 2 <= rmax ** 2)


class RectangularAperture(BaseAperture):
    kind = "aperture_Rectangle"

    @classmethod
    def p(cls, lc, width=1.0, height=1.0, name="", *_):
        return cls(lc, {"width": width, "height": height,
                        "typicaldimension": math.sqrt(width ** 2 + height ** 2)}, name=name)

    def get_boolean_function(self):
        (w, h) = (self.annotations["width"], self.annotations["height"])
        return lambda x, y: (x >= -w * 0.5) * (x <= w * 0.5) * (y >= -h * 0.5) * (y <= h * 0.5)


ACCESSIBLE_APERTURES = {None: BaseAperture, "CircularAperture": CircularAperture,
                        "RectangularAperture": RectangularAperture}


def create_aperture(localcoordinates, ap_dict):
    ap_dict = dict(ap_dict)
    ap_type = ap_dict.pop("type", None)
    return ACCESSIBLE_APERTURES[ap_type].p(localcoordinates, **ap_dict)
