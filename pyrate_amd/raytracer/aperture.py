"""Aperture descriptors (reference: raytracer/aperture.py:34-154).  The predicates
themselves are evaluated per ray inside the HIP propagate step; these classes only
carry the parameters and the frame, with the reference's constructor signatures."""
import math

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


class CircularAperture(BaseAperture):
    kind = "aperture_Circular"

    @classmethod
    def p(cls, lc, maxradius=1.0, minradius=0.0, name="", *_):
        return cls(lc, {"maxradius": maxradius, "minradius": minradius,
                        "typicaldimension": maxradius}, name=name)


class RectangularAperture(BaseAperture):
    kind = "aperture_Rectangle"

    @classmethod
    def p(cls, lc, width=1.0, height=1.0, name="", *_):
        return cls(lc, {"width": width, "height": height,
                        "typicaldimension": math.sqrt(width ** 2 + height ** 2)}, name=name)


ACCESSIBLE_APERTURES = {None: BaseAperture, "CircularAperture": CircularAperture,
                        "RectangularAperture": RectangularAperture}


def create_aperture(localcoordinates, ap_dict):
    ap_dict = dict(ap_dict)
    ap_type = ap_dict.pop("type", None)
    return ACCESSIBLE_APERTURES[ap_type].p(localcoordinates, **ap_dict)
