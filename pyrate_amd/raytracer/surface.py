"""``Surface`` = shape + aperture + root frame (reference: raytracer/surface.py:35-135)."""
from .aperture import BaseAperture, create_aperture
from .localcoordinates import LocalCoordinatesTreeBase
from .surface_shape import Conic


class Surface(LocalCoordinatesTreeBase):
    kind = "surface"

    @classmethod
    def p(cls, rootlc, shape=None, aperture=None, name=""):
        if shape is None:
            shape = Conic.p(rootlc)
        surf = cls(rootlc, name=name)
        aperture_ = BaseAperture.p(rootlc)
        if isinstance(aperture, BaseAperture):
            aperture_ = aperture
        elif isinstance(aperture, dict):
            aperture_ = create_aperture(rootlc, aperture)
        surf.shape = shape
        surf.aperture = aperture_
        return surf

    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, shape):
        if not self.checkForRootConnection(shape.lc):
            raise Exception("Shape coordinate system should be connected to surface coordinate system")
        self._shape = shape

    @property
    def aperture(self):
        return self._aperture

    @aperture.setter
    def aperture(self, apert):
        if not self.checkForRootConnection(apert.lc):
            raise Exception("Aperture coordinate system should be connected to surface coordinate system")
        self._aperture = apert

    # accessor spellings of the reference (surface.py:73-114)
    def getShape(self):
        return self.shape

    def setShape(self, shape):
        self.shape = shape

    def getAperture(self):
        return self.aperture

    def setAperture(self, apert):
        self.aperture = apert

    def getCentralCurvature(self):
        return self.shape.getCentralCurvature()

    def intersect(self, raybundle, remove_rays_outside_aperture=True):
        """intersection + aperture vignetting; mutates the bundle (surface.py:116-135)"""
        from .material.material import propagate_bundle
        propagate_bundle(raybundle, self.shape,
                         self.aperture if remove_rays_outside_aperture else None)
