"""Homogeneous anisotropic medium with a constant epsilon tensor (reference:
raytracer/material/material_anisotropic.py:35-155).  refract / reflect double the ray
count ([sol2, sol3] / -[sol0, sol1] stacking) and are solved per ray on the GPU
(csrc/prt_aniso.h)."""
import numpy as np

from ..globalconstants import standard_wavelength
from .material import Material


class AnisotropicMaterial(Material):
    kind = "anisotropicmaterial"

    @classmethod
    def p(cls, lc, epstensor, name="", comment=""):
        obj = cls(lc, name=name, comment=comment)
        obj.epstensor = np.array(epstensor)
        obj.annotations["epstensor"] = obj.epstensor.tolist()
        return obj

    def get_epsilon_tensor(self, x, wave=standard_wavelength):
        (_, num_pts) = np.shape(x)
        return np.repeat(self.epstensor[:, :, np.newaxis], num_pts, axis=2)
