"""Isotropic media (reference: raytracer/material/material_isotropic.py:39-309)."""
from ..globalconstants import standard_wavelength
from ..variables import FloatVariable
from .material import Material


class IsotropicMaterial(Material):
    kind = "isotropicmaterial"

    def get_optical_index(self, xpos, wave):
        raise NotImplementedError()

    def get_isotropic_epsilon(self, xpos, wave=standard_wavelength):
        return self.get_optical_index(xpos, wave=wave) ** 2


class ConstantIndexGlass(IsotropicMaterial):
    """a glass defined by a single refractive index"""
    kind = "constantindexglass"

    @classmethod
    def p(cls, lc, n=1.0, name="", comment=""):
        obj = cls(lc, name=name, comment=comment)
        obj.n = FloatVariable(n, "refractive index")
        return obj

    def get_optical_index(self, x, wave):
        return self.n.evaluate()


class ModelGlass(IsotropicMaterial):
    """Conrady dispersion n = n0 + A / wave + B / wave**3.5 (n0 [1], A [mm], B [mm**3.5])"""
    kind = "modelglass"

    @classmethod
    def p(cls, lc, n0_A_B=(1.49749699179, 0.0100998734374 * 1e-3,
                           0.000328623343942 * (1e-3) ** 3.5), name="", comment=""):
        (n0, A, B) = n0_A_B
        obj = cls(lc, name=name, comment=comment)
        obj.n0 = FloatVariable(n0, "Conrady n0")
        obj.A = FloatVariable(A, "Conrady A")
        obj.B = FloatVariable(B, "Conrady B")
        return obj

    def get_optical_index(self, x, wave):
        return self.n0() + self.A() / wave + self.B() / (wave ** 3.5)
