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

    # Conrady coefficients from catalogue numbers.  The reference's versions of these three
    # (material_isotropic.py:311-353) call a setter that does not exist and swap A and B; these set
    # the coefficients of the formulas their docstrings describe.
    def calcCoefficientsFrom_nd_vd_PgF(self, nd=1.51680, vd=64.17, PgF=0.5349):
        nF_minus_nC = (nd - 1.) / vd
        b_um = 0.454670392956 * nF_minus_nC * (PgF - 0.445154791693)
        a_um = 1.87513751845 * nF_minus_nC - b_um * 15.2203074842
        self.n0.set_value(nd - 1.70194862906 * a_um - 6.43150432188 * b_um)
        self.A.set_value(a_um * 1e-3)
        self.B.set_value(b_um * (1e-3) ** 3.5)

    def calcCoefficientsFrom_nd_vd(self, nd=1.51680, vd=64.17):
        self.calcCoefficientsFrom_nd_vd_PgF(nd, vd, 0.6438 - 0.001682 * vd)

    def calcCoefficientsFromSchottCode(self, schottCode=517642):
        """six-digit code: first three digits 1000 (nd - 1), last three 10 vd; anything else is
        replaced by N-BK7 like in the reference"""
        if isinstance(schottCode, int) and 1e5 <= schottCode < 1e6:
            (nd, vd) = (1 + 0.001 * (schottCode // 1000), 0.1 * (schottCode % 1000))
        else:
            (nd, vd) = (1.51680, 64.17)
        self.calcCoefficientsFrom_nd_vd(nd, vd)
