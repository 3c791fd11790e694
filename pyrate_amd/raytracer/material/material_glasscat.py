"""``CatalogMaterial``: isotropic medium described by one page of the refractiveindex.info
database, given as the page's dictionary (reference: raytracer/material/material_glasscat.py:290-529).
Only the scalar dispersion n(wavelength) is needed -- it enters the trace as one number per
(material, bundle) -- so this is host-side arithmetic; the database files themselves and the
catalogue browser (``GlassCatalog``) are out of scope (the database submodule is not part of the
reference checkout).

Dispersion formulas follow the refractiveindex.info definitions (wavelength in micrometres):
formula 1 Sellmeier, 2 Sellmeier-2, 3 polynomial, 4 RefractiveIndex.INFO, 5 Cauchy, 6 gases,
7 Herzberger, and "tabulated n" (linear interpolation)."""
import numpy as np

from .material_isotropic import IsotropicMaterial


def _n_formula(typ, c, w):
    """refractive index for wavelength w [um] and coefficient vector c"""
    if typ == "formula 1":
        return np.sqrt(1 + c[0] + np.sum(c[1::2] * w ** 2 / (w ** 2 - c[2::2] ** 2)))
    if typ == "formula 2":
        return np.sqrt(1 + c[0] + np.sum(c[1::2] * w ** 2 / (w ** 2 - c[2::2])))
    if typ == "formula 3":
        return np.sqrt(c[0] + np.sum(c[1::2] * w ** c[2::2]))
    if typ == "formula 4":
        if len(c) > 10:
            idx = np.array([1, 5])
            nsq = c[0] + np.sum(c[idx] * w ** c[idx + 1] / (w ** 2 - c[idx + 2] ** c[idx + 3])) \
                + np.sum(c[9::2] * w ** c[10::2])
        else:
            nsq = c[0] + np.sum(c[1::4] * w ** c[2::4] / (w ** 2 - c[3::4] ** c[4::4]))
        return np.sqrt(nsq)
    if typ == "formula 5":
        return c[0] + np.sum(c[1::2] * w ** c[2::2])
    if typ == "formula 6":
        return 1 + c[0] + np.sum(c[1::2] / (c[2::2] - w ** (-2)))
    if typ == "formula 7":
        den = w ** 2 - 0.028
        a = c[3:]
        return c[0] + c[1] / den + c[2] / den ** 2 + np.sum(a * w ** (2 * np.arange(len(a)) + 2))
    raise Exception("Bad dispersion function type: " + str(typ))


class CatalogMaterial(IsotropicMaterial):
    kind = "material_from_catalog"

    @classmethod
    def p(cls, lc, ymldict, name="", comment=""):
        """ymldict: dictionary of a refractiveindex.info page (keys DATA -> list of
        {type, coefficients | data, wavelength_range})"""
        obj = cls(lc, name=name, comment=comment)
        obj.annotations["yml_dictionary"] = ymldict
        data = ymldict["DATA"]
        if len(data) > 2:
            raise Exception("Max 2 entries for dispersion allowed - n and k.")
        obj.nk_table = []
        for field in data:
            typ = field["type"]
            if typ.startswith("tabulated"):
                tab = np.array([row.split() for row in field["data"].split("\n") if row.strip()], dtype=float)
                obj.nk_table.append((typ, tab, np.array([tab[:, 0].min(), tab[:, 0].max()])))
            else:
                obj.nk_table.append((typ, np.array(field["coefficients"].split(), dtype=float),
                                     np.array(field["wavelength_range"].split(), dtype=float)))
        return obj

    def get_optical_index(self, x, wave):
        w_um = 1000.0 * wave
        n = 0.0
        for (typ, coeff, rng) in self.nk_table:
            if w_um < rng[0] or w_um > rng[1]:
                raise Exception("wavelength out of range: {0} um\nmust be between {1} um and {2} um"
                                .format(w_um, rng[0], rng[1]))
            if typ == "tabulated n":
                n = n + float(np.interp(w_um, coeff[:, 0], coeff[:, 1]))
            elif typ == "tabulated k":
                n = n + 1j * float(np.interp(w_um, coeff[:, 0], coeff[:, 1]))
            elif typ == "tabulated nk":
                n = n + float(np.interp(w_um, coeff[:, 0], coeff[:, 1])) \
                    + 1j * float(np.interp(w_um, coeff[:, 0], coeff[:, 2]))
            else:
                n = n + float(_n_formula(typ, coeff, w_um))
        return n
