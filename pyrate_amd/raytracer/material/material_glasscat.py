"""``CatalogMaterial``: isotropic medium described by one page of the refractiveindex.info
database, given as the page's dictionary (reference: raytracer/material/material_glasscat.py:290-529).
Only the scalar dispersion n(wavelength) is needed -- it enters the trace as one number per
(material, bundle) -- so this is host-side arithmetic.  ``GlassCatalog`` is the browser of a
refractiveindex.info database checkout (``library.yml`` + ``data/*.yml``; reference :36-287); the
database itself is an un-vendored submodule of the reference, so the tests build a miniature
database with the same file structure.

Dispersion formulas follow the refractiveindex.info definitions (wavelength in micrometres):
formula 1 Sellmeier, 2 Sellmeier-2, 3 polynomial, 4 RefractiveIndex.INFO, 5 Cauchy, 6 gases,
7 Herzberger, and "tabulated n" (linear interpolation)."""
import os

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


class GlassCatalog(object):
    """shelf -> book -> page index of a refractiveindex.info database directory"""

    def __init__(self, database_basepath, name=""):
        self.name = name
        self.database_basepath = database_basepath
        self.librarydict = self.read_library(os.path.join(database_basepath, "library.yml"))

    @staticmethod
    def read_yml_file(ymlfilename):
        """parsed YAML of the file, [] when it cannot be opened (reference :64-78)"""
        import yaml
        try:
            with open(ymlfilename, "r") as fh:
                return yaml.safe_load(fh)
        except IOError:
            return []

    @staticmethod
    def _index(entries, key):
        """list of dicts -> dict keyed by entry[key]; entries without the key (DIVIDER rows) are
        dropped"""
        out = {}
        for entry in entries or []:
            if key in entry:
                entry = dict(entry)
                out[entry.pop(key)] = entry
        return out

    convert_list_to_dict = _index

    def get_material_dict_nd_vd_pgf(self, nd_value=1.51680, vd_value=64.17, pgf_value=0.5349):
        raise NotImplementedError()          # not implemented in the reference either (:268-287)

    def get_material_dict_nd_vd(self, nd_value=1.51680, vd_value=64.17):
        raise NotImplementedError()

    def get_material_dict_schott_code(self, schott_code=517642):
        raise NotImplementedError()

    def read_library(self, library_yml_filename):
        lib = self._index(self.read_yml_file(library_yml_filename), "SHELF")
        for shelf in lib.values():
            shelf["content"] = self._index(shelf.get("content"), "BOOK")
            for book in shelf["content"].values():
                book["content"] = self._index(book.get("content"), "PAGE")
        return lib

    # -- browsing ------------------------------------------------------------------
    def get_shelves(self):
        return list(self.librarydict.keys())

    def get_books(self, shelf):
        return list(self.librarydict[shelf]["content"].keys())

    def get_pages(self, shelf, book):
        return list(self.librarydict[shelf]["content"][book]["content"].keys())

    def get_page_long_name(self, shelf, book, page):
        return self.librarydict[shelf]["content"][book]["content"][page]["name"]

    def get_dict_of_long_names(self):
        """{long glass name: (shelf, book, page)}; of two pages with one name the later wins"""
        out = {}
        for shelf in self.get_shelves():
            for book in self.get_books(shelf):
                for page in self.get_pages(shelf, book):
                    out[self.get_page_long_name(shelf, book, page)] = (shelf, book, page)
        return out

    def find_pages_with_long_name(self, searchterm):
        return {name: where for (name, where) in self.get_dict_of_long_names().items()
                if name.find(searchterm) != -1}

    # -- pages -----------------------------------------------------------------------
    def get_material_dict(self, shelf, book, page):
        entry = self.librarydict[shelf]["content"][book]["content"][page]
        return self.read_yml_file(os.path.join(self.database_basepath, "data", entry["data"]))

    def material_dict_from_long_name(self, glass_name):
        allpages = self.get_dict_of_long_names()
        if glass_name not in allpages:
            msg = "glass name " + str(glass_name) + " not found."
            similar = self.find_pages_with_long_name(glass_name)
            if similar:
                msg += " Did you mean: " + str(list(similar.keys()))
            else:
                msg += " No glass names containing this string found."
            raise Exception(msg)
        return self.get_material_dict(*allpages[glass_name])

    def create_material_from_long_name(self, localcoordinates, glass_name):
        return CatalogMaterial.p(localcoordinates, self.material_dict_from_long_name(glass_name))


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
