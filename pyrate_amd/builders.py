"""Convenience builders with the reference's signatures
(pyrateoptics/__init__.py:83-258): build_rotationally_symmetric_optical_system,
build_simple_optical_element, build_simple_optical_system.  Materials may be None
(background), a number (ConstantIndexGlass), a dict {"eps": 3x3}
(AnisotropicMaterial), a dict {"conrady": (n0, A, B)} (ModelGlass), a refractiveindex.info page
dictionary, a ready Material object, or a
glass name looked up in the database under ``material_db_path`` (GlassCatalog)."""
import numpy as np

from .raytracer.globalconstants import numerical_tolerance
from .raytracer.localcoordinates import LocalCoordinates
from .raytracer.material.material import Material
from .raytracer.material.material_anisotropic import AnisotropicMaterial
from .raytracer.material.material_isotropic import ConstantIndexGlass
from .raytracer.optical_element import OpticalElement
from .raytracer.optical_system import OpticalSystem
from .raytracer.surface import Surface
from .raytracer.surface_shape import accessible_shapes


def build_rotationally_symmetric_optical_system(builduplist, **kwargs):
    """builduplist: [(r, cc, thickness, mat, name, optdict), ...] -> (s, stdseq)"""
    out = []
    for (r, cc, thickness, mat, name, optdict) in builduplist:
        curv = 1. / r if abs(r) > numerical_tolerance else 0.
        out.append(({"shape": "Conic", "curv": curv, "cc": cc}, {"decz": thickness}, mat, name, optdict))
    return build_simple_optical_system(out, **kwargs)


def build_simple_optical_element(lc0, builduplist, material_db_path="", name=""):
    elem = OpticalElement.p(lc0, name=name)
    refname = lc0.name
    lastmat = None
    gcat = None
    surflist_for_sequence = []
    for (surfdict, coordbreakdict, mat, surf_name, optdict) in builduplist:
        surfdict = dict(surfdict)
        lc = elem.addLocalCoordinateSystem(
            LocalCoordinates.p(name=surf_name + "_lc", **coordbreakdict), refname=refname)
        shapetype = "shape_" + surfdict.pop("shape", "Conic")
        aperture = surfdict.pop("aperture", None)
        if shapetype not in accessible_shapes:
            raise Exception("shape %s is outside the HIP engine's scope" % shapetype)
        if shapetype == "shape_LinearCombination":
            # the builder's documented special case (pyrateoptics/__init__.py:146-168): the parts come as
            # (coefficient, surface dictionary) pairs and are turned into shapes in the surface's frame
            parts = []
            for (num, (coefficient, part)) in enumerate(surfdict.get("list_of_coefficients_and_shapes", []), 1):
                part = dict(part)
                part_type = "shape_" + part.pop("shape", "Conic")
                if part_type not in accessible_shapes:
                    raise Exception("shape %s is outside the HIP engine's scope" % part_type)
                parts.append((coefficient, accessible_shapes[part_type].p(lc, name="%s_shape%d" % (name, num), **part)))
            shape = accessible_shapes[shapetype].p(lc, name=surf_name + "_linearcombi",
                                                   list_of_coefficients_and_shapes=parts)
        else:
            shape = accessible_shapes[shapetype].p(lc, name=name + "_shape", **surfdict)
        actsurf = Surface.p(lc, name=surf_name + "_surf", aperture=aperture, shape=shape)
        if mat is not None:
            if isinstance(mat, Material):
                key = mat.name
                elem.addMaterial(key, mat)
                mat = key
            elif isinstance(mat, dict) and "DATA" in mat:
                # a refractiveindex.info page dictionary (pyrateoptics/__init__.py:193-196)
                from .raytracer.material.material_glasscat import CatalogMaterial
                key = str(mat.get("SPECS", {}).get("nd", "catalog_" + surf_name))
                elem.addMaterial(key, CatalogMaterial.p(lc, mat))
                mat = key
            elif isinstance(mat, dict) and "conrady" in mat:
                # Conrady model glass n = n0 + A/wave + B/wave**3.5 (prescription importers)
                from .raytracer.material.material_isotropic import ModelGlass
                key = "modelglass_" + str(mat.get("name", surf_name))
                elem.addMaterial(key, ModelGlass.p(lc, tuple(mat["conrady"]), name=key))
                mat = key
            elif isinstance(mat, dict) and "eps" in mat:
                key = "anisotropic_" + surf_name
                elem.addMaterial(key, AnisotropicMaterial.p(lc, np.array(mat["eps"]), name=key))
                mat = key
            else:
                try:
                    n = float(mat)
                except (ValueError, TypeError):
                    if not isinstance(mat, str):
                        raise Exception("material %r: pass an index, a glass name, a page dictionary "
                                        "or a Material object" % (mat,))
                    if gcat is None:
                        from .raytracer.material.material_glasscat import GlassCatalog
                        gcat = GlassCatalog(material_db_path)
                    elem.addMaterial(mat, gcat.create_material_from_long_name(lc, mat))
                else:
                    mat = "constantindexglass_" + str(mat)
                    elem.addMaterial(mat, ConstantIndexGlass.p(lc, n=n))
        elem.addSurface(surf_name, actsurf, (lastmat, mat))
        lastmat = mat
        refname = lc.name
        surflist_for_sequence.append((surf_name, optdict))
    return (elem, (name, surflist_for_sequence))


def build_simple_optical_system(builduplist, material_db_path="", name=""):
    """builduplist: [(surfdict, coordbreakdict, mat, name, optdict), ...] -> (s, stdseq)"""
    s = OpticalSystem.p(name=name)
    lc0 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="object", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    elem_name = "stdelem"
    (elem, elem_seq) = build_simple_optical_element(lc0, builduplist,
                                                    material_db_path=material_db_path, name=elem_name)
    s.addElement(elem_name, elem)
    s.material_background.set_name("background")
    return (s, [elem_seq])


def raytrace(s, seq, numrays, rays_dict, bundletype="collimated", traceoptions=None, wave=None):
    """convenience entry of the README / demos (pyrateoptics/__init__.py:457-465): one list of
    RayPaths per initial bundle, i.e. ``raytrace(...)[0]`` is what the demos pass to ``draw``"""
    from .raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
    from .raytracer.globalconstants import standard_wavelength
    if traceoptions is None:
        traceoptions = {}
    osa = OpticalSystemAnalysis(s, seq)
    osa.aim(numrays, rays_dict, bundletype=bundletype,
            wave=standard_wavelength if wave is None else wave)
    return osa.trace(**traceoptions)
