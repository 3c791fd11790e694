"""Convenience builders with the reference's signatures and return values
(pyrateoptics/__init__.py:83-258): ``build_rotationally_symmetric_optical_system``,
``build_simple_optical_element``, ``build_simple_optical_system``; plus ``raytrace`` (:457-465).

Contract (what callers of the reference rely on):

* a prescription is a list of rows ``(surface dict, coordinate-break dict, material, key, options)``; the rows are
  chained: every surface gets a frame ``<key>_lc`` that hangs on the frame of the row before it (the first one on
  ``lc0``), a shape ``<element name>_shape`` (``"shape"`` names the class, default Conic; ``"aperture"`` is taken out
  of the dictionary), a ``Surface`` ``<key>_surf`` registered under ``key`` with the media (before, after);
* ``material`` says what fills the space BEHIND the row's surface: ``None`` -- the background; a number -- a
  ConstantIndexGlass registered as ``constantindexglass_<number>``; a refractiveindex.info page dictionary -- a
  CatalogMaterial; a string -- a glass name for the catalogue under ``material_db_path``; and, beyond the reference:
  a ready ``Material`` object, ``{"eps": 3x3}`` (AnisotropicMaterial), ``{"conrady": (n0, A, B)}`` (ModelGlass);
* the element is returned with its sequence ``(element name, [(key, options), ...])``; the system builder wraps it
  into a system whose object frame is ``"object"``, whose element is ``"stdelem"`` and whose background medium is
  named ``"background"``, and returns ``(system, [sequence])``.

The body is organised around two small tables -- material kinds (``MATERIAL_KINDS``: first matching recogniser
wins) and a row record (``PrescriptionRow``) -- not around the reference's statement order."""
import collections

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

ELEMENT_KEY = "stdelem"
OBJECT_FRAME = "object"
BACKGROUND_NAME = "background"

PrescriptionRow = collections.namedtuple("PrescriptionRow", "shape_spec frame_spec material key options")


# ---- shapes --------------------------------------------------------------------------------------------------
def _shape_class(kind):
    """the class registered for ``"shape": kind`` (surface_shape.accessible_shapes, pyrateoptics/__init__.py:68-80)"""
    try:
        return accessible_shapes["shape_" + kind]
    except KeyError:
        raise Exception("shape shape_%s is outside the HIP engine's scope" % kind)


def _make_shape(frame, spec, element_name, key):
    """(shape, aperture) of one row.  A LinearCombination lists its parts as (coefficient, surface dictionary) pairs;
    they become shapes in the surface's own frame, numbered from 1 (pyrateoptics/__init__.py:146-168)."""
    spec = dict(spec)
    kind = spec.pop("shape", "Conic")
    aperture = spec.pop("aperture", None)
    cls = _shape_class(kind)
    if kind != "LinearCombination":
        return cls.p(frame, name=element_name + "_shape", **spec), aperture
    parts = []
    for (number, (weight, part_spec)) in enumerate(spec.get("list_of_coefficients_and_shapes", ()), start=1):
        part_spec = dict(part_spec)
        part_cls = _shape_class(part_spec.pop("shape", "Conic"))
        parts.append((weight, part_cls.p(frame, name="%s_shape%d" % (element_name, number), **part_spec)))
    return cls.p(frame, name=key + "_linearcombi", list_of_coefficients_and_shapes=parts), aperture


# ---- materials -----------------------------------------------------------------------------------------------
class _Catalogue(object):
    """the glass database, opened when the first glass NAME asks for it"""

    def __init__(self, path):
        self.path = path
        self._gcat = None

    def material(self, frame, long_name):
        if self._gcat is None:
            from .raytracer.material.material_glasscat import GlassCatalog
            self._gcat = GlassCatalog(self.path)
        return self._gcat.create_material_from_long_name(frame, long_name)


def _is_number(value):
    if isinstance(value, (dict, str, Material)):
        return isinstance(value, str) and _parses_as_float(value)
    return _parses_as_float(value)


def _parses_as_float(value):
    try:
        float(value)
    except (TypeError, ValueError):
        return False
    return True


def _page_material(spec, frame, row, catalogue):
    from .raytracer.material.material_glasscat import CatalogMaterial
    return (str(spec.get("SPECS", {}).get("nd", "catalog_" + row.key)), CatalogMaterial.p(frame, spec))


def _conrady_material(spec, frame, row, catalogue):
    from .raytracer.material.material_isotropic import ModelGlass
    key = "modelglass_" + str(spec.get("name", row.key))
    return (key, ModelGlass.p(frame, tuple(spec["conrady"]), name=key))


def _crystal_material(spec, frame, row, catalogue):
    key = "anisotropic_" + row.key
    return (key, AnisotropicMaterial.p(frame, np.array(spec["eps"]), name=key))


# (recogniser, maker): maker(spec, frame, row, catalogue) -> (key in the element's material table, Material)
MATERIAL_KINDS = (
    (lambda m: isinstance(m, Material), lambda m, frame, row, cat: (m.name, m)),
    (lambda m: isinstance(m, dict) and "DATA" in m, _page_material),
    (lambda m: isinstance(m, dict) and "conrady" in m, _conrady_material),
    (lambda m: isinstance(m, dict) and "eps" in m, _crystal_material),
    (_is_number, lambda m, frame, row, cat: ("constantindexglass_" + str(m), ConstantIndexGlass.p(frame, n=float(m)))),
    (lambda m: isinstance(m, str), lambda m, frame, row, cat: (m, cat.material(frame, m))),
)


def _register_material(element, spec, frame, row, catalogue):
    """key of the medium behind the row's surface (None = background), registering it with the element"""
    if spec is None:
        return None
    for (recognises, make) in MATERIAL_KINDS:
        if recognises(spec):
            (key, material) = make(spec, frame, row, catalogue)
            element.addMaterial(key, material)
            return key
    raise Exception("material %r: pass an index, a glass name, a page dictionary or a Material object" % (spec,))


# ---- builders ------------------------------------------------------------------------------------------------
def build_simple_optical_element(lc0, builduplist, material_db_path="", name=""):
    """[(surfdict, coordbreakdict, mat, name, optdict), ...] -> (element, (name, [(surface key, optdict), ...]))"""
    element = OpticalElement.p(lc0, name=name)
    catalogue = _Catalogue(material_db_path)
    parent = lc0.name
    medium_before = None
    sequence = []
    for row in (PrescriptionRow(*entry) for entry in builduplist):
        frame = element.addLocalCoordinateSystem(LocalCoordinates.p(name=row.key + "_lc", **row.frame_spec), refname=parent)
        (shape, aperture) = _make_shape(frame, row.shape_spec, name, row.key)
        medium_after = _register_material(element, row.material, frame, row, catalogue)
        element.addSurface(row.key, Surface.p(frame, name=row.key + "_surf", aperture=aperture, shape=shape),
                           (medium_before, medium_after))
        sequence.append((row.key, row.options))
        (parent, medium_before) = (frame.name, medium_after)
    return (element, (name, sequence))


def build_simple_optical_system(builduplist, material_db_path="", name=""):
    """[(surfdict, coordbreakdict, mat, name, optdict), ...] -> (s, [sequence of the one element "stdelem"])"""
    system = OpticalSystem.p(name=name)
    object_frame = system.addLocalCoordinateSystem(LocalCoordinates.p(name=OBJECT_FRAME, decz=0.0),
                                                   refname=system.rootcoordinatesystem.name)
    (element, sequence) = build_simple_optical_element(object_frame, builduplist, material_db_path=material_db_path,
                                                       name=ELEMENT_KEY)
    system.addElement(ELEMENT_KEY, element)
    system.material_background.set_name(BACKGROUND_NAME)
    return (system, [sequence])


def build_rotationally_symmetric_optical_system(builduplist, **kwargs):
    """[(radius, conic constant, distance to the surface before, mat, name, optdict), ...] -> (s, stdseq): every row
    is a Conic on the axis; a radius of (numerically) zero means a plane"""
    def conic_row(radius, cc, thickness, mat, key, options):
        flat = abs(radius) <= numerical_tolerance
        return ({"shape": "Conic", "curv": 0. if flat else 1. / radius, "cc": cc}, {"decz": thickness}, mat, key, options)
    return build_simple_optical_system([conic_row(*entry) for entry in builduplist], **kwargs)


def raytrace(s, seq, numrays, rays_dict, bundletype="collimated", traceoptions=None, wave=None):
    """convenience entry of the README / demos (pyrateoptics/__init__.py:457-465): one list of
    RayPaths per initial bundle, i.e. ``raytrace(...)[0]`` is what the demos pass to ``draw``"""
    from .raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
    from .raytracer.globalconstants import standard_wavelength
    osa = OpticalSystemAnalysis(s, seq)
    osa.aim(numrays, rays_dict, bundletype=bundletype, wave=standard_wavelength if wave is None else wave)
    return osa.trace(**(traceoptions or {}))
