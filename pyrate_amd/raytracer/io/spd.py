"""
WinLens ``.SPD`` prescription -> ``(OpticalSystem, sequence)`` for the engine (reference:
pyrateoptics/raytracer/io/spd.py:41-939, ``SPDFile`` / ``ParaxialSystem`` / ``SPDParser``;
SURVEY.md 8 f4).  Host-side only.

An SPD file is a CSV dump of WinLens' "System Data Editor": one row per component (``"<n>",
"Stop" | name, ..., "lens", <nsurf>, "<?>, <gap after>", ..., medium, maker``), followed for lenses
by a ``LENS n`` ... ``LENS n End`` block of alternating ``Surf`` rows (type, radius of curvature,
free radius) and ``Space`` rows (thickness, glass, maker), each glass with a ``GlassIndex`` row
holding its index at the file's reference wavelengths; further down the paraxial summary rows
(``efl``, ``ObjDist`` / ``ImagDist``, ``l`` / ``l'``, ``Track``, pupil radii, field angles).

``SpdPrescription`` keeps all of that; ``SPDParser.create_optical_system`` turns it into the
``(r, cc, thickness, material, name, options)`` tuples of
``build_rotationally_symmetric_optical_system`` with the reference's conventions (spd.py:873-929):
every surface carries the thickness of the space *before* it, the first one 0, and an image plane
is appended at the paraxial image distance.

Glasses: with ``options={"gcat": GlassCatalog, "db_path": ...}`` the glass names go to the
catalogue like in the reference (which cannot run without one).  Without a catalogue -- the
refractiveindex.info database is not part of the reference checkout -- the file's own
``GlassIndex`` values are used: a Conrady ``ModelGlass`` through the (wavelength, index) pairs
of the first three reference wavelengths (exact at d, F, C), or a ``ConstantIndexGlass`` when
the file gives a single index.  ``matdict`` (glass name -> index | Material | page dict)
overrides both.
"""
# materials are handed to the builder as specifications ({"conrady": (n0, A, B), "name": glass});
# the builder creates the Material objects in the frames of the surfaces they start at
import csv
import math
import re

import numpy as np

from ...surface_table import UnsupportedError

PLANE_RADIUS = 1e16          # "radius" of plane surfaces in the tuples (spd.py:249, 929)


def _number(text):
    """leading float of a cell ("587.6", ".5", "5.026905E-02", "12 mm"); 0.0 if there is none"""
    m = re.match(r"\s*[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?", text)
    return float(m.group(0)) if m else 0.0


def _label(cell):
    return cell.replace('"', " ").strip()


class SpdSurface(object):
    def __init__(self, name, kind="sphere", radius=PLANE_RADIUS, free_radius=0.0, is_stop=False):
        self.name = name
        self.kind = kind
        self.radius = radius            # radius of curvature [mm]; PLANE_RADIUS for planes
        self.free_radius = free_radius
        self.is_stop = is_stop


class SpdSpace(object):
    def __init__(self, name, thickness=0.0, medium="air", maker="", indices=None):
        self.name = name
        self.thickness = thickness
        self.medium = medium
        self.maker = maker
        self.indices = indices or []    # index at the file's reference wavelengths (GlassIndex row)

    @property
    def is_air(self):
        return self.medium.lower() in ("air", "")


class SpdComponent(object):
    """a lens (surfaces separated by glass spaces) or a bare stop, plus the gap behind it;
    ``surfaces[j]`` is followed by ``spaces[j]``"""

    def __init__(self, name):
        self.name = name
        self.surfaces = []
        self.spaces = []


class SpdPrescription(object):
    SUMMARY = {"efl": ("efl",), "Mag": ("magnification",), "Track": ("track",),
               "Entr Pup Rad": ("entpup_rad", "Exit Pup Rad", "expup_rad"),
               "ObjDist": ("obj_dist", "ImagDist", "img_dist"),
               "ObjAngle": ("obj_angle", "ObjHeight", "obj_height"),
               "ImagAngle": ("img_angle", "ImagHeight", "img_height"),
               "l": ("l", "l'", "ldash"),
               "ObjNa": ("obj_na", "ImagNa", "img_na")}

    def __init__(self, rows):
        self.components = []
        self.wavelengths_nm = []
        self.stop_radius = None
        for key in ("efl", "magnification", "track", "entpup_rad", "expup_rad", "obj_dist", "img_dist",
                    "obj_angle", "obj_height", "img_angle", "img_height", "l", "ldash", "obj_na", "img_na"):
            setattr(self, key, None)
        self._parse(list(rows))

    @classmethod
    def from_file(cls, filename):
        with open(filename, newline="") as fh:
            return cls(csv.reader(fh, delimiter=","))

    # ---- parsing -----------------------------------------------------------------
    def _parse(self, rows):
        i = 0
        while i < len(rows):
            row = rows[i]
            i += 1
            if not row:
                continue
            head = _label(row[0])
            if re.match(r"[0-9]+", head) and len(row) >= 15:
                # component row of the System Data Editor (spd.py:238-285)
                gap = SpdSpace("Gap", _number(row[6].split(",")[1]) if "," in row[6] else 0.0,
                               row[13].strip(), row[14].strip())
                if _label(row[1]) == "Stop":
                    if not self.components:
                        self.components.append(SpdComponent("NoName"))       # front stop
                    comp = self.components[-1]
                    comp.surfaces.append(SpdSurface("Stop", "plane", PLANE_RADIUS, 5.0, is_stop=True))
                    comp.spaces.append(gap)
                if _label(row[4]) == "lens":
                    (comp, i) = self._parse_lens(rows, i)
                    comp.spaces.append(SpdSpace("Gap", gap.thickness, gap.medium, gap.maker))
                    self.components.append(comp)
                continue
            if head == "Waveband":
                self.wavelengths_nm = [_number(c) for c in row[1:] if c.strip()]
            if head == "ObjNa" and len(row) >= 6 and _label(row[4]) == "Stop Rad":
                self.stop_radius = _number(row[5])
            if head in self.SUMMARY:
                spec = self.SUMMARY[head]
                setattr(self, spec[0], _number(row[1]))
                if len(spec) == 3 and len(row) >= 4 and _label(row[2]) == spec[1]:
                    setattr(self, spec[2], _number(row[3]))

    @staticmethod
    def _parse_lens(rows, i):
        if i < len(rows) and rows[i] and _label(rows[i][0]) == "SpaceIndex":
            i += 1
        m = re.match(r"\s*LENS\s*[0-9]+", rows[i][0]) if i < len(rows) and rows[i] else None
        comp = SpdComponent(_label(rows[i][0]) if m else "NoName")
        if not m:
            return (comp, i)
        i += 1
        end = re.compile(r"\s*%s\s*End" % re.escape(comp.name))
        while i < len(rows):
            row = rows[i]
            i += 1
            if not row:
                continue
            if end.match(row[0]):
                break
            head = _label(row[0])
            if re.match(r"Surf\s*[0-9]+", head):
                kind = _label(row[1])
                if kind not in ("sphere", "plane"):
                    raise UnsupportedError("SPD surface type %r (%s %s)" % (kind, comp.name, head))
                comp.surfaces.append(SpdSurface(head, kind, _number(row[3]), _number(row[9])))
            elif re.match(r"Space\s*[0-9]+", head):
                comp.spaces.append(SpdSpace(head, _number(row[1]), _label(row[3]), _label(row[4])))
            elif head == "GlassIndex" and comp.spaces:
                comp.spaces[-1].indices = [_number(c) for c in row[1:] if c.strip()]
        return (comp, i)

    # ---- derived first-order quantities (spd.py:333-415) -----------------------------
    def pp_obj(self):
        """object-side principal plane w.r.t. the first surface"""
        return -self.l + self.obj_dist

    def pp_img(self):
        """image-side principal plane w.r.t. the last surface"""
        return -self.ldash + self.img_dist

    def thick(self):
        """first to last surface"""
        return self.track - (self.img_dist - self.obj_dist)

    def entpup(self):
        return self.obj_height / math.tan(self.obj_angle / 180. * math.pi) + self.obj_dist

    def expup(self):
        return -self.img_height / math.tan(self.img_angle / 180. * math.pi) + self.img_dist

    def distance_entpup_objplane(self):
        return -self.entpup() + self.obj_dist

    def distance_expup_imgplane(self):
        return -self.expup() + self.img_dist

    def objNA(self):
        return math.sin(-math.atan(self.entpup_rad / self.distance_entpup_objplane()))

    def imgNA(self):
        return math.sin(math.atan(self.expup_rad / self.distance_expup_imgplane()))

    # ---- surface list ------------------------------------------------------------------
    def surface_rows(self):
        """[(radius, thickness of the space BEHIND the surface, that space, name, is_stop)]"""
        out = []
        for comp in self.components:
            for (j, surf) in enumerate(comp.surfaces):
                out.append((surf.radius, comp.spaces[j].thickness, comp.spaces[j],
                            comp.name + " " + surf.name, surf.is_stop))
        return out


class ParaxialSystem(object):
    """Thick-lens model with WinLens' conventions: origin at the first surface, +z towards the
    image (spd.py:478-760)."""

    def __init__(self, effective_focal_length, system_thickness, principal_plane_obj,
                 principal_plane_img, entrance_pupil, exit_pupil, exit_pupil_rad, object_distance,
                 object_angle):
        self.object_dist = object_distance
        self.efl = effective_focal_length
        self.pp_obj = principal_plane_obj
        self.pp_img = principal_plane_img
        self.entpup = entrance_pupil
        self.expup = exit_pupil
        self.expup_rad = exit_pupil_rad
        self.thick = system_thickness
        self.obj_angle = object_angle
        self.spd = None
        self.NAimg = abs(math.sin(math.atan(self.expup_rad / (self.img_dist() - self.expup))))
        self.NAobj = self.NAimg * self.field_size_img() / self.field_size_obj()
        self.entpup_rad = abs(math.tan(math.asin(self.NAobj)) * (self.obj_dist() - self.entpup))

    @classmethod
    def create(cls, spd_filename):
        spd = SpdPrescription.from_file(spd_filename)
        psys = cls(spd.efl, spd.thick(), spd.pp_obj(), spd.pp_img(), spd.entpup(), spd.expup(),
                   spd.expup_rad, spd.obj_dist, spd.obj_angle)
        psys.spd = spd
        return psys

    def obj_dist(self):
        return self.object_dist

    def rear_focus(self):
        return self.efl + self.pp_img

    def front_focus(self):
        return -(self.efl - self.pp_obj)

    def img_dist(self, obj_dist=None):
        """paraxial image distance w.r.t. the last surface (Gaussian lens equation between the
        principal planes)"""
        if obj_dist is None:
            obj_dist = self.obj_dist()
        z = obj_dist - self.pp_obj
        return 1.0 / (1.0 / self.efl + 1.0 / z) + self.pp_img

    def image_points(self, field_pts):
        zi = self.img_dist() - self.pp_img
        zo = self.pp_obj - self.obj_dist()
        return -np.asarray(field_pts) / zo * zi

    def field_size_obj(self):
        return abs(math.tan(self.obj_angle / 180. * math.pi) * (self.obj_dist() - self.entpup))

    def field_size_img(self):
        return float(self.image_points(np.array([[self.field_size_obj()], [0.0]]))[0, 0])

    def img_angle(self):
        return math.atan2(self.field_size_img(), self.img_dist() - self.expup) / math.pi * 180.

    def mag(self):
        return (self.img_dist() - self.pp_img) / (self.obj_dist() - self.pp_obj)

    def distance_exit_pupil_image_plane(self):
        return self.img_dist() - self.expup


def conrady_through(waves_mm, indices):
    """(n0, A, B) of n = n0 + A/w + B/w**3.5 through three (wavelength [mm], index) pairs"""
    w = np.asarray(waves_mm[:3], dtype=float)
    m = np.stack((np.ones(3), 1. / w, w ** -3.5), axis=1)
    return tuple(float(v) for v in np.linalg.solve(m, np.asarray(indices[:3], dtype=float)))


class SPDParser(object):
    """``SPDParser(filename).create_optical_system(...) -> (OpticalSystem, seq)``; ``psys`` is the
    ``ParaxialSystem`` of the file (``psys.spd`` the parsed prescription)."""

    def __init__(self, filename, name=""):
        self.name = name
        self.psys = ParaxialSystem.create(filename)

    def _material(self, space, matdict, use_catalog):
        if space.medium in matdict:
            return matdict[space.medium]
        if space.is_air:
            return 1.0                               # spd.py:883-885
        if use_catalog:
            return space.medium                      # resolved by name through material_db_path
        spd = self.psys.spd
        n = [v for v in space.indices if v > 0]
        if not n:
            raise UnsupportedError("glass %r has no GlassIndex row and no catalogue was given" % space.medium)
        if len(n) >= 3 and len(spd.wavelengths_nm) >= 3:
            return {"conrady": conrady_through([w * 1e-6 for w in spd.wavelengths_nm], n),
                    "name": space.medium}
        return n[0]

    def components(self, matdict=None, use_catalog=False):
        """the builduplist of build_rotationally_symmetric_optical_system"""
        matdict = {} if matdict is None else matdict
        rows = self.psys.spd.surface_rows()
        out = []
        shared = {}
        for (i, (radius, _, space, name, is_stop)) in enumerate(rows):
            before = rows[i - 1][1] if i > 0 else 0.0            # thickness re-association (:913-925)
            if space.medium not in shared:
                shared[space.medium] = self._material(space, matdict, use_catalog)
            out.append((radius, 0.0, before, shared[space.medium], name, {"is_stop": is_stop}))
        out.append((PLANE_RADIUS, 0.0, self.psys.img_dist(), None, "img", {"is_stop": False}))
        return out

    def create_optical_system(self, matdict=None, options=None):
        from ...builders import build_rotationally_symmetric_optical_system
        options = {} if options is None else options
        use_catalog = options.get("gcat") is not None or bool(options.get("db_path"))
        comps = self.components(matdict, use_catalog)
        return build_rotationally_symmetric_optical_system(
            comps, name=self.name, material_db_path=options.get("db_path", "") or "")
