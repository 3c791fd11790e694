"""
Zemax ``.ZMX`` prescription -> ``(OpticalSystem, sequence)`` for the engine (reference:
pyrateoptics/raytracer/io/zmx.py:52-841, ``ZMXParser``).

Host-side only: the file is tokenised once into ``ZmxPrescription`` (header keywords + one
record per ``SURF`` block), from which ``create_optical_system`` builds this package's object
graph with the reference's conventions:

* surface ``i`` sits in a frame ``surf<i>`` whose ``decz`` is the ``DISZ`` of surface ``i-1``
  (zmx.py:612-638); ``SURF 0`` is the object and is not part of the sequence (:816-820);
* ``TYPE`` STANDARD -> Conic, EVENASPH -> Asphere with ``PARM 1..8`` as A2..A16,
  BICONICX -> Biconic (``PARM 1`` = Rx, ``PARM 2`` = ccx), COORDBRK -> a shapeless surface in a
  frame with ``PARM 1..5`` = decx, decy, tiltx, tilty, tiltz [deg] and ``PARM 6`` = order flag
  (zmx.py:788-814), FZERNSAG -> LinearCombination of an Asphere and a decentred ZernikeFringe
  (:723-760), GRID_SAG -> GridSag through the file's sag samples (:761-786);
* ``GLAS MIRROR`` -> ``is_mirror`` with the medium unchanged; a name found in ``matdict`` ->
  that material; model glasses (code 1) -> ConstantIndexGlass(nd) when vd = 0, otherwise a
  Conrady ModelGlass on the normal line (the reference's ``calcCoefficientsFrom_nd_vd`` calls a
  setter that does not exist, material_isotropic.py:325-327; the formulas of its docstring are
  restated in ``conrady_from_nd_vd``);
* ``SQAP`` / ``CLAP`` / ``OBDC`` -> rectangular / circular aperture in a decentred frame
  ``surf<i>_ap`` (zmx.py:676-704); ``STOP`` -> ``is_stop``.

Difference to the reference, on purpose: a ``GLAS`` line with fewer than nine arguments is
ignored there (read_args_for_keyword needs all of them, zmx.py:135-143); here missing trailing
arguments default to 0.
"""
import codecs
import math
import re

from ...surface_table import UnsupportedError
from ..aperture import CircularAperture, RectangularAperture
from ..globalconstants import degree, numerical_tolerance
from ..localcoordinates import LocalCoordinates
from ..material.material_isotropic import ConstantIndexGlass, ModelGlass
from ..optical_element import OpticalElement
from ..optical_system import OpticalSystem
from ..surface import Surface
from ..surface_shape import Asphere, Biconic, Conic, GridSag, LinearCombination, ZernikeFringe


def conrady_from_nd_vd(nd, vd, PgF=None):
    """Conrady coefficients (n0, A [mm], B [mm**3.5]) of a glass given by nd, Abbe number and
    partial dispersion (default: on the normal line, PgF = 0.6438 - 0.001682 vd);
    material_isotropic.py:311-339."""
    if PgF is None:
        PgF = 0.6438 - 0.001682 * vd
    nF_minus_nC = (nd - 1.) / vd
    B = (0.454670392956 * nF_minus_nC * (PgF - 0.445154791693)) * (1e-3) ** 3.5
    A = (1.87513751845 * nF_minus_nC - B / (1e-3) ** 3.5 * 15.2203074842) * 1e-3
    n0 = nd - 1.70194862906e3 * A - 6.43150432188 * (1e3 ** 3.5) * B
    return (n0, A, B)


def _numbers(tokens, types):
    """tokens -> typed list; missing trailing tokens default to the type's zero"""
    out = []
    for (i, t) in enumerate(types):
        out.append(t(tokens[i]) if i < len(tokens) else t())
    return out


class ZmxSurface(object):
    """one SURF block: keyword -> list of token lists, with typed accessors"""

    def __init__(self, number, lines):
        self.number = number
        self.words = {}
        for line in lines:
            tok = line.split()
            if tok:
                self.words.setdefault(tok[0], []).append(tok[1:])

    def has(self, key):
        return key in self.words

    def first(self, key, types, default=None):
        if key not in self.words:
            return default
        return _numbers(self.words[key][0], types)

    def scalar(self, key, typ, default=None):
        r = self.first(key, (typ,))
        return default if r is None else r[0]

    def numbered(self, key, types):
        """all lines of a numbered keyword (PARM n value, XDAT n value ...) -> {n: rest}"""
        out = {}
        for tok in self.words.get(key, []):
            vals = _numbers(tok, (int,) + tuple(types))
            out[vals[0]] = vals[1:] if len(types) > 1 else vals[1]
        return out

    @property
    def type(self):
        return self.scalar("TYPE", str, "STANDARD")

    @property
    def glass(self):
        g = self.first("GLAS", (str, int, int, float, float, float, int, int, int))
        if g is None:
            return None
        keys = ("name", "code", "pickupsurface", "nd", "vd", "pd", "vnd", "vvd", "vpd")
        return dict(zip(keys, g))


class ZmxPrescription(object):
    """tokenised ZMX file: ``header`` (keyword -> list of token lists) and ``surfaces``"""

    def __init__(self, text):
        self.header = {}
        self.surfaces = []
        # a block starts at every line that does not begin with white space (zmx.py:102-109)
        for blk in re.split(r"\n(?=\S)", text.replace("\r\n", "\n").replace("\r", "\n")):
            lines = blk.split("\n")
            tok = lines[0].split()
            if not tok:
                continue
            if tok[0] == "SURF":
                self.surfaces.append(ZmxSurface(int(tok[1]), [l.strip() for l in lines[1:]]))
            else:
                self.header.setdefault(tok[0], []).append(tok[1:])

    @classmethod
    def from_file(cls, filename):
        with open(filename, "rb") as fh:
            raw = fh.read()
        # Zemax writes either plain ASCII / latin-1 or UTF-16 with a byte-order mark (zmx.py:65-75)
        if raw.startswith(codecs.BOM_UTF16_LE) or raw.startswith(codecs.BOM_UTF16_BE):
            text = raw.decode("utf-16")
        else:
            text = raw.decode("latin-1")
        return cls(text)

    def head(self, key, types, default=None):
        if key not in self.header:
            return default
        return _numbers(self.header[key][0], types)


class ZMXParser(object):
    """``ZMXParser(filename).create_optical_system(matdict) -> (OpticalSystem, seq)``"""

    def __init__(self, filename, name=""):
        self.name = name
        self.filename = filename
        self.prescription = ZmxPrescription.from_file(filename)

    # ---- header ---------------------------------------------------------------
    def read_name_and_notes(self):
        hdr = self.prescription.header
        name = " ".join(hdr["NAME"][0]) if "NAME" in hdr else None
        notes = []
        for tok in hdr.get("NOTE", []):
            s = " ".join(tok)
            if s not in ("4", "0"):
                notes.append(s[2:])
        return (name, notes)

    def read_field(self):
        """pupil definition, field points and wavelengths (zmx.py:406-490)"""
        p = self.prescription
        pupildef = {}
        if "ENPD" in p.header:
            pupildef["ENPD"] = p.head("ENPD", (float,))[0]
        for key in ("FNUM", "OBNA"):
            if key in p.header:
                pupildef[key] = p.head(key, (float, int))
        if "FLOA" in p.header:
            pupildef["FLOA"] = " ".join(p.header["FLOA"][0])
        ftyp = p.head("FTYP", (int,) * 7)
        if ftyp is None:
            return {}
        nfield = ftyp[2]
        nwave = ftyp[3]
        xf = [float(v) for v in p.header["XFLN"][0][:nfield]]
        yf = [float(v) for v in p.header["YFLN"][0][:nfield]]
        waves = [_numbers(tok, (int, float, float)) for tok in p.header.get("WAVM", [])][:nwave]
        return {"fieldpoints_type": ftyp[0],
                "telecentric_object_space": bool(ftyp[1]),
                "fieldpoints_number": nfield,
                "wavelengths_number": nwave,
                "fieldpoints_normalization": ["radial", "rectangle"][ftyp[4]],
                "afocal_image_space": bool(ftyp[6]),
                "fieldpoints_pupildef": pupildef,
                "fieldpoints": list(zip(xf, yf)),
                "wavelengths": [(w * 1e-3, weight) for (_, w, weight) in waves]}     # um -> mm

    def create_initial_bundle(self, enpd_default=None, obna_default=None, fnum_default=None):
        """field points -> list of bundle dictionaries for ``OpticalSystemAnalysis.aim``
        (zmx.py:492-540)"""
        fielddict = self.read_field()
        pupildef = fielddict["fieldpoints_pupildef"]
        fields = fielddict["fieldpoints"]
        (radius_height, radius_angle) = (None, None)
        if "ENPD" in pupildef:
            radius_height = radius_angle = 0.5 * pupildef["ENPD"]
        elif "OBNA" in pupildef:
            (obna, flag) = pupildef["OBNA"]
            radius_height = obna
            radius_angle = math.asin(obna) if flag == 0 else 2 * math.asin(obna)
        if fielddict["fieldpoints_type"] == 1:          # object height
            return [{"startx": xf, "starty": yf, "radius": radius_height} for (xf, yf) in fields]
        if fielddict["fieldpoints_type"] == 0:          # angle [deg]
            return [{"anglex": -xf * degree, "angley": -yf * degree, "radius": radius_angle}
                    for (xf, yf) in fields]
        return [{}]

    # ---- system ------------------------------------------------------------------
    def _shape_for(self, surf, lc):
        typ = surf.type
        curv = surf.scalar("CURV", float, 0.0)
        conic = surf.scalar("CONI", float, 0.0)
        parm = surf.numbered("PARM", (float,))
        if typ == "STANDARD":
            return Conic.p(lc, curv=curv, cc=conic)
        if typ == "EVENASPH":
            return Asphere.p(lc, curv=curv, cc=conic,
                             coefficients=[parm.get(1 + i, 0.0) for i in range(8)])
        if typ == "BICONICX":
            rx = parm.get(1, 0.0)
            return Biconic.p(lc, curvy=curv, ccy=conic,
                             curvx=(0.0 if abs(rx) < 1e-16 else 1. / rx), ccx=parm.get(2, 0.0))
        if typ == "FZERNSAG":
            # Zernike fringe sag = even asphere + fringe Zernike series in a decentred frame
            # (PARM 9, 10); XDAT 1 = number of terms, XDAT 2 = norm radius, XDAT 3.. = coefficients
            # (zmx.py:723-760)
            xdat = surf.numbered("XDAT", (float, int, int, float))
            numterms = int(xdat[1][0])
            normradius = xdat[2][0]
            zcoeffs = [xdat[i + 3][0] if (i + 3) in xdat else 0.0 for i in range(numterms)]
            lcz = lc.addChild(LocalCoordinates.p(name="surf%d_zerndec" % surf.number,
                                                 decx=parm.get(9, 0.0), decy=parm.get(10, 0.0)))
            return LinearCombination.p(lc, list_of_coefficients_and_shapes=[
                (1.0, Asphere.p(lc, curv=curv, cc=conic, name="surf%d_zernasph" % surf.number)),
                (1.0, ZernikeFringe.p(lcz, normradius=normradius, coefficients=zcoeffs,
                                      name="surf%d_zernike" % surf.number))])
        if typ == "GRID_SAG":
            # GDAT nx ny dx dy, GARR n sag dzdx dzdy d2zdxdy (only the sag column is used; zmx.py:761-786)
            import numpy as np
            (nx, ny, dx, dy) = surf.first("GDAT", (int, int, float, float))
            garr = surf.numbered("GARR", (float, float, float, float))
            sag = np.array([garr[key][0] for key in sorted(garr.keys())])
            xvec = np.linspace(-nx * dx * 0.5, nx * dx * 0.5, nx)
            yvec = np.linspace(-ny * dy * 0.5, ny * dy * 0.5, ny)
            # samples run along x, row by row from the top (+y) row down.  (The reference reshapes
            # to (nx, ny), which coincides with this for the square grids it can handle and fails
            # in RectBivariateSpline for the others.)
            zgrid = np.flipud(sag.reshape(ny, nx)).T
            return GridSag.p(lc, (xvec, yvec, zgrid))
        return None                                       # COORDBRK and unknown types: plane

    def create_optical_system(self, matdict=None, options=None, elementname="zmxelem"):
        matdict = {} if matdict is None else matdict
        presc = self.prescription
        (name, _) = self.read_name_and_notes()
        s = OpticalSystem.p(name=name or "")
        lc0 = s.addLocalCoordinateSystem(LocalCoordinates.p(name="object", decz=0.0),
                                         refname=s.rootcoordinatesystem.name)
        elem = OpticalElement.p(lc0, name=elementname)
        if matdict:
            for (key, mat) in matdict.items():
                mat.lc = lc0            # one material frame for the whole file (zmx.py:569-573)
                elem.addMaterial(key, mat)
        else:
            # without a material dictionary only mirrors and model glasses can be built (:575-598)
            for surf in presc.surfaces:
                g = surf.glass
                if g is not None and g["name"] != "MIRROR" and g["code"] != 1:
                    return (None, [(elementname, [])])

        refname = lc0.name
        sequence = []
        (lastmat, lastsurfname, surfname) = (None, None, None)
        thickness = 0.0
        for surf in presc.surfaces:
            lastthickness = thickness
            lastsurfname = surfname
            surfname = "surf%d" % surf.number
            thickness = surf.scalar("DISZ", float, 0.0)
            if math.isinf(thickness):
                thickness = 0.0                                  # object at infinity
            opts = {}
            if surf.has("STOP"):
                opts["is_stop"] = True
            lc = s.addLocalCoordinateSystem(LocalCoordinates.p(name=surfname, decz=lastthickness),
                                            refname=refname)
            glass = surf.glass
            matname = None
            if glass is not None:
                if glass["name"] == "MIRROR":
                    matname = lastmat
                    opts["is_mirror"] = True
                if matdict.get(glass["name"]) is not None:
                    matname = glass["name"]
                if glass["code"] == 1:
                    matname = "modelglass_%s_%d" % (surfname, len(elem.materials))
                    if abs(glass["vd"]) < numerical_tolerance:
                        mat = ConstantIndexGlass.p(lc, glass["nd"])
                    else:
                        mat = ModelGlass.p(lc, conrady_from_nd_vd(glass["nd"], glass["vd"]))
                    elem.addMaterial(matname, mat)
            elif surf.type == "COORDBRK":
                matname = lastmat

            obdc = surf.first("OBDC", (float, float))
            lcap = s.addLocalCoordinateSystem(
                LocalCoordinates.p(name=surfname + "_ap", **({} if obdc is None else
                                                             {"decx": obdc[0], "decy": obdc[1]})),
                refname=surfname)
            sqap = surf.first("SQAP", (float, float))
            clap = surf.first("CLAP", (float, float))
            aper = None
            if sqap is not None and clap is None:
                aper = RectangularAperture.p(lcap, width=2 * sqap[0], height=2 * sqap[1])
            elif clap is not None and sqap is None:
                aper = CircularAperture.p(lcap, minradius=clap[0], maxradius=clap[1])

            if surf.type == "COORDBRK":
                parm = surf.numbered("PARM", (float,))
                lc.decx.set_value(parm.get(1, 0.0))
                lc.decy.set_value(parm.get(2, 0.0))
                lc.tiltx.set_value(parm.get(3, 0.0) * degree)
                lc.tilty.set_value(parm.get(4, 0.0) * degree)
                lc.tiltz.set_value(parm.get(5, 0.0) * degree)
                lc.tiltThenDecenter = bool(parm.get(6, 0))
                lc.update()
                actsurf = Surface.p(lc, name=surfname)
            else:
                shape = self._shape_for(surf, lc)
                actsurf = Surface.p(lc, name=surfname, shape=shape, aperture=aper) \
                    if shape is not None else Surface.p(lc, name=surfname)
            if lastsurfname is not None:
                elem.addSurface(surfname, actsurf, (lastmat, matname))
                sequence.append((surfname, opts))
            lastmat = matname
            refname = lc.name
        s.addElement(elementname, elem)
        return (s, [(elementname, sequence)])
