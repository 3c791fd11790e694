"""
The BASELINE.json workloads as surface-table records plus their deterministic
input bundles.  Pure host-side bookkeeping: prescription
numbers -> dict records (the same records ``surface_table.flatten_sequence``
produces from an object graph; ``tests/test_golden_tables.py`` pins that
equivalence against tables flattened from the real reference objects).
"""
import math

import numpy as np

# wavelengths [mm] (raytracer/globalconstants.py:33-47)
FLINE = 0.4861e-3
DLINE = 0.5876e-3
CLINE = 0.6563e-3

_I3 = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]


def rect_grid(nray):
    """RectGrid.getGrid (sampling2d/raster.py:40-60): square raster clipped to the unit disk."""
    n_per_dim = int(round(math.sqrt(nray * 4.0 / math.pi)))
    dx = 1. / n_per_dim
    x1d = np.linspace(-1 + .25 * dx, 1 - .25 * dx, n_per_dim)
    (xpup, ypup) = np.meshgrid(x1d, x1d)
    xpup = np.reshape(xpup, n_per_dim ** 2)
    ypup = np.reshape(ypup, n_per_dim ** 2)
    ind = (xpup ** 2 + ypup ** 2) <= 1
    return (xpup[ind], ypup[ind])


def meridional_fan(nray):
    """MeridionalFan.getGrid (sampling2d/raster.py:127-133)."""
    return (np.zeros(nray), np.linspace(-1, 1, nray))


def _shape_record(surfdict):
    shape = surfdict.get("shape", "Conic")
    if shape == "Conic":
        return {"type": "conic", "curv": float(surfdict.get("curv", 0.0)),
                "cc": float(surfdict.get("cc", 0.0))}
    if shape == "Asphere":
        return {"type": "asphere", "curv": float(surfdict.get("curv", 0.0)),
                "cc": float(surfdict.get("cc", 0.0)),
                "coeffs": [float(a) for a in surfdict.get("coefficients", [])]}
    if shape == "Biconic":
        return {"type": "biconic", "curvx": float(surfdict.get("curvx", 0.0)),
                "curvy": float(surfdict.get("curvy", 0.0)), "ccx": float(surfdict.get("ccx", 0.0)),
                "ccy": float(surfdict.get("ccy", 0.0)),
                "coeffs": [[float(a), float(b)] for (a, b) in surfdict.get("coefficients", [])]}
    if shape == "XYPolynomials":
        return {"type": "xypoly", "normradius": float(surfdict.get("normradius", 1.0)),
                "terms": [[int(i), int(j), float(c)] for (i, j, c) in surfdict.get("coefficients", [])]}
    raise ValueError(shape)


def _aperture_record(ap):
    if ap is None:
        return {"type": "none"}
    ap = dict(ap)
    t = ap.pop("type", None)
    if t == "CircularAperture":
        return {"type": "circular", "minradius": float(ap.get("minradius", 0.0)),
                "maxradius": float(ap.get("maxradius", 1.0))}
    if t == "RectangularAperture":
        return {"type": "rectangular", "width": float(ap.get("width", 1.0)),
                "height": float(ap.get("height", 1.0))}
    return {"type": "none"}


def simple_system_records(builduplist, background_n=1.0):
    """
    Table of a centred system given like build_simple_optical_system's builduplist
    (pyrateoptics/__init__.py:214-258): entries (surfdict, {"decz": thickness_before},
    material_after, name, optdict); material_after is None (background), a float
    (ConstantIndexGlass) or a dict {"eps": 3x3} (AnisotropicMaterial).  All frames are
    z-translations of the root frame.
    """
    records = []
    z = 0.0
    cur = {"type": "isotropic", "n": float(background_n)}
    for (surfdict, coordbreak, mat, _name, optdict) in builduplist:
        z += float(coordbreak.get("decz", 0.0))
        mirror = bool(optdict.get("is_mirror", False))
        if not mirror:
            if mat is None:
                cur = {"type": "isotropic", "n": float(background_n)}
            elif isinstance(mat, dict):
                eps = np.asarray(mat["eps"], dtype=complex)
                cur = {"type": "anisotropic", "eps_re": eps.real.tolist(), "eps_im": eps.imag.tolist()}
            else:
                cur = {"type": "isotropic", "n": float(mat)}
        g = [0.0, 0.0, z]
        records.append({
            "shape": _shape_record(surfdict),
            "B_shape": _I3, "g_shape": g,
            "aperture": _aperture_record(surfdict.get("aperture")),
            "B_ap": _I3, "g_ap": g,
            "interaction": "mirror" if mirror else "refract",
            "material": dict(cur),
            "B_mat": _I3,
        })
    return records


def rotsym_builduplist(tuples):
    """build_rotationally_symmetric_optical_system's (r, cc, thickness, mat, name, opts)
    tuples (pyrateoptics/__init__.py:83-121) -> build_simple_optical_system form."""
    out = []
    for (r, cc, thickness, mat, name, optdict) in tuples:
        curv = 1. / r if abs(r) > 1e-17 else 0.
        out.append(({"shape": "Conic", "curv": curv, "cc": cc}, {"decz": thickness}, mat, name, optdict))
    return out


# ---- config 2: 12-surface double Gauss (Rudolph 1897), d-line indices -------------
# prescription: demos/data/double_gauss_rudolph_1897_v2.spd:6-40, 49
DOUBLE_GAUSS_GLASSES = {
    # name: (nd, nF, nC)  spd:10,13,20
    "N-KF9": (1.52345716953278, 1.53055953979492, 1.52039921283722),
    "LLF1": (1.54813778400421, 1.5565505027771, 1.54456448554993),
    "F5": (1.60341715812683, 1.614617228508, 1.59874367713928),
}
DOUBLE_GAUSS_WAVES_MM = (587.6e-6, 486.1e-6, 656.3e-6, 440e-6, 700e-6)   # spd:5


def conrady_fit(nd, nF, nC):
    """Conrady n = n0 + A/w + B/w^3.5 (ModelGlass, material_isotropic.py:299-309) through
    the three (d, F, C) indices of the SPD file."""
    w = np.array([587.6e-6, 486.1e-6, 656.3e-6])
    m = np.vstack((np.ones(3), 1. / w, 1. / w ** 3.5)).T
    return tuple(np.linalg.solve(m, np.array([nd, nF, nC])))


def double_gauss_tuples(wave=None):
    """(r, cc, thickness, n_after, name, opts) for the 12 traced surfaces."""
    def n(glass):
        (nd, nF, nC) = DOUBLE_GAUSS_GLASSES[glass]
        if wave is None:
            return nd
        (n0, a, b) = conrady_fit(nd, nF, nC)
        return n0 + a / wave + b / wave ** 3.5
    return [
        (43.5219015164416, 0, 0.0, n("N-KF9"), "lens1front", {}),
        (-22.9137468057709, 0, 6.63155126149407, n("LLF1"), "lens1cement", {}),
        (-54.9830148717816, 0, 5.21926274183504, None, "lens1rear", {}),
        (-39.5941875735904, 0, 5.86288269028215, n("F5"), "lens2front", {}),
        (208.80831176475, 0, 4.91037116991389, None, "lens2rear", {}),
        (0, 0, 3.65022739559149, None, "stop", {"is_stop": True}),
        (-208.80831176475, 0, 3.65022739559149, n("F5"), "lens4front", {}),
        (39.5941875735904, 0, 4.91037116991389, None, "lens4rear", {}),
        (54.9830148717816, 0, 5.86288269028215, n("LLF1"), "lens5front", {}),
        (22.9137468057709, 0, 5.21926274183504, n("N-KF9"), "lens5cement", {}),
        (-43.5219015164416, 0, 6.63155126149407, None, "lens5rear", {}),
        (0, 0, 102.415457416957, None, "image", {}),
    ]


def double_gauss_records(wave=None):
    return simple_system_records(rotsym_builduplist(double_gauss_tuples(wave)))


def double_gauss_bundle(nrays, rpup=5.0, z0=-10.0, field_deg=0.0):
    """demo_doublegauss.py:106-117 / 189-213 pattern: RectGrid disk x rpup, collimated,
    E0 = k x ex.  Returns (x0, k0, E0) numpy (3, N)."""
    (px, py) = rect_grid(nrays)
    field = field_deg * math.pi / 180.
    starty = z0 * math.tan(field)     # chief ray through the vertex of surface 1
    o = np.vstack((rpup * px, rpup * py + starty, z0 * np.ones_like(px)))
    k = np.zeros_like(o)
    k[1, :] = math.sin(field)
    k[2, :] = math.cos(field)
    e0 = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T
    return (o, k, np.ascontiguousarray(e0))


def double_gauss_bundle_device(nrays, device, rpup=5.0, z0=-10.0, field_deg=0.0, lo=0, hi=None, uniform=False):
    """the same bundle generated on the GPU (bit-identical to double_gauss_bundle; rays
    [lo, hi) of the raster only -- a rank's shard).  Returns (x0, k0, e0, n_total); with ``uniform``
    (x0, engine.UniformFirst(k, E), None, n_total): the collimated bundle's k and E as one vector each."""
    from . import engine
    field = field_deg * math.pi / 180.
    k = (0.0, math.sin(field), math.cos(field))
    e = (0.0, k[2], -k[1])                        # k x ex
    return engine.collimated_bundle_device(nrays, rpup, (0.0, z0 * math.tan(field), z0), k, e, device,
                                           lo=lo, hi=hi, uniform=uniform)


# ---- the reference's own benchmark (demos/demo_benchmark.py:47-78) -----------------
BENCHMARK_HALF_ANGLE = 10.0 * math.pi / 180.    # "radius": 10 * degree of its divergent bundle


def benchmark_tuples():
    """(r, cc, thickness, n_after, name, opts) of demo_benchmark.py:49-58: four lenses of n = 1.7 / 1.5 around a stop"""
    return [
        (-5.922, 0, 2.0, 1.7, "surf1", {}),
        (-3.160, 0, 3.0, None, "surf2", {}),
        (15.884, 0, 5.0, 1.7, "surf3", {}),
        (-12.756, 0, 3.0, None, "surf4", {}),
        (0, 0, 3.0, None, "stop", {"is_stop": True}),
        (3.125, 0, 2.0, 1.5, "surf5", {}),
        (1.479, 0, 3.0, None, "surf6", {}),
        (0, 0, 19.0, None, "surf7", {}),
    ]


def benchmark_records():
    return simple_system_records(rotsym_builduplist(benchmark_tuples()))


def divergent_bundle(nrays, radius=BENCHMARK_HALF_ANGLE, raster=None, n=1.0):
    """OpticalSystemAnalysis.divergent_bundle (analysis/optical_system_analysis.py:124-165) from the origin into an
    isotropic background: one start point, per-ray unit vectors fanned out over the raster's angles; E = a unit
    vector perpendicular to k (the reference's comes from an eigenproblem; the trace does not depend on which)"""
    (ax, ay) = (raster or rect_grid)(nrays)
    o = np.zeros((3, ax.shape[0]))
    u = np.vstack((np.sin(radius * ax) * np.cos(radius * ay), np.sin(radius * ay), np.cos(radius * ax) * np.cos(radius * ay)))
    k = n * u
    e = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T
    e = e / np.linalg.norm(e, axis=0)
    return (o, k, np.ascontiguousarray(e))


# ---- config 1: cemented doublet (demos/demo_doublet.py:48-101) --------------------
def doublet_builduplist(mat1=1.5168, mat2=1.6727):
    ap = {"type": "CircularAperture", "maxradius": 12.7}
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic", "curv": 1. / 62.8, "aperture": ap}, {"decz": -1.048}, mat1, "front", {}),
        ({"shape": "Conic", "curv": -1. / 45.7, "aperture": ap}, {"decz": 4.0}, mat2, "cement", {}),
        ({"shape": "Conic", "curv": -1. / 128.2, "aperture": ap}, {"decz": 2.5}, None, "rear", {}),
        ({"shape": "Conic"}, {"decz": 97.2}, None, "image", {}),
    ]


def doublet_records():
    return simple_system_records(doublet_builduplist())


def collimated_bundle(nrays, radius, startz, raster=rect_grid, angley=0.0, anglex=0.0, n=1.0):
    """OpticalSystemAnalysis.collimated_bundle (analysis/optical_system_analysis.py:83-122)
    for an isotropic background: k = n * unitvector; E is left to the caller."""
    (px, py) = raster(nrays)
    o = np.vstack((radius * px, radius * py, startz * np.ones_like(px)))
    u = np.zeros_like(o)
    u[0, :] = math.sin(angley) * math.cos(anglex)
    u[1, :] = math.sin(anglex)
    u[2, :] = math.cos(angley) * math.cos(anglex)
    return (o, n * u)


# ---- config 3: even asphere (demos/demo_asphere.py:47-57) -------------------------
def asphere_builduplist(coefficients=(0.0, 1e-7, -1e-10), curv=-1. / 50., cc=-1.):
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic"}, {"decz": 5.0}, 1.5168, "front", {}),
        ({"shape": "Asphere", "curv": curv, "cc": cc, "coefficients": list(coefficients)},
         {"decz": 20.0}, None, "back", {}),
        ({"shape": "Conic"}, {"decz": 100.0}, None, "image", {}),
    ]


def asphere_records(**kw):
    return simple_system_records(asphere_builduplist(**kw))


# ---- the XY-polynomial companion of config 3: demo_asphere.py's geometry with a freeform back surface -------
def xypoly_terms(degree=4, scale=1e-3):
    """XYPolynomials coefficients (surface_shape.py:780-858) [(i, j, c_ij), ...]: the paraboloid -r^2/60 plus small
    terms of every order 2 .. ``degree`` (12 terms for degree 4), sorted by (i, j)"""
    terms = []
    for i in range(degree + 1):
        for j in range(degree + 1 - i):
            if i + j < 2:
                continue
            c = scale * (-1.0) ** (i + j) / (10.0 ** (i + j))
            if (i, j) in ((2, 0), (0, 2)):
                c += -1.0 / 60.0
            terms.append((i, j, c))
    return terms


def xypoly_builduplist(degree=4, scale=1e-3):
    """stop, plane front (n = 1.5168), XY-polynomial back surface, image: demo_asphere.py:47-57 with the asphere
    replaced by a 12-term polynomial freeform (build_simple_optical_system form)"""
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic"}, {"decz": 5.0}, 1.5168, "front", {}),
        ({"shape": "XYPolynomials", "normradius": 1.0, "coefficients": xypoly_terms(degree, scale)},
         {"decz": 20.0}, None, "back", {}),
        ({"shape": "Conic"}, {"decz": 100.0}, None, "image", {}),
    ]


def xypoly_records(degree=4, scale=1e-3):
    return simple_system_records(xypoly_builduplist(degree, scale))


# ---- config 4: anisotropic doublet (demos/demo_anisotropic_doublet.py:55-121) ------
def uniaxial_eps(n_o, n_e, axis):
    axis = np.asarray(axis, dtype=float)
    axis = axis / np.linalg.norm(axis)
    return n_o ** 2 * np.eye(3) + (n_e ** 2 - n_o ** 2) * np.outer(axis, axis)


def aniso_doublet_records(eps1=None, eps2=None):
    if eps1 is None:
        eps1 = 1.5168 ** 2 * np.eye(3)          # demo_anisotropic_doublet.py:92
    if eps2 is None:
        eps2 = 1.6727 ** 2 * np.eye(3)          # :93
    return simple_system_records(doublet_builduplist({"eps": eps1}, {"eps": eps2}))


CALCITE_TILTED = dict(n_o=1.658, n_e=1.486, axis=(0.0, math.sin(0.3), math.cos(0.3)))
