"""
The optical systems + bundles of the golden cases, written ONCE against an ``api``
namespace so that the very same construction code runs against
  * the real reference (oracle/make_golden.py, api = pyrateoptics classes) and
  * this repo's host mirror (tests, api = pyrate_amd classes).
"""
import math
import types

import numpy as np

DLINE = 0.5876e-3


def mirror_api():
    from pyrate_amd import builders
    from pyrate_amd.raytracer import aperture, localcoordinates, optical_element, optical_system, ray, \
        surface, surface_shape
    from pyrate_amd.raytracer.material import material_anisotropic, material_isotropic
    return types.SimpleNamespace(
        OpticalSystem=optical_system.OpticalSystem, OpticalElement=optical_element.OpticalElement,
        LocalCoordinates=localcoordinates.LocalCoordinates, Surface=surface.Surface,
        Conic=surface_shape.Conic, Asphere=surface_shape.Asphere, XYPolynomials=surface_shape.XYPolynomials,
        Biconic=surface_shape.Biconic, ZernikeFringe=surface_shape.ZernikeFringe,
        ZernikeANSI=surface_shape.ZernikeANSI, LinearCombination=surface_shape.LinearCombination,
        GridSag=surface_shape.GridSag,
        CircularAperture=aperture.CircularAperture, RectangularAperture=aperture.RectangularAperture,
        ConstantIndexGlass=material_isotropic.ConstantIndexGlass, ModelGlass=material_isotropic.ModelGlass,
        AnisotropicMaterial=material_anisotropic.AnisotropicMaterial, RayBundle=ray.RayBundle,
        build_simple_optical_system=builders.build_simple_optical_system,
        build_rotationally_symmetric_optical_system=builders.build_rotationally_symmetric_optical_system)


def rect_grid(nray):
    """RectGrid.getGrid (sampling2d/raster.py:40-60)"""
    n_per_dim = int(round(math.sqrt(nray * 4.0 / math.pi)))
    dx = 1. / n_per_dim
    x1d = np.linspace(-1 + .25 * dx, 1 - .25 * dx, n_per_dim)
    (xpup, ypup) = np.meshgrid(x1d, x1d)
    xpup = np.reshape(xpup, n_per_dim ** 2)
    ypup = np.reshape(ypup, n_per_dim ** 2)
    ind = (xpup ** 2 + ypup ** 2) <= 1
    return (xpup[ind], ypup[ind])


def disk_bundle_arrays(nrays, rpup, z0, field_deg=0.0, efield="kxex"):
    (px, py) = rect_grid(nrays)
    field = field_deg * math.pi / 180.
    starty = z0 * math.tan(field)
    o = np.vstack((rpup * px, rpup * py + starty, z0 * np.ones_like(px)))
    k = np.zeros_like(o)
    k[1, :] = math.sin(field)
    k[2, :] = math.cos(field)
    e0 = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T if efield == "kxex" else None
    return (o, k, e0)


def doublet(api, mat1=None, mat2=None, stop_radius=None):
    """demos/demo_doublet.py:48-101 / demo_anisotropic_doublet.py:55-121, object by object.
    mat1 / mat2: callables lc -> Material (default BK7 / SF5 constant-index glasses)."""
    s = api.OpticalSystem.p(name='os')
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="stop", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="surf1", decz=-1.048), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="surf2", decz=4.0), refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="surf3", decz=2.5), refname=lc2.name)
    lc4 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="image", decz=97.2), refname=lc3.name)
    stopsurf = api.Surface.p(lc0, name="stopsurf",
                             aperture=(api.CircularAperture.p(lc0, maxradius=stop_radius)
                                       if stop_radius else None))
    frontsurf = api.Surface.p(lc1, name="frontsurf", shape=api.Conic.p(lc1, curv=1. / 62.8),
                              aperture=api.CircularAperture.p(lc1, maxradius=12.7))
    cementsurf = api.Surface.p(lc2, name="cementsurf", shape=api.Conic.p(lc2, curv=-1. / 45.7),
                               aperture=api.CircularAperture.p(lc2, maxradius=12.7))
    rearsurf = api.Surface.p(lc3, name="rearsurf", shape=api.Conic.p(lc3, curv=-1. / 128.2),
                             aperture=api.CircularAperture.p(lc3, maxradius=12.7))
    image = api.Surface.p(lc4, name="imagesurf")
    elem = api.OpticalElement.p(lc0, name="thorlabs_AC_254-100-A")
    m1 = mat1(lc1) if mat1 else api.ConstantIndexGlass.p(lc1, n=1.5168)
    m2 = mat2(lc2) if mat2 else api.ConstantIndexGlass.p(lc2, n=1.6727)
    elem.addMaterial("mat1", m1)
    elem.addMaterial("mat2", m2)
    elem.addSurface("stop", stopsurf, (None, None))
    elem.addSurface("front", frontsurf, (None, "mat1"))
    elem.addSurface("cement", cementsurf, ("mat1", "mat2"))
    elem.addSurface("rear", rearsurf, ("mat2", None))
    elem.addSurface("image", image, (None, None))
    s.addElement("AC254-100", elem)
    seq = [("AC254-100", [("stop", {"is_stop": True}), ("front", {}), ("cement", {}),
                          ("rear", {}), ("image", {})])]
    return (s, seq)


def aniso_doublet(api, eps1, eps2, stop_radius=None):
    return doublet(api, lambda lc: api.AnisotropicMaterial.p(lc, eps1, name="crystal1"),
                   lambda lc: api.AnisotropicMaterial.p(lc, eps2, name="crystal2"), stop_radius)


def crystal_mirror(api, eps, tilt_deg=10.0):
    """crystal slab whose rear face is a mirror INSIDE the crystal (reflection in an anisotropic
    medium: the two backward solutions, material_anisotropic.py:115-155); the folded beam leaves
    the crystal into the background at a tilted exit plane.  Geometry after
    demos/demo_anisotropic_mirror.py (flat faces, 10 deg tilts)."""
    t = tilt_deg * math.pi / 180.
    s = api.OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="stop", decz=1.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="front", decz=10.0), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="rear", decz=5.0, tiltx=t),
                                     refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="exit", decz=-5.0, tiltx=-t),
                                     refname=lc2.name)
    elem = api.OpticalElement.p(lc0, name="slab")
    elem.addMaterial("crystal", api.AnisotropicMaterial.p(lc1, eps, name="crystal"))
    elem.addSurface("stop", api.Surface.p(lc0), (None, None))
    elem.addSurface("front", api.Surface.p(lc1, shape=api.Conic.p(lc1, curv=0.0),
                                           aperture=api.CircularAperture.p(lc1, maxradius=10.0)),
                    (None, "crystal"))
    elem.addSurface("rear", api.Surface.p(lc2, shape=api.Conic.p(lc2, curv=0.0),
                                          aperture=api.CircularAperture.p(lc2, maxradius=10.0)),
                    ("crystal", "crystal"))
    elem.addSurface("exit", api.Surface.p(lc3), ("crystal", None))
    s.addElement("slab", elem)
    seq = [("slab", [("stop", {}), ("front", {}), ("rear", {"is_mirror": True}), ("exit", {})])]
    return (s, seq)


def crystal_inside(api, eps, tilt_deg=10.0, mirror=True, eps2=None):
    """a sequence that never leaves the crystal: refraction into a slab, its tilted rear face (a mirror inside
    the crystal, or an interface to a second crystal), and an end plane whose medium is still a crystal.  With a
    complex (absorbing) eps every wave vector inside is complex and every interface doubles the rays; nothing
    downstream depends on the arbitrary E basis an isotropic medium would bring in."""
    t = tilt_deg * math.pi / 180.
    s = api.OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="stop", decz=1.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="front", decz=10.0), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="rear", decz=5.0, tiltx=t),
                                     refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="end", decz=-5.0 if mirror else 5.0, tiltx=-t),
                                     refname=lc2.name)
    elem = api.OpticalElement.p(lc0, name="slab")
    elem.addMaterial("crystal", api.AnisotropicMaterial.p(lc1, eps, name="crystal"))
    elem.addMaterial("crystal2", api.AnisotropicMaterial.p(lc2, eps if eps2 is None else eps2, name="crystal2"))
    behind = "crystal" if mirror else "crystal2"
    elem.addSurface("stop", api.Surface.p(lc0), (None, None))
    elem.addSurface("front", api.Surface.p(lc1, shape=api.Conic.p(lc1, curv=0.01),
                                           aperture=api.CircularAperture.p(lc1, maxradius=10.0)),
                    (None, "crystal"))
    elem.addSurface("rear", api.Surface.p(lc2, shape=api.Conic.p(lc2, curv=-0.02),
                                          aperture=api.CircularAperture.p(lc2, maxradius=10.0)),
                    ("crystal", behind))
    elem.addSurface("end", api.Surface.p(lc3), (behind, behind))
    s.addElement("slab", elem)
    seq = [("slab", [("stop", {}), ("front", {}), ("rear", {"is_mirror": mirror}), ("end", {})])]
    return (s, seq)


def absorbing_detector(api, n_abs=3.9 + 0.02j):
    """a singlet in front of an ABSORBING isotropic medium (complex refractive index, e.g. a silicon detector) behind
    the last surface: every hit point is real, the last bundle's wave vector is complex
    (material_isotropic.py:137-161 with a complex index)"""
    s = api.OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="stop", decz=1.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="front", decz=5.0), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="back", decz=4.0), refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="detector", decz=20.0, tiltx=0.1), refname=lc2.name)
    elem = api.OpticalElement.p(lc0, name="lens")
    elem.addMaterial("glass", api.ConstantIndexGlass.p(lc1, 1.5168))
    elem.addMaterial("absorber", api.ConstantIndexGlass.p(lc3, n_abs))
    elem.addSurface("stop", api.Surface.p(lc0), (None, None))
    elem.addSurface("front", api.Surface.p(lc1, shape=api.Conic.p(lc1, curv=0.02),
                                           aperture=api.CircularAperture.p(lc1, maxradius=10.0)), (None, "glass"))
    elem.addSurface("back", api.Surface.p(lc2, shape=api.Conic.p(lc2, curv=-0.03)), ("glass", None))
    elem.addSurface("detector", api.Surface.p(lc3, shape=api.Conic.p(lc3, curv=-0.01)), (None, "absorber"))
    s.addElement("lens", elem)
    seq = [("lens", [("stop", {}), ("front", {}), ("back", {}), ("detector", {})])]
    return (s, seq)


def tilted(api):
    """decentred / tilted frames (both tilt orders), a tilted material frame, a rectangular
    aperture in its own rotated frame, an annular circular aperture, a ModelGlass."""
    s = api.OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(
        api.LocalCoordinates.p(name="s1", decz=10.0, decx=0.3, tiltx=0.05, tilty=-0.03), refname=lc0.name)
    lc1ap = s.addLocalCoordinateSystem(
        api.LocalCoordinates.p(name="s1ap", decy=0.4, tiltz=0.3), refname=lc1.name)
    lc1m = s.addLocalCoordinateSystem(
        api.LocalCoordinates.p(name="s1mat", tiltx=0.2, tiltz=-0.1), refname=lc1.name)
    lc2 = s.addLocalCoordinateSystem(
        api.LocalCoordinates.p(name="s2", decz=6.0, decy=-0.2, tiltx=-0.04, tiltz=0.1,
                               tiltThenDecenter=1), refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="img", decz=60.0, tilty=0.02),
                                     refname=lc2.name)
    s1 = api.Surface.p(lc1, shape=api.Conic.p(lc1, curv=1. / 40., cc=-0.5),
                       aperture=api.RectangularAperture.p(lc1ap, width=11.0, height=9.0))
    s2 = api.Surface.p(lc2, shape=api.Conic.p(lc2, curv=-1. / 55., cc=0.3),
                       aperture=api.CircularAperture.p(lc2, maxradius=5.5, minradius=0.8))
    s3 = api.Surface.p(lc3)
    elem = api.OpticalElement.p(lc0, name="tilted")
    elem.addMaterial("glass", api.ModelGlass.p(lc1m))
    elem.addSurface("s1", s1, (None, "glass"))
    elem.addSurface("s2", s2, ("glass", None))
    elem.addSurface("img", s3, (None, None))
    s.addElement("tilted", elem)
    seq = [("tilted", [("s1", {}), ("s2", {}), ("img", {})])]
    return (s, seq)


def xypoly_builduplist():
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic", "curv": 1. / 80.}, {"decz": 5.0}, 1.5168, "front", {}),
        ({"shape": "XYPolynomials", "normradius": 10.0,
          "coefficients": [(0, 2, -0.12), (2, 0, -0.1), (2, 1, 0.01), (0, 3, -0.004),
                           (4, 0, 0.002), (1, 1, 0.003)]},
         {"decz": 12.0}, None, "back", {}),
        ({"shape": "Conic"}, {"decz": 80.0}, None, "image", {}),
    ]


ZERNIKE_FRINGE_COEFFS = [0.0, 0.02, -0.015, 0.04, -0.03, 0.02, 0.012, -0.01, 0.008, 0.0, 0.004, -0.003,
                         0.0025, 0.002, -0.0015, 0.001, 0.0, 0.0008, -0.0006, 0.0005, 0.0, 0.0004, -0.0003,
                         0.0002, 0.00015]
ZERNIKE_ANSI_COEFFS = [0.0, -0.01, 0.02, 0.015, 0.03, -0.02, 0.006, -0.005, 0.004, 0.003, 0.002, -0.0015,
                       0.001, 0.0008, -0.0005]


# rotationally symmetric fringe terms only (Z4 defocus, Z9 / Z16 / Z25 spherical): for m = 0 the reference's Zernike
# gradient IS the derivative of its sag (to 6e-12, checked against the reference), so its normals are a parity target
ZERNIKE_FRINGE_SYMMETRIC = [0.04 if j == 4 else 0.008 if j == 9 else 0.001 if j == 16 else 0.00015 if j == 25 else 0.0
                            for j in range(1, 26)]


def zernike_builduplist(indexing="Fringe", symmetric=False):
    """a freeform back surface given as a Zernike series (25 fringe / 15 ANSI terms, up to 8th order)"""
    coeffs = ZERNIKE_FRINGE_COEFFS if indexing == "Fringe" else ZERNIKE_ANSI_COEFFS
    if symmetric:
        coeffs = ZERNIKE_FRINGE_SYMMETRIC
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic", "curv": 1. / 60.}, {"decz": 5.0}, 1.5168, "front", {}),
        ({"shape": "Zernike" + indexing, "normradius": 9.0, "coefficients": coeffs},
         {"decz": 10.0}, None, "back", {}),
        ({"shape": "Conic"}, {"decz": 70.0}, None, "image", {}),
    ]


def zernike_combination_system(api):
    """Zemax-style "Zernike fringe sag" mirror: LinearCombination of a conic asphere and a fringe
    Zernike series in a decentred frame (the object graph zmx.py:723-760 builds), used in reflection"""
    s = api.OpticalSystem.p(name="zcombo")
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="mirror", decz=40.0, tiltx=0.12), refname=lc0.name)
    lcz = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="mirror_zern", decx=0.7, decy=-1.1),
                                     refname=lc1.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="img", decz=-35.0, tiltx=0.12), refname=lc1.name)
    shape = api.LinearCombination.p(lc1, list_of_coefficients_and_shapes=[
        (1.0, api.Asphere.p(lc1, curv=-1. / 90., cc=-0.8, coefficients=[0.0, 2e-6])),
        (1.0, api.ZernikeFringe.p(lcz, normradius=12.0, coefficients=[0.0, 0.01, -0.02, 0.03, 0.015, -0.01,
                                                                      0.006, 0.004, -0.003]))])
    elem = api.OpticalElement.p(lc0, name="zc")
    elem.addSurface("mirror", api.Surface.p(lc1, shape=shape,
                                            aperture=api.CircularAperture.p(lc1, maxradius=11.0)), (None, None))
    elem.addSurface("img", api.Surface.p(lc2), (None, None))
    s.addElement("zc", elem)
    return (s, [("zc", [("mirror", {"is_mirror": True}), ("img", {})])])


def rotated_combination_system(api):
    """a freeform lens surface as the reference composes one: LinearCombination of a conic asphere and an XY
    polynomial whose frame is decentred AND rotated about the surface's axis (tiltz) -- surface_shape.py:709-748
    evaluates every part in its own frame -- in refraction, followed by a plane back surface and an image plane"""
    s = api.OpticalSystem.p(name="rotcombo")
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="front", decz=12.0, tilty=0.05), refname=lc0.name)
    lcp = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="front_poly", decx=0.8, decy=-0.5, tiltz=0.6),
                                     refname=lc1.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="back", decz=6.0), refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="img", decz=40.0), refname=lc2.name)
    shape = api.LinearCombination.p(lc1, list_of_coefficients_and_shapes=[
        (0.9, api.Asphere.p(lc1, curv=1. / 35., cc=-0.6, coefficients=[0.0, 3e-6])),
        (1.2, api.XYPolynomials.p(lcp, normradius=10.0, coefficients=[(1, 0, 0.02), (0, 2, 0.05), (2, 1, -0.04),
                                                                      (3, 0, 0.03), (1, 3, 0.02), (0, 4, -0.015)]))])
    elem = api.OpticalElement.p(lc0, name="rc")
    elem.addMaterial("glass", api.ConstantIndexGlass.p(lc1, 1.6))
    elem.addSurface("front", api.Surface.p(lc1, shape=shape, aperture=api.CircularAperture.p(lc1, maxradius=9.0)),
                    (None, "glass"))
    elem.addSurface("back", api.Surface.p(lc2), ("glass", None))
    elem.addSurface("img", api.Surface.p(lc3), (None, None))
    s.addElement("rc", elem)
    return (s, [("rc", [("front", {}), ("back", {}), ("img", {})])])


def evanescent_slab(api):
    """plane crystal slab (uniaxial, n_o = 1.35, n_e = 2.1, axis along x) immersed in a dense
    medium (n = 1.9): beyond ~45 degrees one of the two transmitted modes is evanescent"""
    from pyrate_amd import systems
    eps = systems.uniaxial_eps(1.35, 2.1, (1.0, 0.0, 0.0))
    s = api.OpticalSystem.p(matbackground=api.ConstantIndexGlass.p(api.LocalCoordinates.p(name="bg"), 1.9))
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="entry", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="crystal", decz=5.0), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="exit", decz=5.0), refname=lc1.name)
    elem = api.OpticalElement.p(lc0, name="e")
    elem.addMaterial("c", api.AnisotropicMaterial.p(lc1, eps))
    elem.addSurface("entry", api.Surface.p(lc0), (None, None))
    elem.addSurface("crystal", api.Surface.p(lc1), (None, "c"))
    elem.addSurface("exit", api.Surface.p(lc2), ("c", None))
    s.addElement("e", elem)
    return (s, [("e", [("entry", {}), ("crystal", {}), ("exit", {})])])


def evanescent_bundle_arrays(n=32):
    ang = np.linspace(0.2, 1.1, n)
    x0 = np.vstack((np.zeros(n), np.zeros(n), np.full(n, -1.0)))
    k0 = 1.9 * np.vstack((np.sin(ang) * 0.6, np.sin(ang) * 0.8, np.cos(ang)))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    return (x0, k0, e0)


def prism(api):
    """dispersing prism: two plane faces tilted by +-30 degrees about a common centre, Conrady
    ModelGlass (demos/demo_prism.py geometry)"""
    deg = math.pi / 180.
    s = api.OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="stop", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lcc = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="prismcenter", decz=50.0), refname=lc0.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="surf1", decz=-10.0, tiltx=30. * deg),
                                     refname=lcc.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="surf2", decz=10.0, tiltx=-30. * deg),
                                     refname=lcc.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="image", decz=50.0), refname=lcc.name)
    elem = api.OpticalElement.p(lc0, name="prism")
    elem.addMaterial("glass", api.ModelGlass.p(lc1))
    elem.addSurface("stop", api.Surface.p(lc0), (None, None))
    elem.addSurface("surf1", api.Surface.p(lc1, shape=api.Conic.p(lc1, curv=0),
                                           aperture=api.CircularAperture.p(lc1, maxradius=20.0)), (None, "glass"))
    elem.addSurface("surf2", api.Surface.p(lc2, shape=api.Conic.p(lc2, curv=0),
                                           aperture=api.CircularAperture.p(lc2, maxradius=20.0)), ("glass", None))
    elem.addSurface("image", api.Surface.p(lc3), (None, None))
    s.addElement("prism", elem)
    seq = [("prism", [("stop", {"is_stop": True}), ("surf1", {}), ("surf2", {}), ("image", {})])]
    return (s, seq)


PRISM_RAYS = {"radius": 5.0, "startz": -5., "starty": -20., "anglex": 23 * math.pi / 180.}


def gridsag_data():
    """a smooth freeform sampled on a 25 x 21 grid (not an exact polynomial of the spline's degree)"""
    x = np.linspace(-10.0, 10.0, 25)
    y = np.linspace(-9.0, 9.0, 21)
    (X, Y) = np.meshgrid(x, y, indexing="ij")
    Z = -0.004 * (X ** 2 + 1.3 * Y ** 2) + 0.08 * np.sin(0.25 * X) * np.cos(0.2 * Y) + 1e-5 * X ** 3 * Y
    return (x, y, Z)


def gridsag_system(api):
    """front plane, freeform back surface given as a sag grid (GridSag), tilted image plane"""
    s = api.OpticalSystem.p(name="gridsag")
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lc1 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="front", decz=5.0), refname=lc0.name)
    lc2 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="back", decz=8.0, tiltx=0.03), refname=lc1.name)
    lc3 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="img", decz=60.0), refname=lc2.name)
    elem = api.OpticalElement.p(lc0, name="gs")
    elem.addMaterial("glass", api.ConstantIndexGlass.p(lc1, 1.5168))
    elem.addSurface("front", api.Surface.p(lc1, shape=api.Conic.p(lc1, curv=0.0)), (None, "glass"))
    elem.addSurface("back", api.Surface.p(lc2, shape=api.GridSag.p(lc2, gridsag_data())), ("glass", None))
    elem.addSurface("img", api.Surface.p(lc3), (None, None))
    s.addElement("gs", elem)
    return (s, [("gs", [("front", {}), ("back", {}), ("img", {})])])


def biconic_builduplist():
    """a toric / biconic front surface with polynomial terms (the shape of demos/demo_hud.py)"""
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Biconic", "curvx": 1. / 45., "curvy": 1. / 70., "ccx": -0.4, "ccy": 0.2,
          "coefficients": [(1e-4, 0.3), (-2e-7, -0.5)]}, {"decz": 5.0}, 1.5168, "front", {}),
        ({"shape": "Conic", "curv": -1. / 90.}, {"decz": 6.0}, None, "back", {}),
        ({"shape": "Conic"}, {"decz": 70.0}, None, "image", {}),
    ]


GRAZING_DOME = dict(radius=20.0, decz=30.0, tilt_deg=30.0, rpup=5.0, margin=0.03, z0=-5.0)


def grazing_dome_builduplist():
    """an even asphere that is nearly a hemisphere (R = 20 mm + a small r^4 term), convex towards a bundle tilted by 30
    degrees whose topmost rays pass 0.03 mm below the line that touches the dome: angles of incidence up to 84.4 degrees --
    the Newton iteration's g'(t) = d . grad is 0.1 there, where ADVICE r5 (medium) suspected the 1e-8 stop rule"""
    g = GRAZING_DOME
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Asphere", "curv": 1. / g["radius"], "cc": 0.0, "coefficients": [0.0, 1e-7, 0.0]},
         {"decz": g["decz"]}, 1.5168, "dome", {}),
        ({"shape": "Conic"}, {"decz": 40.0}, None, "image", {}),
    ]


def grazing_dome_bundle_centre():
    """start height (at z0) of the centre of the bundle whose top ray misses the tangent line by ``margin``"""
    import math
    g = GRAZING_DOME
    th = math.radians(g["tilt_deg"])
    (yt, zt) = (g["radius"] * math.cos(th), g["decz"] + g["radius"] - g["radius"] * math.sin(th))
    return yt - (zt - g["z0"]) * math.tan(th) - g["margin"] - g["rpup"]


def mirrors_builduplist():
    """paraboloid mirror + tilted flat fold mirror"""
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Conic", "curv": -1. / 200., "cc": -1.0}, {"decz": 50.0}, None, "primary", {"is_mirror": True}),
        ({"shape": "Conic"}, {"decz": -60.0, "tiltx": 0.2}, None, "fold", {"is_mirror": True}),
        ({"shape": "Conic"}, {"decz": 30.0}, None, "image", {}),
    ]


def hud_like_builduplist():
    """off-axis biconic mirrors in tilted / decentred frames (the kind of system of the
    reference's demos/demo_hud.py:87-104), all rays valid"""
    return [
        ({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
        ({"shape": "Biconic", "curvx": -1. / 300., "curvy": -1. / 260., "ccx": -0.8, "ccy": 0.3,
          "coefficients": [(2e-6, 0.2)]}, {"decz": 40.0, "tiltx": 0.25, "decy": 1.0}, None, "m1",
         {"is_mirror": True}),
        ({"shape": "Biconic", "curvx": 1. / 500., "curvy": 1. / 420., "ccx": 0.0, "ccy": -0.5,
          "coefficients": []}, {"decz": -35.0, "tiltx": 0.25, "tiltThenDecenter": 1}, None, "m2",
         {"is_mirror": True}),
        ({"shape": "Conic"}, {"decz": 45.0, "tiltx": -0.05}, None, "image", {}),
    ]


def two_element_system(api):
    """two OpticalElements in one system: exercises the element loop of
    OpticalSystem.seqtrace (bundle duplicated at the element boundary, material reset to
    the background at every element)"""
    s = api.OpticalSystem.p()
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=0.0),
                                     refname=s.rootcoordinatesystem.name)
    lcs = []
    ref = lc0.name
    for (i, dz) in enumerate((5.0, 4.0, 10.0, 3.0, 60.0)):
        lc = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="l%d" % i, decz=dz), refname=ref)
        lcs.append(lc)
        ref = lc.name
    e1 = api.OpticalElement.p(lc0, name="e1")
    e1.addMaterial("g", api.ConstantIndexGlass.p(lcs[0], n=1.6))
    e1.addSurface("a", api.Surface.p(lcs[0], shape=api.Conic.p(lcs[0], curv=1. / 30.)), (None, "g"))
    e1.addSurface("b", api.Surface.p(lcs[1], shape=api.Conic.p(lcs[1], curv=-1. / 45.),
                                     aperture=api.CircularAperture.p(lcs[1], maxradius=6.0)), ("g", None))
    e2 = api.OpticalElement.p(lc0, name="e2")
    e2.addMaterial("h", api.ConstantIndexGlass.p(lcs[2], n=1.5))
    e2.addSurface("c", api.Surface.p(lcs[2], shape=api.Conic.p(lcs[2], curv=1. / 50., cc=-0.8)), (None, "h"))
    e2.addSurface("d", api.Surface.p(lcs[3]), ("h", None))
    e2.addSurface("img", api.Surface.p(lcs[4]), (None, None))
    s.addElement("e1", e1)
    s.addElement("e2", e2)
    seq = [("e1", [("a", {}), ("b", {})]), ("e2", [("c", {}), ("d", {}), ("img", {})])]
    return (s, seq)


def write_mini_glass_database(basepath, pages):
    """A miniature refractiveindex.info checkout (library.yml + data/<shelf>/<book>/<page>.yml) built
    from page dictionaries ``{key: page}``: shelves / books / pages / DIVIDER rows like the real
    library file, long names "<KEY> (TEST)".  Returns {key: long name}."""
    import os
    import yaml
    keys = sorted(pages.keys())
    names = {}
    shelves = [{"SHELF": "glass", "name": "GLASS - glasses", "content": [{"DIVIDER": "Test glasses"}]},
               {"SHELF": "other", "name": "OTHER - miscellaneous", "content": []}]
    for (i, key) in enumerate(keys):
        shelf = shelves[0] if i % 3 != 2 else shelves[1]
        book = key.split("_")[0]
        rel = "%s/%s/%s.yml" % (shelf["SHELF"], book, key)
        names[key] = "%s (TEST)" % key.upper()
        entry = {"PAGE": key, "name": names[key], "data": rel}
        for b in shelf["content"]:
            if b.get("BOOK") == book:
                b["content"].append(entry)
                break
        else:
            shelf["content"].append({"BOOK": book, "name": book.upper(),
                                     "content": [{"DIVIDER": "pages"}, entry]})
        full = os.path.join(basepath, "data", rel)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "w") as f:
            yaml.safe_dump(pages[key], f)
    with open(os.path.join(basepath, "library.yml"), "w") as f:
        yaml.safe_dump(shelves, f)
    return names


def catalog_doublet_tuples(names):
    """cemented doublet whose two glasses are given by catalogue NAME (resolved through
    material_db_path, pyrateoptics/__init__.py:183-196)"""
    return [(0.0, 0.0, 0.0, None, "stop", {"is_stop": True}),
            (62.8, 0.0, 3.0, names["formula1_nbk7"], "front", {}),
            (-45.7, 0.0, 4.0, names["formula2"], "cement", {}),
            (-128.2, 0.0, 2.5, None, "rear", {}),
            (0.0, 0.0, 97.2, None, "image", {})]


def paraxial_summary(tuples, obj_dist, stop_radius, obj_angle_deg=5.0):
    """First-order data of a rotationally symmetric prescription given as builder tuples
    (r, cc, thickness_before, n_after | None, name, opts): y-nu matrices -> the summary rows a
    WinLens SPD file carries (efl, principal planes via l / l', pupils, field)."""
    def surf_matrix(r, n1, n2):
        c = 0.0 if (r == 0 or abs(r) > 1e15) else 1. / r
        return np.array([[1., 0.], [-(n2 - n1) * c, 1.]])

    def gap_matrix(t, n):
        return np.array([[1., t / n], [0., 1.]])

    surfs = tuples[:-1]                                   # without the image plane
    mats = []
    n = 1.0
    for (r, _, t, n_after, _, opts) in surfs:
        n2 = 1.0 if n_after is None else float(n_after)
        mats.append((gap_matrix(t, n), surf_matrix(r, n, n2), opts.get("is_stop", False)))
        n = n2
    total = np.eye(2)
    istop = [i for (i, m) in enumerate(mats) if m[2]][0]
    front = np.eye(2)
    for (i, (g, s, _)) in enumerate(mats):
        total = s.dot(g).dot(total) if i > 0 else s.dot(total)
        if i == istop:
            front = total.copy()
    rear = total.dot(np.linalg.inv(front))                # stop -> last surface
    (A, B, C, D) = (total[0, 0], total[0, 1], total[1, 0], total[1, 1])
    efl = -1. / C
    pp_obj = (D - 1.) / C
    pp_img = (1. - A) / C
    thick = sum(t for (_, _, t, _, _, _) in surfs[1:])
    img_dist = 1. / (1. / efl + 1. / (obj_dist - pp_obj)) + pp_img
    mag = (img_dist - pp_img) / (obj_dist - pp_obj)
    # entrance pupil: object point z_e (from the first surface) imaged onto the stop by the front group
    (a, b, c, d) = front.ravel()
    z_e = b / a                                            # y_stop = a*y - ... = 0 for rays from (z_e, axis)
    m_e = 1. / a if abs(a) > 1e-300 else 1.0              # stop height / pupil height
    entpup = z_e
    entpup_rad = abs(stop_radius / (a - c * 0.0)) if a != 0 else stop_radius
    (a2, b2, c2, d2) = rear.ravel()
    expup = -b2 / d2                                       # from the last surface
    expup_rad = abs(stop_radius * (a2 - b2 * c2 / d2))
    obj_height = math.tan(math.radians(obj_angle_deg)) * (entpup - obj_dist)
    img_height = mag * obj_height
    img_angle = math.degrees(math.atan(-img_height / (expup - img_dist)))
    out = dict(efl=efl, mag=mag, obj_dist=obj_dist, img_dist=img_dist, l=obj_dist - pp_obj,
               ldash=img_dist - pp_img, track=thick + img_dist - obj_dist, stop_rad=stop_radius,
               entpup_rad=entpup_rad, expup_rad=expup_rad, obj_angle=obj_angle_deg,
               obj_height=obj_height, img_angle=img_angle, img_height=img_height)
    return {k: float(v) for (k, v) in out.items()}


def write_synthetic_spd(path, groups, stop_after, glass_indices, waves_nm, summary, gaps):
    """Writes a WinLens-SPD-shaped CSV file (component rows, LENS blocks with Surf / Space /
    GlassIndex rows, paraxial summary rows) for lens groups ``[[(radius, free_radius), ...], ...]``
    with glass names / thicknesses ``glass_indices[group] = [(name, thickness, (n_d, n_F, n_C)), ...]``,
    a stop component behind group ``stop_after`` and air gaps ``gaps`` behind every component."""
    def q(s):
        return '"%s"' % s
    rows = [",".join([q("-- Version 5.0 file --"), q("01-01-2024")]), ",".join([q("synthetic"), q(""), "1"]),
            "#FALSE#", "0,#FALSE#,#FALSE#,#FALSE#,0",
            ",".join(["%d,1,0,0" % len(waves_nm)] + ["%r" % w for w in waves_nm])]
    comp_no = 0
    gap_iter = iter(gaps)

    def component_row(label, kind, nsurf, gap):
        cells = [q(str(comp_no)), q(label), "#TRUE#", q("Nom" if kind else ""), q(kind), str(nsurf),
                 q("0, %r" % gap), "#TRUE#", q(""), q(""), q(""), q(""), q(""), q("air"), q(" "), q(""), q(""),
                 q("NonSH"), q(""), "0"]
        return ",".join(cells)

    for (g, surfs) in enumerate(groups):
        comp_no += 1
        rows.append(component_row("   ", "lens", len(surfs), next(gap_iter)))
        rows.append(q(" LENS %d" % comp_no))
        for (j, (radius, free_radius)) in enumerate(surfs):
            rows.append(",".join([q("   Surf  %d" % (j + 1)), q("sphere"), "0", "%r" % radius, "#TRUE#", q(""),
                                  q(""), "0", "%r" % waves_nm[0], "%r" % free_radius, "0", q(""), q(""), q(""),
                                  q("FALSE"), q("FALSE|0|")]))
            if j < len(surfs) - 1:
                (gname, thickness, nn) = glass_indices[g][j]
                rows.append(",".join([q("   Space %d" % (j + 1)), "%r" % thickness, q(""), q(gname),
                                      q("Schott                        ")] + [q("")] * 7))
                rows.append(",".join([q("      GlassIndex")] + ["%r" % v for v in nn] + [q(""), q("")]))
        rows.append(q(" LENS %d End" % comp_no))
        if g == stop_after:
            comp_no += 1
            rows.append(component_row("Stop", "", 0, next(gap_iter)).replace("#TRUE#", "#FALSE#", 1))
    sm = summary
    rows += [",".join([q("Defocus"), "0", "0", q("")]),
             ",".join([q("Waveband")] + ["%r" % w for w in waves_nm]),
             ",".join([q("efl"), "%r" % sm["efl"]]),
             ",".join([q("ObjDist"), "%r" % sm["obj_dist"], q("   ImagDist"), "%r" % sm["img_dist"]]),
             ",".join([q("l"), "%r" % sm["l"], q("   l'"), "%r" % sm["ldash"]]),
             ",".join([q("Mag"), "%r" % sm["mag"], q("   AngMag"), "0"]),
             ",".join([q("Track"), "%r" % sm["track"]]),
             ",".join([q("ObjNa"), "0.0065", q("   ImagNa"), "0.0513", q("Stop Rad"), "%r" % sm["stop_rad"]]),
             ",".join([q("Entr Pup Rad"), "%r" % sm["entpup_rad"], q("Exit Pup Rad"), "%r" % sm["expup_rad"]]),
             ",".join([q("ObjAngle"), "%r" % sm["obj_angle"], q("   ObjHeight"), "%r" % sm["obj_height"]]),
             ",".join([q("ImagAngle"), "%r" % sm["img_angle"], q("ImagHeight"), "%r" % sm["img_height"]]),
             ",".join(["400", "800", "632.8", ".001", ".8057059", "805.7059", "587.56", "0", "0", "0"]),
             q("synthetic.SPD")]
    with open(path, "w", newline="") as f:
        f.write("\r\n".join(rows) + "\r\n")


def synthetic_double_gauss_spd(path):
    """the double Gauss of pyrate_amd.systems written as an SPD file; returns the builder tuples
    (d-line indices) it corresponds to"""
    from pyrate_amd import systems
    tuples = systems.double_gauss_tuples()
    by_nd = {v[0]: (k, v) for (k, v) in systems.DOUBLE_GAUSS_GLASSES.items()}
    summary = paraxial_summary(tuples, -1000.0, 5.0)
    t = [tp[2] for tp in tuples]                           # thickness before surface i
    r = [tp[0] for tp in tuples]
    n = [tp[3] for tp in tuples]

    def glass(i):
        (name, nn) = by_nd[n[i]]
        return (name, t[i + 1], nn)
    groups = [[(r[0], 12.0), (r[1], 12.0), (r[2], 12.0)], [(r[3], 12.0), (r[4], 12.0)],
              [(r[6], 12.0), (r[7], 12.0)], [(r[8], 12.0), (r[9], 12.0), (r[10], 12.0)]]
    glass_indices = [[glass(0), glass(1)], [glass(3)], [glass(6)], [glass(8), glass(9)]]
    gaps = [t[3], t[5], t[6], t[8], 0.0]                   # behind lens 1, lens 2, stop, lens 3, lens 4
    write_synthetic_spd(path, groups, 1, glass_indices, [round(w * 1e6, 4) for w in systems.DOUBLE_GAUSS_WAVES_MM], summary, gaps)
    return tuples


def write_spd_glass_database(basepath):
    """refractiveindex.info-shaped database holding the three glasses of the double Gauss as
    "formula 5" pages (n = c0 + c1 w**-1 + c2 w**-3.5, w in um: the Conrady model through the
    d, F, C indices); F5 additionally sits where the reference's SPD importer looks for it by
    shelf / book / page (io/spd.py:890-893)."""
    import os
    import yaml
    from pyrate_amd import systems
    books = {}
    for (name, nn) in systems.DOUBLE_GAUSS_GLASSES.items():
        (n0, a, b) = systems.conrady_fit(*nn)
        page = {"DATA": [{"type": "formula 5", "wavelength_range": "0.35 1.1",
                          "coefficients": "%r %r -1 %r -3.5" % (float(n0), float(a) * 1e3, float(b) * 1e3 ** 3.5)}],
                "SPECS": {"nd": float(nn[0])}}
        book = "SCHOTT-F" if name == "F5" else "SCHOTT-" + name.split("-")[0][:1]
        rel = "glass/schott/%s.yml" % name
        books.setdefault(book, []).append({"PAGE": name, "name": name, "data": rel})
        full = os.path.join(basepath, "data", rel)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "w") as f:
            yaml.safe_dump(page, f)
    lib = [{"SHELF": "glass", "name": "GLASS", "content":
            [{"BOOK": b, "name": b, "content": pages} for (b, pages) in sorted(books.items())]}]
    with open(os.path.join(basepath, "library.yml"), "w") as f:
        yaml.safe_dump(lib, f)


def random_object_graph(api, seed):
    """random object graph (nested frames with both tilt orders, every shape class, aperture and
    material frames, mirrors, crystals) built through either API namespace"""
    rng = np.random.RandomState(31000 + seed)
    s = api.OpticalSystem.p()
    lc_prev = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=float(rng.uniform(0, 2))),
                                         refname=s.rootcoordinatesystem.name)
    elem = api.OpticalElement.p(lc_prev, name="e")
    seq = []
    last = None
    for j in range(int(rng.randint(2, 7))):
        kw = {k: float(rng.uniform(-0.4, 0.4)) for k in ("decx", "decy", "tiltx", "tilty", "tiltz") if rng.rand() < 0.5}
        kw["decz"] = float(rng.uniform(1, 9))
        kw["tiltThenDecenter"] = int(rng.randint(0, 2))
        lc = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="s%d" % j, **kw), refname=lc_prev.name)
        lcs = lc
        if rng.rand() < 0.3:            # shape in its own nested frame
            lcs = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="sh%d" % j, decx=float(rng.uniform(-0.2, 0.2)),
                                                                    tiltz=float(rng.uniform(-0.5, 0.5))), refname=lc.name)
        kind = int(rng.randint(0, 6))
        c = float(rng.uniform(-1, 1) / rng.uniform(15, 90))
        if kind == 0:
            shape = api.Conic.p(lcs, curv=c, cc=float(rng.uniform(-1.5, 1)))
        elif kind == 1:
            shape = api.Asphere.p(lcs, curv=c, cc=float(rng.uniform(-1.5, 1)),
                                  coefficients=[float(v) for v in rng.uniform(-1e-5, 1e-5, int(rng.randint(0, 4)))])
        elif kind == 2:
            shape = api.Biconic.p(lcs, curvx=c, ccx=float(rng.uniform(-1, 0.5)), curvy=c * 0.7, ccy=float(rng.uniform(-1, 0.5)),
                                  coefficients=[(float(rng.uniform(-1e-5, 1e-5)), float(rng.uniform(-0.5, 0.5)))])
        elif kind == 3:
            shape = api.XYPolynomials.p(lcs, normradius=float(rng.uniform(5, 12)),
                                        coefficients=[(int(i), int(k), float(rng.uniform(-0.05, 0.05)))
                                                      for (i, k) in ((2, 0), (0, 2), (1, 2), (3, 1))])
        elif kind == 4:
            shape = api.ZernikeFringe.p(lcs, normradius=float(rng.uniform(6, 12)),
                                        coefficients=[float(v) for v in rng.uniform(-0.02, 0.02, int(rng.randint(1, 12)))])
        else:
            lcz = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="z%d" % j, decx=float(rng.uniform(-1, 1)),
                                                                    decy=float(rng.uniform(-1, 1))), refname=lcs.name)
            shape = api.LinearCombination.p(lcs, list_of_coefficients_and_shapes=[
                (float(rng.uniform(0.5, 1.5)), api.Asphere.p(lcs, curv=c, cc=-0.5, coefficients=[0.0, 1e-6])),
                (float(rng.uniform(0.5, 1.5)), api.ZernikeFringe.p(lcz, normradius=10.0,
                                                                   coefficients=[float(v) for v in rng.uniform(-0.02, 0.02, 6)]))])
        aper = None
        r = rng.rand()
        if r < 0.3:
            aper = api.CircularAperture.p(lc, maxradius=float(rng.uniform(3, 8)), minradius=float(rng.choice([0.0, 0.4])))
        elif r < 0.5:
            lca = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="a%d" % j, decy=0.3, tiltz=0.2), refname=lc.name)
            aper = api.RectangularAperture.p(lca, width=float(rng.uniform(6, 12)), height=float(rng.uniform(6, 12)))
        surf = api.Surface.p(lc, shape=shape, aperture=aper)
        mirror = j > 0 and rng.rand() < 0.2
        mat = last
        if not mirror:
            r = rng.rand()
            if r < 0.25:
                mat = None
            else:
                mat = "m%d" % j
                lcm = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="mf%d" % j, tiltx=float(rng.uniform(-0.6, 0.6)),
                                                                        tilty=float(rng.uniform(-0.6, 0.6))), refname=lc.name)
                if r < 0.55:
                    elem.addMaterial(mat, api.ConstantIndexGlass.p(lcm, float(rng.uniform(1.3, 1.9))))
                elif r < 0.7:
                    elem.addMaterial(mat, api.ModelGlass.p(lcm))
                else:
                    e = rng.uniform(-0.2, 0.2, (3, 3))
                    elem.addMaterial(mat, api.AnisotropicMaterial.p(lcm, np.eye(3) * 2.4 + e + e.T))
        elem.addSurface("s%d" % j, surf, (last, mat))
        seq.append(("s%d" % j, {"is_mirror": True} if mirror else {}))
        last = mat
        lc_prev = lc
    s.addElement("e", elem)
    return (s, [("e", seq)])
