"""
GPU tests of the drop-in layer: systems built with the host mirror classes
(pyrate_amd.raytracer.*, same construction code as the reference run, tests/systems_zoo.py)
and traced through ``OpticalSystem.seqtrace`` must return the reference's RayPath
structure -- number of bundles (incl. the duplicated one per element), (P,3,N) shapes,
compaction, rayID, valid -- with x and k inside the parity tolerance.
"""
import math

import numpy as np
import pytest

import _golden
import systems_zoo as zoo
from pyrate_amd import systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(gpu_device):
    return zoo.mirror_api()


def bundle_of(api, case):
    return api.RayBundle(x0=case.x0, k0=case.k0, Efield0=case.E0, wave=case.wave)


def assert_paths_match(path, ref_bundles, rtol_x=1e-10, atol_k=1e-10, loose_x=None):
    assert len(path.raybundles) == len(ref_bundles)
    for (i, (rb, ref)) in enumerate(zip(path.raybundles, ref_bundles)):
        assert rb.x.shape == ref["x"].shape, (i, rb.x.shape, ref["x"].shape)
        assert rb.k.shape == ref["k"].shape
        assert rb.valid.shape == ref["valid"].shape and rb.valid.dtype == bool
        assert np.array_equal(rb.rayID, ref["id"]), i
        assert np.array_equal(rb.valid, ref["valid"].astype(bool)), i
        assert rb.Efield.shape == ref["x"].shape
        for p in range(ref["x"].shape[0]):
            v = ref["valid"][p].astype(bool)
            if not v.any():
                continue
            xr = ref["x"][p][:, v]
            err = np.abs(rb.x[p][:, v] - xr)
            tol = rtol_x * _golden.relative_scale(xr)
            if loose_x is not None:
                tol = tol + loose_x
            assert np.all(err <= tol), (i, p, float(err.max()))
            kr = np.real(ref["k"][p][:, v])
            fin = np.all(np.isfinite(kr), axis=0)
            ek = np.abs(np.real(rb.k[p][:, v])[:, fin] - kr[:, fin])
            assert np.all(ek <= atol_k + (loose_x if loose_x is not None else 0.0)), (i, p, float(ek.max()))


def test_seqtrace_doublet_structure_and_values(api):
    case = _golden.load_case("doublet_clipped")
    (s, seq) = zoo.doublet(api)
    ib = bundle_of(api, case)
    rpaths = s.seqtrace(ib, seq)
    assert len(rpaths) == 1
    assert_paths_match(rpaths[0], case.raw_bundles)
    # complex k in (collimated_bundle) -> complex k out, like the reference
    assert np.iscomplexobj(case.k0) and np.iscomplexobj(rpaths[0].raybundles[-1].k)
    # the initial bundle is not modified (optical_system.py:74)
    assert ib.x.shape[0] == 1
    # README-style access pattern
    x_img = rpaths[0].raybundles[-1].x[-1, 0, :]
    assert x_img.shape[0] == case.raw_bundles[-1]["x"].shape[2]


@pytest.mark.parametrize("name,builder", [
    ("double_gauss_wide", lambda api: api.build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())),
    ("double_gauss_defaultE", lambda api: api.build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())),
    ("tilted_frames", lambda api: zoo.tilted(api)),
    ("mirrors", lambda api: api.build_simple_optical_system(zoo.mirrors_builduplist())),
    ("two_elements", lambda api: zoo.two_element_system(api)),
])
def test_seqtrace_isotropic_cases(api, name, builder):
    case = _golden.load_case(name)
    (s, seq) = builder(api)
    e0 = None if name == "double_gauss_defaultE" else case.E0
    ib = api.RayBundle(x0=case.x0, k0=case.k0, Efield0=e0, wave=case.wave)
    rpaths = s.seqtrace(ib, seq)
    assert_paths_match(rpaths[0], case.raw_bundles)


def test_seqtrace_explicit_shape(api):
    case = _golden.load_case("xypoly_field5")
    (s, seq) = api.build_simple_optical_system(zoo.xypoly_builduplist())
    rpaths = s.seqtrace(bundle_of(api, case), seq)
    # reference hit points carry the fsolve residual (SURVEY.md headline 4): loose absolute tolerance
    assert_paths_match(rpaths[0], case.raw_bundles, loose_x=1e-9)


def test_seqtrace_of_imported_zmx_prescription(api):
    """lenssystem.ZMX (the reference's own test file) -> this package's ZMXParser -> seqtrace:
    bundle by bundle what the reference's parser + trace returned"""
    import os
    from pyrate_amd.raytracer.io.zmx import ZMXParser
    case = _golden.load_case("zmx_lenssystem")
    zp = ZMXParser(os.path.join(_golden.GOLDEN_DIR, "lenssystem.ZMX"))
    (s, seq) = zp.create_optical_system({"BK7": api.ConstantIndexGlass.p(api.LocalCoordinates.p(name="t"), 1.5168)})
    rpaths = s.seqtrace(bundle_of(api, case), seq)
    assert_paths_match(rpaths[0], case.raw_bundles, loose_x=1e-8)    # 13 fsolve surfaces in a row


def test_seqtrace_of_imported_spd_prescription(api, tmp_path):
    """synthetic WinLens SPD file of the double Gauss -> SPDParser (glasses from the file's own
    GlassIndex rows) -> seqtrace at the F line == what the reference's importer + trace returned"""
    from pyrate_amd.raytracer.io.spd import SPDParser
    case = _golden.load_case("spd_double_gauss_Fline")
    f = str(tmp_path / "dg.spd")
    zoo.synthetic_double_gauss_spd(f)
    (s, seq) = SPDParser(f).create_optical_system()
    rpaths = s.seqtrace(bundle_of(api, case), seq)
    assert_paths_match(rpaths[0], case.raw_bundles)


@pytest.mark.parametrize("name,builder", [
    ("zernike_fringe_field3", lambda api: api.build_simple_optical_system(zoo.zernike_builduplist("Fringe"))),
    ("zernike_ansi_field2", lambda api: api.build_simple_optical_system(zoo.zernike_builduplist("ANSI"))),
    ("zernike_combination_mirror", lambda api: zoo.zernike_combination_system(api)),
])
def test_seqtrace_zernike_surfaces(api, name, builder):
    """bundles up to the hit points ON the Zernike surface == the reference's; behind it the
    reference refracts with a normal that is not its surface's normal (_golden.REFERENCE_NORMAL_DEFECT),
    so the rest of the path is checked against the oracle"""
    from oracle import seqtrace_np as oracle
    case = _golden.load_case(name)
    (s, seq) = builder(api)
    rp = s.seqtrace(bundle_of(api, case), seq)[0]
    nb = _golden.REFERENCE_NORMAL_DEFECT[name] + 2           # bundles whose points are all trusted
    head = type(rp)(rp.raybundles[0])
    head.raybundles = rp.raybundles[:nb]
    assert_paths_match(head, case.raw_bundles[:nb], loose_x=1e-7)      # reference: fsolve, xtol 1e-6
    assert len(rp.raybundles) == len(case.raw_bundles)
    with np.errstate(all="ignore"):
        out = oracle.trace(case.table, case.x0, case.k0, case.E0)
    img = rp.raybundles[-1]
    ok = out[-1]["valid"] if len(case.table) == 1 else out[-2]["valid_out"]
    assert img.num_rays == int(np.sum(ok))
    assert np.allclose(img.x[0], out[-1]["x_hit"][:, ok], rtol=0, atol=1e-10)
    assert np.allclose(np.real(img.k[0]), np.real(out[-1]["k_out"])[:, ok], rtol=0, atol=1e-12)


def test_seqtrace_symmetric_zernike_surface_whole_path(api):
    """a Zernike surface with m = 0 terms only: the reference's gradient is right there, so EVERY bundle -- also the
    ones behind the refraction at the Zernike surface -- is compared with the reference's"""
    case = _golden.load_case("zernike_fringe_symmetric_field3")
    (s, seq) = api.build_simple_optical_system(zoo.zernike_builduplist("Fringe", symmetric=True))
    assert_paths_match(s.seqtrace(bundle_of(api, case), seq)[0], case.raw_bundles, loose_x=1e-7)


def test_seqtrace_gridsag_surface(api):
    case = _golden.load_case("gridsag_field2")
    (s, seq) = zoo.gridsag_system(api)
    rpaths = s.seqtrace(bundle_of(api, case), seq)
    assert_paths_match(rpaths[0], case.raw_bundles, loose_x=1e-8)      # reference: fsolve, xtol 1e-4


@pytest.mark.parametrize("name", ["hud_patent_axis", "hud_patent_field-15"])
def test_seqtrace_hud_patent_prism(api, name):
    """demos/demo_hud.py's prism (14 surfaces, biconic faces, one hit twice) built from the mirror classes: every
    bundle == the reference's, no ray lost -- at -15 degrees the hit points on the first face lie where the
    biconic's evaluation is noisy and the Newton steps stall at 1e-14 instead of 1e-15 (explicit_t's noise floor)"""
    from demos import demo_hud
    case = _golden.load_case(name)
    (s, seq) = demo_hud.build(api)
    rp = s.seqtrace(bundle_of(api, case), seq)[0]
    assert rp.raybundles[-1].num_rays == case.raw_bundles[-1]["x"].shape[-1] == 140
    assert_paths_match(rp, case.raw_bundles, loose_x=1e-7)
    # the plugin-granular loop (Surface.intersect / Material.refract / reflect per surface) through the same prism
    assert_paths_match(s._seqtrace_generic(bundle_of(api, case), seq, False)[0], case.raw_bundles, loose_x=1e-7)


def test_seqtrace_three_mirror_anastigmat_with_off_axis_paraboloid(api):
    """demos/demo_mirrors.py built with this package's builder: every bundle == the reference's (8 surfaces, four
    reflections in air, the paraboloid hit 35 mm from its vertex)"""
    from demos import demo_mirrors
    case = _golden.load_case("tma_paraboloid_field0p5")
    (s, seq) = demo_mirrors.build(api.build_simple_optical_system)
    assert_paths_match(s.seqtrace(bundle_of(api, case), seq)[0], case.raw_bundles)


def test_seqtrace_rotated_combination_surface(api):
    """a LinearCombination lens surface whose polynomial part is decentred and rotated about the axis, built from
    the mirror classes: every bundle == the reference's (hit points on the freeform surface by its fsolve, xtol
    1e-6), from the fused trace and from the plugin-granular loop"""
    case = _golden.load_case("rotated_combination_lens")
    (s, seq) = zoo.rotated_combination_system(api)
    assert_paths_match(s.seqtrace(bundle_of(api, case), seq)[0], case.raw_bundles, loose_x=1e-7)
    assert_paths_match(s._seqtrace_generic(bundle_of(api, case), seq, False)[0], case.raw_bundles, loose_x=1e-7)


def test_plugin_granular_path_matches_fused(api):
    """OpticalElement.seqtrace (Material.propagate / refract per surface + device compaction)
    gives the same RayPath as the fused launch"""
    case = _golden.load_case("double_gauss_wide")
    (s, seq) = api.build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    fused = s.seqtrace(bundle_of(api, case), seq)[0]
    generic = s._seqtrace_generic(bundle_of(api, case), seq, False)[0]
    assert_paths_match(generic, case.raw_bundles)
    for (a, b) in zip(fused.raybundles, generic.raybundles):
        assert np.array_equal(a.rayID, b.rayID)
        assert np.allclose(a.x, b.x, rtol=0, atol=1e-12, equal_nan=True)


def aniso_system(api, which, stop_radius=None):
    if which == "isoeps":
        return zoo.aniso_doublet(api, 1.5168 ** 2 * np.eye(3), 1.6727 ** 2 * np.eye(3))
    c = systems.CALCITE_TILTED
    eps1 = systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"])
    eps2 = systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))
    return zoo.aniso_doublet(api, eps1, eps2, stop_radius)


@pytest.mark.parametrize("name,which,stop", [("aniso_doublet_isoeps", "isoeps", None),
                                             ("aniso_doublet_uniaxial", "uni", None),
                                             ("aniso_doublet_uniaxial_clipped", "uni", None),
                                             ("aniso_doublet_uniaxial_stopped", "uni", 8.0)])
def test_seqtrace_anisotropic(api, name, which, stop):
    case = _golden.load_case(name)
    (s, seq) = aniso_system(api, which, stop)
    rpaths = s.seqtrace(bundle_of(api, case), seq)
    assert len(rpaths) == 1
    assert rpaths[0].containsSplitted()
    assert_paths_match(rpaths[0], case.raw_bundles)
    # E of the crystal bundles solves the wave equation: (eps - k^2 + k k^T) E = 0
    rb = rpaths[0].raybundles[3]
    eps = np.asarray(case.table[1]["material"]["eps_re"])
    (k, E) = (np.real(rb.k[0]), np.real(rb.Efield[0]))
    res = eps.dot(E) - np.sum(k * k, axis=0) * E + k * np.sum(k * E, axis=0)
    assert np.nanmax(np.abs(res)) < 1e-12


@pytest.mark.parametrize("which,stop", [("isoeps", None), ("uni", None), ("uni", 8.0)])
def test_seqtrace_anisotropic_dense_path_equals_plugin_path(api, which, stop):
    """one engine trace + lazily carved bundles == the per-surface propagate / refract loop"""
    case = _golden.load_case("aniso_doublet_uniaxial_stopped" if stop else "aniso_doublet_uniaxial_clipped")
    (s, seq) = aniso_system(api, which, stop)
    dense = s.seqtrace(bundle_of(api, case), seq)[0]
    # (the dense trace of lossless crystals carries no E fields: they are made when a bundle's Efield / direction is
    # looked at -- the comparisons of Efield and returnKtoD below go through that path)
    assert dense.dense is not None and dense.dense.e_out is None and dense.dense.k_out_im is not None
    plugin = s._seqtrace_generic(bundle_of(api, case), seq, False)[0]
    assert len(dense.raybundles) == len(plugin.raybundles)
    for (a, b) in zip(dense.raybundles, plugin.raybundles):
        assert a.splitted == b.splitted
        assert np.array_equal(a.rayID, b.rayID)
        assert np.array_equal(a.valid, b.valid)
        assert a.k.dtype == b.k.dtype
        assert np.allclose(a.x, b.x, rtol=0, atol=1e-12, equal_nan=True)
        assert np.allclose(a.k, b.k, rtol=0, atol=1e-13, equal_nan=True)
        if a.splitted:
            assert np.allclose(a.Efield, b.Efield, rtol=0, atol=1e-12, equal_nan=True)
            assert np.allclose(a.returnKtoD(), b.returnKtoD(), rtol=0, atol=1e-12, equal_nan=True)


@pytest.mark.parametrize("name,tilt", [("aniso_mirror_uniaxial", 10.0), ("aniso_mirror_biaxial", 7.0)])
def test_seqtrace_crystal_mirror(api, name, tilt):
    """reflection inside a crystal (backward solution pair), dense path and plugin path"""
    case = _golden.load_case(name)
    eps = np.asarray(case.table[1]["material"]["eps_re"])
    (s, seq) = zoo.crystal_mirror(api, eps, tilt_deg=tilt)
    rp = s.seqtrace(bundle_of(api, case), seq)
    assert len(rp) == 1 and rp[0].raybundles[-1].num_rays == 4 * case.x0.shape[1]
    assert_paths_match(rp[0], case.raw_bundles)
    assert_paths_match(s._seqtrace_generic(bundle_of(api, case), seq, False)[0], case.raw_bundles)


def test_seqtrace_crystal_mirror_splitup(api):
    case = _golden.load_case("aniso_mirror_uniaxial_split")
    eps = np.asarray(case.table[1]["material"]["eps_re"])
    (s, seq) = zoo.crystal_mirror(api, eps)
    rpaths = s.seqtrace(bundle_of(api, case), seq, splitup=True)
    assert len(rpaths) == case.npaths == 4
    for (rp, ref) in zip(rpaths, _raw_paths(case)):
        assert_paths_match(rp, ref)


def test_seqtrace_anisotropic_splitup_forks_four_paths(api):
    case = _golden.load_case("aniso_doublet_uniaxial_split")
    (s, seq) = aniso_system(api, "uni")
    rpaths = s.seqtrace(bundle_of(api, case), seq, splitup=True)
    assert len(rpaths) == case.npaths == 4
    for (rp, ref) in zip(rpaths, [z for z in _raw_paths(case)]):
        assert_paths_match(rp, ref)


def _raw_paths(case):
    import os
    z = np.load(os.path.join(_golden.GOLDEN_DIR, case.name + ".npz"))
    out = []
    for pi in range(case.npaths):
        pre = "" if pi == 0 else "p%d_" % pi
        nb = int(z[pre + "nb"])
        out.append([dict(x=z[pre + "b%d_x" % i], k=z[pre + "b%d_k" % i],
                         valid=z[pre + "b%d_valid" % i], id=z[pre + "b%d_id" % i]) for i in range(nb)])
    return out


def test_surface_and_shape_plugin_calls(api):
    """Shape.intersect, Surface.intersect(remove_rays_outside_aperture), getSag/getGrad/getNormal"""
    lc = api.LocalCoordinates.p(name="lc", decz=3.0)
    shape = api.Conic.p(lc, curv=1. / 20., cc=-0.3)
    surf = api.Surface.p(lc, shape=shape, aperture=api.CircularAperture.p(lc, maxradius=2.0))
    (o, k, _) = zoo.disk_bundle_arrays(200, 4.0, 0.0)
    b1 = api.RayBundle(o, k, None)
    surf.intersect(b1)
    b2 = api.RayBundle(o, k, None)
    surf.intersect(b2, remove_rays_outside_aperture=False)
    b3 = api.RayBundle(o, k, None)
    shape.intersect(b3)
    assert b1.x.shape == (2, 3, o.shape[1])
    r2 = b1.x[1, 0] ** 2 + b1.x[1, 1] ** 2
    assert np.array_equal(b1.valid[1], r2 <= 4.0)
    assert b2.valid[1].all() and np.array_equal(b2.x, b3.x)
    z = shape.getSag(b1.x[1, 0], b1.x[1, 1])
    assert np.allclose(z, b1.x[1, 2] - 3.0, rtol=0, atol=1e-13)
    n = shape.getNormal(b1.x[1, 0], b1.x[1, 1])
    assert np.allclose(np.sum(n * n, axis=0), 1.0)
    g = shape.getGrad(np.array([0.5]), np.array([-0.25]))
    assert g.shape == (3, 1)


def test_dropin_seqtrace_on_duck_typed_objects(api):
    """pyrate_amd.dropin.seqtrace only READS the object graph through the attributes the
    reference's loop uses, so stripped-down stand-ins (no methods at all) are enough"""
    import types
    from pyrate_amd import dropin
    case = _golden.load_case("two_elements")
    (s, seq) = zoo.two_element_system(api)

    def strip(obj, names):
        return types.SimpleNamespace(**{n: getattr(obj, n) for n in names})

    def strip_lc(lc):
        return types.SimpleNamespace(localbasis=lc.localbasis, globalcoordinates=lc.globalcoordinates)
    elements = {}
    for (ek, el) in s.elements.items():
        surfaces = {}
        for (sk, su) in el.surfaces.items():
            shape = types.SimpleNamespace(kind=su.shape.kind, lc=strip_lc(su.shape.lc),
                                          curvature=su.shape.curvature, conic=su.shape.conic)
            ap = types.SimpleNamespace(kind=su.aperture.kind, annotations=su.aperture.annotations,
                                       lc=strip_lc(su.aperture.lc))
            surfaces[sk] = types.SimpleNamespace(shape=shape, aperture=ap)
        mats = {mk: types.SimpleNamespace(lc=strip_lc(m.lc), get_optical_index=m.get_optical_index)
                for (mk, m) in el.materials.items()}
        elements[ek] = types.SimpleNamespace(surfaces=surfaces, materials=mats, annotations=el.annotations)
    bg = types.SimpleNamespace(lc=strip_lc(s.material_background.lc),
                               get_optical_index=s.material_background.get_optical_index)
    plain_system = types.SimpleNamespace(elements=elements, material_background=bg)
    ref_bundle = types.SimpleNamespace(x=case.x0[None], k=case.k0[None], Efield=case.E0[None],
                                       rayID=np.arange(case.x0.shape[1]), wave=case.wave)
    rpaths = dropin.seqtrace(plain_system, ref_bundle, seq)
    assert_paths_match(rpaths[0], case.raw_bundles)


def test_raybundle_api_contract(api):
    """RayBundle data contract of the reference (ray.py:34-115): shapes, defaults, rayID,
    cumulative valid on append, deepcopy independence, device-tensor inputs"""
    import copy
    import torch
    n = 7
    x0 = np.arange(3 * n, dtype=float).reshape(3, n)
    k0 = np.tile(np.array([[0.], [0.], [1.]]), (1, n))
    rb = api.RayBundle(x0, k0, None)
    assert rb.x.shape == (1, 3, n) and rb.k.shape == (1, 3, n) and rb.valid.shape == (1, n)
    assert rb.valid.dtype == bool and rb.valid.all()
    assert np.array_equal(rb.rayID, np.arange(n))
    assert np.array_equal(rb.Efield[0], np.vstack((np.zeros(n), np.ones(n), np.zeros(n))))     # ray.py:71-73
    assert rb.wave == 0.5876e-3 and rb.splitted is False
    v1 = np.array([1, 1, 0, 1, 1, 1, 1], dtype=bool)
    rb.append(x0 + 1, k0, None, v1)
    v2 = np.array([1, 0, 1, 1, 1, 1, 1], dtype=bool)
    rb.append(x0 + 2, k0, None, v2)
    assert rb.x.shape == (3, 3, n)
    assert np.array_equal(rb.valid[2], v1 & v2)                  # cumulative (ray.py:100)
    rb2 = copy.deepcopy(rb)
    rb2.append(x0 + 3, k0, None, np.ones(n, dtype=bool))
    assert rb.x.shape[0] == 3 and rb2.x.shape[0] == 4            # independent histories
    ids = np.array([10, 11, 12, 13, 14, 15, 16])
    rb3 = api.RayBundle(x0, k0.astype(complex), np.ones((3, n)) * (1 + 1j), rayID=ids, wave=0.6e-3, splitted=True)
    assert np.array_equal(rb3.rayID, ids) and np.iscomplexobj(rb3.k) and np.iscomplexobj(rb3.Efield)
    assert rb3.splitted and rb3.wave == 0.6e-3
    dev = rb.device
    rb4 = api.RayBundle(torch.from_numpy(x0).to(dev), torch.from_numpy(k0).to(dev), None)
    assert np.array_equal(rb4.x[0], x0) and rb4.x_dev().is_cuda
    with pytest.raises(ValueError):
        api.RayBundle(x0, k0[:, :3], None)
    with pytest.raises(NotImplementedError):
        api.RayBundle(x0, k0 + 0.1j, None)                       # a start inside an absorbing medium


def test_custom_ray_ids_survive_compaction(api):
    case = _golden.load_case("doublet_clipped")
    (s, seq) = zoo.doublet(api)
    ids = np.arange(case.x0.shape[1]) * 3 + 100
    ib = api.RayBundle(x0=case.x0, k0=case.k0, Efield0=case.E0, rayID=ids, wave=case.wave)
    rp = s.seqtrace(ib, seq)[0]
    for (rb, ref) in zip(rp.raybundles, case.raw_bundles):
        assert np.array_equal(rb.rayID, ref["id"] * 3 + 100)


def test_raybundle_local_helpers(api):
    """returnLocalComponents / returnLocalD / getLocalSurfaceNormal (ray.py:118-161)"""
    (s, seq) = zoo.tilted(api)
    case = _golden.load_case("tilted_frames")
    rp = s.seqtrace(bundle_of(api, case), seq)[0]
    rb = rp.raybundles[2]                       # bundle created at surface s1
    surf = s.elements["tilted"].surfaces["s1"]
    mat = s.elements["tilted"].materials["glass"]
    (xl, kl, el) = rb.returnLocalComponents(surf.shape.lc, 0)
    # the hit points lie on the conic in the shape frame
    z = oracle_sag(case.table[0]["shape"], xl[0], xl[1])
    assert np.allclose(xl[2], z, rtol=0, atol=1e-12)
    n = rb.getLocalSurfaceNormal(surf, mat, rb.x[0])
    assert np.allclose(np.sum(n * n, axis=0), 1.0)
    # Snell in the material frame: the tangential component of k is continuous
    k_in = mat.lc.returnGlobalToLocalDirections(np.real(rp.raybundles[1].k[1]))[:, rp.raybundles[1].valid[1]]
    ids_in = rp.raybundles[1].rayID[rp.raybundles[1].valid[1]]
    sel = np.isin(ids_in, rb.rayID)
    k_out = mat.lc.returnGlobalToLocalDirections(np.real(rb.k[0]))
    t_in = k_in[:, sel] - np.sum(k_in[:, sel] * n, axis=0) * n
    t_out = k_out - np.sum(k_out * n, axis=0) * n
    assert np.allclose(t_in, t_out, rtol=0, atol=1e-12)
    d = rb.returnLocalD(surf.shape.lc, 0)
    assert np.allclose(np.sum(d * d, axis=0), 1.0)


def oracle_sag(shape, x, y):
    from oracle import seqtrace_np
    return seqtrace_np.shape_sag(shape, x, y)


@pytest.mark.parametrize("name,which", [("aniso_doublet_uniaxial_split", "doublet"),
                                        ("aniso_mirror_uniaxial_split", "mirror")])
def test_dropin_splitup_on_foreign_objects_carves_the_paths_out_of_the_dense_trace(api, name, which):
    """dropin.seqtrace(..., splitup=True) on an object graph that has no methods of this package (here: stand-ins
    without ``_seqtrace_generic``): the four forked paths come out of ONE dense trace, in the reference's order"""
    import types
    from pyrate_amd import dropin
    case = _golden.load_case(name)
    if which == "doublet":
        (s, seq) = aniso_system(api, "uni")
    else:
        (s, seq) = zoo.crystal_mirror(api, np.asarray(case.table[1]["material"]["eps_re"]))
    foreign = types.SimpleNamespace(elements=s.elements, material_background=s.material_background)
    rpaths = dropin.seqtrace(foreign, bundle_of(api, case), seq, splitup=True)
    assert len(rpaths) == case.npaths == 4
    for (rp, ref) in zip(rpaths, _raw_paths(case)):
        assert_paths_match(rp, ref)


def test_system_update_in_place_equals_a_new_system(gpu_device):
    """prt_system_update (DeviceSystem.update): a table replaced in place, stream-ordered, gives exactly the trace of
    a system created from that table -- conic, asphere (longer / shorter coefficient lists within the capacity) and
    crystal tables (hot blocks + walk program); a table that does not fit is refused and nothing changes; traces
    enqueued before an update see the old table"""
    import torch
    from pyrate_amd import engine, systems
    (o, k, e0) = systems.double_gauss_bundle(4000, field_deg=3.0)
    (x0, k0, e0d) = [engine.to_device_rays(a, gpu_device) for a in (o, k, e0)]

    def trace(sysd):
        r = sysd.trace(x0, k0, e0d)
        S = sysd.n_surfaces
        return [r.x_hit[s].clone() for s in range(S)] + [r.k_out[s].clone() for s in range(S)] + \
               [r.valid_out[s].clone() for s in range(S)]

    def same(a, b):
        return all(torch.equal(torch.nan_to_num(p), torch.nan_to_num(q)) for (p, q) in zip(a, b))
    recs_a = systems.double_gauss_records()
    recs_b = systems.double_gauss_records(486.1e-6)
    sys_a = engine.DeviceSystem(recs_a, 0)
    ref_a = trace(sys_a)
    ref_b = trace(engine.DeviceSystem(recs_b, 0))
    assert not same(ref_a, ref_b)
    queued = sys_a.trace(x0, k0, e0d)                    # enqueued BEFORE the update: must see table a
    assert sys_a.update(recs_b)
    got_b = trace(sys_a)
    assert same(got_b, ref_b)
    assert same([queued.x_hit[-1], queued.k_out[-1]], [ref_a[11], ref_a[23]])
    for _ in range(12):                                  # the staging ring wraps around
        assert sys_a.update(recs_a) and same(trace(sys_a), ref_a)
        assert sys_a.update(recs_b) and same(trace(sys_a), ref_b)
    assert not sys_a.update(recs_a[:-1])                 # another number of surfaces: refused, nothing changed
    assert not sys_a.update(systems.asphere_records() + recs_a[4:])    # needs a longer side array than allocated
    assert same(trace(sys_a), ref_b)
    # aspheres: fewer coefficients fit the arrays of more
    (o4, k4, e4) = systems.double_gauss_bundle(3000, rpup=9.0, z0=-5.0, field_deg=5.0)
    (x4, k4d, e4d) = [engine.to_device_rays(a, gpu_device) for a in (o4, k4, e4)]
    big = systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8, 1e-11), curv=-1. / 30., cc=-1.5)
    small = systems.asphere_records(coefficients=(5e-4, -2e-6), curv=-1. / 40., cc=-1.0)
    s_big = engine.DeviceSystem(big, 0)
    want = engine.DeviceSystem(small, 0).trace(x4, k4d, e4d)
    assert s_big.update(small)
    got = s_big.trace(x4, k4d, e4d)
    assert all(torch.equal(torch.nan_to_num(got.x_hit[q]), torch.nan_to_num(want.x_hit[q])) and
               torch.equal(torch.nan_to_num(got.k_out[q]), torch.nan_to_num(want.k_out[q])) for q in range(4))
    # crystals: another pair of tensors, same structure
    c = systems.CALCITE_TILTED
    rec1 = systems.aniso_doublet_records(systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]), systems.uniaxial_eps(1.67, 1.60, (0.2, 0.0, 0.98)))
    rec2 = systems.aniso_doublet_records(systems.uniaxial_eps(1.60, 1.52, (0.0, 0.1, 0.99)), systems.uniaxial_eps(1.70, 1.62, (0.3, 0.0, 0.95)))
    (oc, kc, ec) = systems.double_gauss_bundle(2000, rpup=11.0, z0=-5.0, field_deg=1.0)
    (xc, kcd, ecd) = [engine.to_device_rays(a, gpu_device, pitched=False) for a in (oc, kc, ec)]
    s1 = engine.DeviceSystem(rec1, 0)
    want = engine.DeviceSystem(rec2, 0).trace(xc, kcd, ecd)
    assert s1.update(rec2)
    got = s1.trace(xc, kcd, ecd)
    assert all(torch.equal(torch.nan_to_num(got.x_hit[q]), torch.nan_to_num(want.x_hit[q])) and
               torch.equal(torch.nan_to_num(got.k_out[q]), torch.nan_to_num(want.k_out[q])) for q in range(5))
    assert float((got.k_out[2] - engine.DeviceSystem(rec1, 0).trace(xc, kcd, ecd).k_out[2]).abs().nan_to_num().max()) > 1e-3
    assert not s1.update(systems.doublet_records())      # no crystals: another structure (no walk program)


def test_dispatch_recycles_device_systems_for_tables_never_seen_before(gpu_device):
    """the optimiser's pattern through the drop-in layer: every call a new table.  The device-system cache overwrites
    its oldest entry in place (same results as fresh systems; the cache does not grow; tables that alternate are
    still found)"""
    from pyrate_amd import systems
    from pyrate_amd.builders import build_rotationally_symmetric_optical_system
    from pyrate_amd.raytracer import _dispatch
    from pyrate_amd.raytracer.ray import RayBundle
    _dispatch.clear()
    (s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    (o, k, e0) = systems.double_gauss_bundle(2000, field_deg=2.0)
    ib = RayBundle(o, k, e0, wave=systems.DLINE)
    curv = s.elements["stdelem"].surfaces["lens1front"].shape.curvature
    c0 = curv()
    outs = []
    for i in range(40):
        curv.set_value(c0 * (1.0 + 1e-4 * i))
        outs.append(s.seqtrace(ib, seq)[0].raybundles[-1].x[-1].copy())
    assert len(_dispatch._CACHE) <= _dispatch._RECYCLE_FROM + 1
    _dispatch.clear()
    for i in (39, 0, 17):
        curv.set_value(c0 * (1.0 + 1e-4 * i))
        assert np.array_equal(s.seqtrace(ib, seq)[0].raybundles[-1].x[-1], outs[i])       # fresh systems: same bits
    curv.set_value(c0)


def test_a_failed_in_place_update_poisons_the_system_and_leaves_the_caches(gpu_device, monkeypatch):
    """prt_system_update enqueues up to five copies; a failure after the first leaves the device with
    pieces of two tables.  The library then refuses the system for good (PRT_ERR_DEVICE from every entry point),
    DeviceSystem.update closes it, and the dispatch cache of the drop-in layer drops it -- the next trace builds a
    fresh system and is right.  (PRT_TEST_FAIL_UPDATE=k -- honoured only with PRT_TEST_HOOKS, conftest.py -- makes the
    k-th enqueue of an update fail instead of being issued: k = 2 leaves the new records on the stream and the old
    coefficient arrays in place, the partial update the poisoning exists for.)"""
    import ctypes
    from pyrate_amd import _lib, engine, systems
    from pyrate_amd.builders import build_rotationally_symmetric_optical_system
    from pyrate_amd.raytracer import _dispatch
    from pyrate_amd.raytracer.ray import RayBundle
    (o, k, e0) = systems.double_gauss_bundle(1500, field_deg=2.0)
    (x0, k0, e0d) = [engine.to_device_rays(a, gpu_device) for a in (o, k, e0)]
    sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
    ref = sysd.trace(x0, k0, e0d).x_hit[-1].clone()
    handle = sysd._h
    monkeypatch.setenv("PRT_TEST_FAIL_UPDATE", "2")
    with pytest.raises(_lib.PrtError) as err:
        sysd.update(systems.double_gauss_records(486.1e-6))
    assert err.value.code == _lib.ERR_DEVICE and "unusable" in str(err.value)
    assert sysd._h is None                                          # closed: nobody can launch on the mixed table
    monkeypatch.delenv("PRT_TEST_FAIL_UPDATE")
    # the library's own refusal, on a system kept alive by hand
    lib = _lib.load()
    sys2 = engine.DeviceSystem(systems.double_gauss_records(), 0)
    monkeypatch.setenv("PRT_TEST_FAIL_UPDATE", "1")
    table = engine.pack_table(systems.double_gauss_records(486.1e-6))
    rc = lib.prt_system_update(sys2._h, table, sys2.n_surfaces, ctypes.c_void_p(0))
    monkeypatch.delenv("PRT_TEST_FAIL_UPDATE")
    assert rc == _lib.ERR_DEVICE
    with pytest.raises(_lib.PrtError) as err:
        sys2.trace(x0, k0, e0d)
    assert "unusable" in str(err.value)
    assert lib.prt_system_update(sys2._h, table, sys2.n_surfaces, ctypes.c_void_p(0)) == _lib.ERR_DEVICE
    # through the drop-in layer: the recycled entry that failed is gone from the caches, the next call is right
    _dispatch.clear()
    (s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    ib = RayBundle(o, k, e0, wave=systems.DLINE)
    curv = s.elements["stdelem"].surfaces["lens1front"].shape.curvature
    c0 = curv()
    for i in range(_dispatch._RECYCLE_FROM + 2):
        curv.set_value(c0 * (1.0 + 1e-4 * i))
        s.seqtrace(ib, seq)
    monkeypatch.setenv("PRT_TEST_FAIL_UPDATE", "1")
    curv.set_value(c0 * 1.01)
    with pytest.raises(_lib.PrtError):
        s.seqtrace(ib, seq)
    monkeypatch.delenv("PRT_TEST_FAIL_UPDATE")
    assert all(v._h for v in _dispatch._CACHE.values())
    curv.set_value(c0)
    got = s.seqtrace(ib, seq)[0].raybundles[-1].x[-1]
    assert np.array_equal(got, ref.cpu().numpy())          # same bundle, same table: the same bits
    _dispatch.clear()
    del handle


def test_a_hand_built_record_list_edited_in_place_is_keyed_by_content(gpu_device):
    """the identity fast path of the device-system cache stands for content only for the records
    surface_table's memo owns; a caller's own record dictionaries may be edited between two traces"""
    import copy
    from pyrate_amd import dropin, systems
    from pyrate_amd.raytracer import _dispatch
    _dispatch.clear()
    recs = copy.deepcopy(systems.double_gauss_records())
    a = _dispatch.system_for(recs, gpu_device)
    assert _dispatch.system_for(recs, gpu_device) is a               # same content: the cached system
    recs[0]["shape"]["curv"] *= 1.05                                 # edited IN PLACE: same list, same dictionaries
    b = _dispatch.system_for(recs, gpu_device)
    assert b is not a and b.records[0]["shape"]["curv"] == recs[0]["shape"]["curv"]
    _dispatch.clear()
