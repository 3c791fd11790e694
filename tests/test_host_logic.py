"""CPU-only tests of the host side: the mirror classes flatten to the same surface
tables as the real reference objects (tables stored in the golden fixtures), table
packing, the C ABI exports every symbol of include/prt.h, frame arithmetic."""
import ctypes
import json
import math
import os
import re

import numpy as np
import pytest

import _golden
import systems_zoo as zoo
from pyrate_amd import surface_table as st
from pyrate_amd import systems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flatten(s, seq, wave):
    return st.flatten_sequence(s, seq, wave)


def _assert_tables_equal(a, b):
    assert len(a) == len(b)
    for (ra, rb) in zip(a, b):
        assert json.dumps(ra, sort_keys=True) == json.dumps(rb, sort_keys=True)


@pytest.fixture(scope="module")
def api():
    return zoo.mirror_api()


def test_mirror_doublet_table_equals_reference(api):
    (s, seq) = zoo.doublet(api)
    (recs, lengths) = _flatten(s, seq, zoo.DLINE)
    _assert_tables_equal(recs, _golden.load_case("doublet").table)
    assert lengths == [5]


def test_mirror_double_gauss_table_equals_reference(api):
    (s, seq) = api.build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("double_gauss_axis").table)
    _assert_tables_equal(systems.double_gauss_records(), _golden.load_case("double_gauss_axis").table)
    (s, seq) = api.build_rotationally_symmetric_optical_system(systems.double_gauss_tuples(486.1e-6))
    _assert_tables_equal(_flatten(s, seq, 486.1e-6)[0], _golden.load_case("double_gauss_Fline").table)


def test_mirror_tilted_frames_table_equals_reference(api):
    """tilt conventions, both tilt orders, ModelGlass dispersion at the C line"""
    (s, seq) = zoo.tilted(api)
    ref = _golden.load_case("tilted_frames")
    (recs, _) = _flatten(s, seq, ref.wave)
    for (a, b) in zip(recs, ref.table):
        for key in ("B_shape", "g_shape", "B_ap", "g_ap", "B_mat"):
            assert np.allclose(np.asarray(a[key]), np.asarray(b[key]), rtol=0, atol=1e-15), key
        assert a["aperture"] == b["aperture"] and a["shape"] == b["shape"]
        assert a["material"]["n"] == pytest.approx(b["material"]["n"], rel=1e-15)


def test_mirror_explicit_and_mirror_tables(api):
    (s, seq) = api.build_simple_optical_system(systems.asphere_builduplist([1e-3, -1e-6, 1e-8], -1. / 30., -1.5))
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("asphere_strong_axis").table)
    (s, seq) = api.build_simple_optical_system(zoo.xypoly_builduplist())
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("xypoly_axis").table)
    (s, seq) = api.build_simple_optical_system(zoo.biconic_builduplist())
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("biconic_axis").table)
    (s, seq) = api.build_simple_optical_system(zoo.mirrors_builduplist())
    ref = _golden.load_case("mirrors").table
    recs = _flatten(s, seq, zoo.DLINE)[0]
    assert [r["interaction"] for r in recs] == [r["interaction"] for r in ref] == \
        ["refract", "mirror", "mirror", "refract"]
    for (a, b) in zip(recs, ref):
        assert np.allclose(np.asarray(a["B_shape"]), np.asarray(b["B_shape"]), rtol=0, atol=1e-15)
        assert np.allclose(np.asarray(a["g_shape"]), np.asarray(b["g_shape"]), rtol=0, atol=1e-13)


def test_mirror_two_elements_and_aniso_tables(api):
    (s, seq) = zoo.two_element_system(api)
    (recs, lengths) = _flatten(s, seq, zoo.DLINE)
    _assert_tables_equal(recs, _golden.load_case("two_elements").table)
    assert lengths == [2, 3]
    c = systems.CALCITE_TILTED
    eps1 = systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"])
    eps2 = systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))
    (s, seq) = zoo.aniso_doublet(api, eps1, eps2)
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("aniso_doublet_uniaxial").table)
    _assert_tables_equal(systems.aniso_doublet_records(eps1, eps2), _golden.load_case("aniso_doublet_uniaxial").table)


def test_structural_errors_raise_like_the_reference(api):
    s = api.OpticalSystem.p()
    lc_loose = api.LocalCoordinates.p(name="loose")
    elem = api.OpticalElement.p(lc_loose, name="e")
    with pytest.raises(Exception):
        s.addElement("e", elem)                   # optical_system.py:224-227
    lc0 = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="a"), refname=s.rootcoordinatesystem.name)
    elem = api.OpticalElement.p(lc0, name="e")
    with pytest.raises(Exception):
        elem.addSurface("x", api.Surface.p(lc_loose), (None, None))      # optical_element.py:73-76
    with pytest.raises(Exception):
        elem.addMaterial("m", api.ConstantIndexGlass.p(lc_loose, 1.5))   # :104-107
    with pytest.raises(Exception):
        api.Surface.p(lc0, shape=api.Conic.p(lc_loose))                  # surface.py:105-108


def test_unsupported_shapes_are_rejected():
    class UserDll(object):
        kind = "shape_ZMXDLLShape"
    with pytest.raises(st.UnsupportedError):
        st.describe_shape(UserDll())


def test_pack_table_flags_and_coefficients():
    recs = _golden.load_case("tilted_frames").table
    tab = st.pack_table(recs)
    assert tab[0].frame_flags == 0                       # tilted shape, own aperture frame, tilted material
    assert tab[0].ap_type == 2 and tab[1].ap_type == 1 and tab[2].ap_type == 0
    assert tab[1].ap_p0 == 0.8 and tab[1].ap_p1 == 5.5
    assert tab[2].frame_flags & st.FRAME_AP_IS_SHAPE
    recs = _golden.load_case("double_gauss_axis").table
    tab = st.pack_table(recs)
    assert all(t.frame_flags == 7 for t in tab)
    recs = _golden.load_case("xypoly_axis").table
    tab = st.pack_table(recs)
    assert tab[2].shape_type == 2 and tab[2].n_coeffs == 6
    assert tab[2].coeffs[0] == -0.12 * (1. / 10.0 ** 2) and (tab[2].xpow[0], tab[2].ypow[0]) == (0, 2)
    recs = _golden.load_case("asphere_mild_axis").table
    tab = st.pack_table(recs)
    assert tab[2].shape_type == 1 and tab[2].n_coeffs == 3 and tab[2].coeffs[1] == 1e-7


def test_classify_eps():
    assert st.classify_eps(2.25 * np.eye(3), np.zeros((3, 3)))[0] == st.ANISO_ISOTROPIC
    eps = systems.uniaxial_eps(1.658, 1.486, (0.0, math.sin(0.3), math.cos(0.3)))
    (cls, eo, ee, axis) = st.classify_eps(eps, np.zeros((3, 3)))
    assert cls == st.ANISO_UNIAXIAL
    assert eo == pytest.approx(1.658 ** 2, rel=1e-14) and ee == pytest.approx(1.486 ** 2, rel=1e-14)
    assert abs(abs(np.dot(axis, [0.0, math.sin(0.3), math.cos(0.3)])) - 1) < 1e-14
    rec = np.diag([2.4, 2.56, 2.8])
    assert st.classify_eps(rec, np.zeros((3, 3)))[0] == st.ANISO_GENERAL
    # a complex (absorbing) tensor has no class of its own: the complex solver takes it as it is
    assert st.classify_eps(np.eye(3), 0.1 * np.eye(3))[0] == st.ANISO_GENERAL


def test_c_abi_exports_every_declared_symbol():
    """libprt.so loads without a GPU and exports exactly what include/prt.h declares"""
    from pyrate_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "prt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(prt_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.PROTOTYPES.keys())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.prt_abi_version() == _lib.ABI_VERSION == 7
    assert lib.prt_sizeof_surface() == ctypes.sizeof(st.PrtSurface)
    assert lib.prt_strerror(-2) == b"unsupported shape/material"
    # argument validation happens before any device work
    assert lib.prt_system_create(None, 0, 0, None) == -1
    assert lib.prt_trace(None, 0, 0, None, None, None, None, 0, 0, None, None, None, None, None, None) == -1
    assert lib.prt_trace_seq(None, 0, 0, None, None, None, None, 0, None, None, None, None, 0, None) == -1
    # prt_trace_ex: the struct must carry this library's sizeof (a caller built against another layout is refused)
    assert lib.prt_sizeof_trace_args() == ctypes.sizeof(_lib.PrtTraceArgs)
    args = _lib.PrtTraceArgs()
    assert lib.prt_trace_ex(None, None) == -1 and lib.prt_trace_ex(None, ctypes.byref(args)) == -1
    assert b"struct_bytes" in lib.prt_last_error()
    args.struct_bytes = ctypes.sizeof(_lib.PrtTraceArgs)
    assert lib.prt_trace_ex(None, ctypes.byref(args)) == -1 and b"null system" in lib.prt_last_error()
    assert lib.prt_recommended_pitch(9994476) == 9994752 and lib.prt_recommended_pitch(512) == 512
    assert lib.prt_compact_scratch_bytes(0) > 0
    assert lib.prt_arena_alloc(None, 0, None, None, None, -1, 0, -1, None) == -1 and lib.prt_arena_free(None, None, None) == -1


def test_localcoordinates_roundtrips(api):
    """the properties reference tests/test_localcoordinates.py:63-214 pins"""
    rng = np.random.RandomState(0)
    root = api.LocalCoordinates.p(name="root")
    lc1 = root.addChild(api.LocalCoordinates.p(name="a", decx=0.3, decy=-1.0, decz=5.0,
                                               tiltx=0.2, tilty=-0.4, tiltz=1.1))
    lc2 = lc1.addChild(api.LocalCoordinates.p(name="b", decz=2.0, tiltx=-0.7, tiltThenDecenter=1))
    pts = rng.rand(3, 20)
    for lc in (lc1, lc2):
        assert np.allclose(lc.returnGlobalToLocalPoints(lc.returnLocalToGlobalPoints(pts)), pts)
        assert np.allclose(lc.returnGlobalToLocalDirections(lc.returnLocalToGlobalDirections(pts)), pts)
        assert np.allclose(np.dot(lc.localbasis.T, lc.localbasis), np.eye(3))
        d = lc.returnLocalToGlobalDirections(pts)
        assert np.allclose(np.sum(d * d, axis=0), np.sum(pts * pts, axis=0))
    assert np.allclose(lc1.returnOtherToActualPoints(lc1.returnActualToOtherPoints(pts, lc2), lc2), pts)


def test_rect_grid_matches_reference_raster():
    """RectGrid: 1e4 requested -> 9 917 points"""
    (px, py) = systems.rect_grid(10000)
    assert px.shape[0] == 9917
    assert np.all(px ** 2 + py ** 2 <= 1)
    case = _golden.load_case("double_gauss_axis")
    (o, k, e0) = systems.double_gauss_bundle(256)
    assert np.array_equal(o, case.x0) and np.array_equal(k, case.k0) and np.array_equal(e0, case.E0)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """no CPU fallback: without libprt.so every product entry point raises ImportError"""
    from pyrate_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libprt.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    from pyrate_amd import engine
    with pytest.raises(ImportError):
        engine.DeviceSystem(_golden.load_case("doublet").table, 0)


def test_no_product_module_imports_the_oracle():
    """the oracle is test infrastructure: nothing under pyrate_amd/ may import it"""
    import glob
    for path in glob.glob(os.path.join(ROOT, "pyrate_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "seqtrace_np" not in src and "seqtrace_c" not in src, path


def test_no_product_module_knows_the_host_build_of_the_kernels():
    """tests/hostemu (libprt's sources compiled for the host: the sanitizer build of the `-m "not gpu"` suite) is test
    infrastructure too: nothing under pyrate_amd/, bench.py, benchmarks/ or __graft_entry__.py mentions it, and the
    library path of the product cannot be pointed at it by accident (another file name, built by tests/ only)"""
    import glob
    paths = glob.glob(os.path.join(ROOT, "pyrate_amd", "**", "*.py"), recursive=True) + \
        glob.glob(os.path.join(ROOT, "benchmarks", "*.py")) + glob.glob(os.path.join(ROOT, "pyrate_amd", "csrc", "*")) + \
        [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for path in paths:
        if path.endswith((".so", ".json")):
            continue
        assert "hostemu" not in open(path, errors="replace").read(), path


def test_catalog_material_dispersion_matches_reference(api):
    """CatalogMaterial n(wavelength) for every dispersion formula type == the reference's
    material_glasscat.py (values generated by oracle/make_golden.py: tests/golden/dispersion.json);
    N-BK7: nd = 1.5168 (public SCHOTT value)"""
    from pyrate_amd.raytracer.material.material_glasscat import CatalogMaterial
    gold = json.load(open(os.path.join(_golden.GOLDEN_DIR, "dispersion.json")))
    lc = api.LocalCoordinates.p(name="d")
    for (key, page) in gold["pages"].items():
        mat = CatalogMaterial.p(lc, page)
        n = [mat.get_optical_index(None, w) for w in gold["waves_mm"]]
        assert np.allclose(n, gold["n"][key], rtol=1e-15, atol=0), key
    bk7 = CatalogMaterial.p(lc, gold["pages"]["formula1_nbk7"])
    assert abs(bk7.get_optical_index(None, 0.5875618e-3) - 1.5168) < 2e-5
    with pytest.raises(Exception):
        bk7.get_optical_index(None, 5e-3)                    # wavelength out of range
    # a catalogue dictionary as material in the builders (pyrateoptics/__init__.py:193-196)
    (s, seq) = api.build_simple_optical_system([
        ({"shape": "Conic", "curv": 0.01}, {"decz": 1.0}, gold["pages"]["formula1_nbk7"], "f", {}),
        ({"shape": "Conic"}, {"decz": 3.0}, None, "b", {})])
    recs = st.flatten_sequence(s, seq, 0.5876e-3)[0]
    assert abs(recs[0]["material"]["n"] - gold["n"]["formula1_nbk7"][2]) < 1e-15


def _packed_polynomial(rec_shape, x, y):
    """evaluate what pack_record stored for a polynomial-type shape (the device formula, in NumPy)"""
    r = st.pack_record({"shape": rec_shape, "aperture": {"type": "none"}, "interaction": "refract",
                        "material": {"type": "isotropic", "n": 1.0},
                        "B_shape": np.eye(3).tolist(), "g_shape": [0, 0, 0], "B_ap": np.eye(3).tolist(),
                        "g_ap": [0, 0, 0], "B_mat": np.eye(3).tolist()})
    first = r.n_asphere if r.shape_type == 4 else 0
    F = np.zeros_like(x)
    for t in range(first, r.n_coeffs):
        F = F + r.coeffs[t] * x ** r.xpow[t] * y ** r.ypow[t]
    if r.shape_type == 4 and r.asphere_scale != 0.0:
        from oracle import seqtrace_np as oracle
        F = F + r.asphere_scale * oracle.asphere_F(r.curv, r.cc, [r.coeffs[q] for q in range(r.n_asphere)], x, y)
    return (F, r)


def test_zernike_and_combination_shapes_equal_reference():
    """(i) the oracle's Zernike / LinearCombination sag == the reference's getSag, its gradient ==
    the derivative of that sag; (ii) the monomial expansion the device evaluates == the same sag"""
    from oracle import seqtrace_np as oracle
    z = np.load(os.path.join(_golden.GOLDEN_DIR, "zernike_shapes.npz"))
    recs = json.loads(str(z["records_json"]))
    (x, y) = (z["x"], z["y"])
    for key in ("fringe", "ansi", "combination"):
        assert np.allclose(oracle.shape_sag(recs[key], x, y), z[key + "_sag"], rtol=0, atol=1e-14)
        # gradient: the derivative of the REFERENCE'S sag (4th-order central differences taken with
        # the reference), which the reference's own gradzernike is not (off by 2e-3 here)
        g = oracle.shape_grad(recs[key], x, y)
        assert np.allclose(-g[0], z[key + "_dsag_dx"], rtol=0, atol=1e-12)
        assert np.allclose(-g[1], z[key + "_dsag_dy"], rtol=0, atol=1e-12)
        assert np.abs(-z[key + "_grad"][0] - z[key + "_dsag_dx"]).max() > 1e-3
        (F, r) = _packed_polynomial(recs[key], x, y)
        assert np.allclose(F, z[key + "_sag"], rtol=0, atol=2e-14), key
        assert r.n_coeffs <= st.PRT_MAX_COEFFS
    assert recs["combination"]["parts"][1]["offset"] == [0.7, -1.1, 0.0]


def test_combination_part_rotated_about_the_axis_equals_reference(api):
    """LinearCombination with a polynomial part in a frame decentred AND rotated about the combination's z axis
    (surface_shape.py:709-748): the oracle's sag / gradient == the reference's getSag / getGrad; the monomial
    expansion the device evaluates (rotate, then shift) == the same sag; the mirror classes flatten to the
    reference's record; a part TILTED against the axis is refused (the reference's gradF is then not the gradient
    of its F)"""
    from oracle import seqtrace_np as oracle
    from pyrate_amd import polyshape
    z = np.load(os.path.join(_golden.GOLDEN_DIR, "rotated_combination_shape.npz"))
    rec = json.loads(str(z["record_json"]))
    (x, y) = (z["x"], z["y"])
    assert "rot" in rec["parts"][1] and "rot" not in rec["parts"][0]
    assert np.allclose(oracle.shape_sag(rec, x, y), z["sag"], rtol=0, atol=1e-14)
    assert np.allclose(oracle.shape_grad(rec, x, y), z["grad"], rtol=0, atol=1e-14)
    (F, r) = _packed_polynomial(rec, x, y)
    assert np.allclose(F, z["sag"], rtol=0, atol=2e-14) and r.n_coeffs <= st.PRT_MAX_COEFFS
    # rotated(): a quarter turn maps x -> y, y -> -x
    quarter = polyshape.rotated({(2, 1): 1.0}, ((0.0, -1.0), (1.0, 0.0)))          # xs = y, ys = -x: xs^2 ys = -x y^2
    assert {k: round(v, 15) for (k, v) in quarter.items() if abs(v) > 1e-15} == {(1, 2): -1.0}
    (s, seq) = zoo.rotated_combination_system(api)
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("rotated_combination_lens").table)
    lc = api.LocalCoordinates.p(name="c")
    lct = lc.addChild(api.LocalCoordinates.p(name="tilted", tiltx=0.1))
    lc.update()
    bad = api.LinearCombination.p(lc, list_of_coefficients_and_shapes=[
        (1.0, api.XYPolynomials.p(lct, normradius=1.0, coefficients=[(2, 0, 0.01)]))])
    with pytest.raises(st.UnsupportedError):
        st.describe_shape(bad)


def test_zernike_index_maps_and_monomials():
    from pyrate_amd import polyshape
    assert [polyshape.fringe_nm(j) for j in (1, 2, 3, 4, 5, 9, 16, 36)] == \
        [(0, 0), (1, 1), (1, -1), (2, 0), (2, 2), (4, 0), (6, 0), (10, 0)]
    assert [polyshape.ansi_nm(j) for j in (1, 2, 3, 4, 5, 6)] == [(0, 0), (1, -1), (1, 1), (2, -2), (2, 0), (2, 2)]
    assert polyshape.zernike_monomials(2, 0) == {(2, 0): 2, (0, 2): 2, (0, 0): -1}         # 2 rho^2 - 1
    assert polyshape.zernike_monomials(3, -1) == {(2, 1): 3, (0, 3): 3, (0, 1): -2}        # (3 rho^3 - 2 rho) sin
    shifted = polyshape.shifted({(2, 0): 1.0}, 0.5, 0.0)                                     # (x - 0.5)^2
    assert shifted == {(2, 0): 1.0, (1, 0): -1.0, (0, 0): 0.25}


def test_mirror_zernike_tables_equal_reference(api):
    (s, seq) = api.build_simple_optical_system(zoo.zernike_builduplist("Fringe"))
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("zernike_fringe_field3").table)
    (s, seq) = api.build_simple_optical_system(zoo.zernike_builduplist("ANSI"))
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("zernike_ansi_field2").table)
    (s, seq) = zoo.zernike_combination_system(api)
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("zernike_combination_mirror").table)


def test_mirror_hud_prism_table_equals_reference(api):
    """demos/demo_hud.py built from the mirror classes flattens to the table the reference's own classes gave
    (14 surfaces, every frame hung on the object frame with tiltThenDecenter=False, one biconic face twice); the
    symmetric Zernike lens likewise"""
    from demos import demo_hud
    (s, seq) = demo_hud.build(api)
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("hud_patent_axis").table)
    (s, seq) = api.build_simple_optical_system(zoo.zernike_builduplist("Fringe", symmetric=True))
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("zernike_fringe_symmetric_field3").table)


def test_gridsag_oracle_and_table(api):
    """GridSag: oracle (scipy spline rebuilt from the record's knots / coefficients) == the
    reference's getSag / getGrad; the mirror class flattens to the reference's record"""
    from oracle import seqtrace_np as oracle
    z = np.load(os.path.join(_golden.GOLDEN_DIR, "gridsag_shape.npz"))
    rec = json.loads(str(z["record_json"]))
    assert np.array_equal(oracle.shape_sag(rec, z["x"], z["y"]), z["sag"])
    assert np.array_equal(oracle.shape_grad(rec, z["x"], z["y"]), z["grad"])
    (s, seq) = zoo.gridsag_system(api)
    _assert_tables_equal(_flatten(s, seq, zoo.DLINE)[0], _golden.load_case("gridsag_field2").table)
    table = st.pack_table(_golden.load_case("gridsag_field2").table)
    assert table[1].shape_type == 5 and table[1].grid_nx == 29 and table[1].grid_ny == 25 and table[1].aux


def test_pupil_rasters_equal_reference():
    """every raster of sampling2d/raster.py: same points in the same order (the random one under
    the same NumPy seed); Poisson-disk: minimum distance and coverage"""
    from pyrate_amd.sampling2d import raster
    ref = json.load(open(os.path.join(_golden.GOLDEN_DIR, "rasters.json")))
    cases = {"rect_300": (raster.RectGrid(), (300,)), "rect_17": (raster.RectGrid(), (17,)),
             "hex_200": (raster.HexGrid(), (200,)), "hex_31": (raster.HexGrid(), (31,)),
             "meridional_9_0": (raster.MeridionalFan(), (9, 0.)), "meridional_8_30": (raster.MeridionalFan(), (8, 30.)),
             "sagital_7_0": (raster.SagitalFan(), (7, 0.)), "sagital_6_45": (raster.SagitalFan(), (6, 45.)),
             "chiefcoma_20": (raster.ChiefAndComa(), (1, 20.)), "single": (raster.Single(), (1, 0.25, -0.5)),
             "circular_100_eq": (raster.CircularGrid(), (100, True)),
             "circular_50_area": (raster.CircularGrid(), (50, False))}
    for (key, (obj, args)) in cases.items():
        (x, y) = obj.getGrid(*args)
        assert np.array_equal(x, np.array(ref[key]["x"])) and np.array_equal(y, np.array(ref[key]["y"])), key
    np.random.seed(12345)
    (x, y) = raster.RandomGrid().getGrid(500)
    assert np.array_equal(x, np.array(ref["random_500_seed12345"]["x"]))
    assert np.array_equal(y, np.array(ref["random_500_seed12345"]["y"]))
    np.random.seed(3)
    (x, y) = raster.PoissonDiskSampling().getGrid(300)
    r = 1. / int(round(math.sqrt(300 * 4.0 / math.pi)))
    d2 = (x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2 + np.eye(len(x))
    assert d2.min() >= r * r * (1 - 1e-12) and np.all(x ** 2 + y ** 2 <= 1)
    assert len(x) > 150            # (the minimum distance 1/n_per_dim packs about 2.6 nray points, in the reference too)
    assert raster.Single(0.1, 0.2).getGrid(1) == (np.array([0.1]), np.array([0.2]))


def test_small_api_completions(api):
    """aperture predicates, Euler factorisations, tensor transforms, Surface accessors, ModelGlass
    from catalogue numbers"""
    lc = api.LocalCoordinates.p(name="f", decx=0.3, tiltx=0.2, tilty=-0.1, tiltz=0.4)
    child = lc.addChild(api.LocalCoordinates.p(name="c", decz=2.0, tilty=0.3, tiltThenDecenter=1))
    assert lc.getChildren() == [child] and "c (" in lc.pprint()
    for order in (0, 1):
        m = lc.calculateMatrixFromTilt(0.2, -0.1, 0.4, order)
        assert np.allclose(lc.calculateTiltFromMatrix(m, order), (0.2, -0.1, 0.4), atol=1e-15)
    t = np.random.RandomState(1).rand(3, 3, 5)
    g = child.returnLocalToGlobalTensors(t)
    assert np.allclose(g[:, :, 2], child.localbasis.dot(t[:, :, 2]).dot(child.localbasis.T))
    assert np.allclose(child.returnGlobalToLocalTensors(g), t)
    assert np.allclose(lc.returnOtherToActualTensors(child.returnActualToOtherTensors(t, lc), lc), lc.returnGlobalToLocalTensors(child.returnLocalToGlobalTensors(t)))
    (x, y) = (np.array([0., 3., 5.1, -2.]), np.array([0., 4., 0., 2.]))
    assert list(api.CircularAperture.p(lc, maxradius=5., minradius=1.).are_points_in_aperture(x, y)) == [False, True, False, True]
    assert list(api.RectangularAperture.p(lc, width=6., height=5.).are_points_in_aperture(x, y)) == [True, False, False, True]
    from pyrate_amd.raytracer.aperture import BaseAperture
    assert BaseAperture.p(lc).are_points_in_aperture(x, y).all()
    surf = api.Surface.p(lc, shape=api.Conic.p(lc, curv=0.02))
    assert surf.getShape() is surf.shape and surf.getAperture() is surf.aperture and surf.getCentralCurvature() == 0.02
    mg = api.ModelGlass.p(lc)
    mg.calcCoefficientsFrom_nd_vd(1.5168, 64.17)
    (nd, nF, nC) = [mg.get_optical_index(None, w) for w in (0.5875618e-3, 0.4861327e-3, 0.6562725e-3)]
    assert abs(nd - 1.5168) < 2e-6 and abs((nd - 1) / (nF - nC) - 64.17) < 0.05
    mg.calcCoefficientsFromSchottCode(517642)
    assert abs(mg.get_optical_index(None, 0.5875618e-3) - 1.517) < 2e-6


def test_header_is_plain_c_and_links(tmp_path):
    """include/prt.h compiles as C99 (the boundary is a C ABI, not a C++ one) and a C program
    links against libprt.so"""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi_check.c"
    src.write_text("""#include <stddef.h>
#include "prt.h"
int main(void) {
    prt_surface_t table[2];
    prt_system_t *sys = NULL;
    (void)table;
    if (prt_abi_version() != PRT_ABI_VERSION) return 1;
    if (prt_sizeof_surface() != (int32_t)sizeof(prt_surface_t)) return 2;
    (void)sys;
    return 0;
}
""")
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "pyrate_amd", "csrc")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                   check=True)
    exe = tmp_path / "abi_check"
    subprocess.run([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lprt",
                    "-Wl,-rpath," + libdir], check=True)
    # sizeof / version agree between the C compiler's view of the header and the library
    # (runs without a GPU: neither call touches the device)
    assert subprocess.run([str(exe)]).returncode == 0


def test_builder_turns_linear_combination_parts_into_shapes(api):
    """build_simple_optical_system's special case for shape "LinearCombination"
    (pyrateoptics/__init__.py:146-168): (coefficient, surface dictionary) pairs become shapes in the
    surface's frame -- the flattened table equals the one of the object graph built by hand"""
    parts = [(1.0, {"shape": "Asphere", "curv": -1. / 80., "cc": -0.6, "coefficients": [0.0, 3e-6]}),
             (0.5, {"shape": "XYPolynomials", "normradius": 10.0,
                    "coefficients": [(2, 0, 0.02), (0, 2, -0.015), (2, 1, 0.004)]})]
    blist = [({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
             ({"shape": "LinearCombination", "list_of_coefficients_and_shapes": parts},
              {"decz": 12.0}, 1.55, "freeform", {}),
             ({"shape": "Conic"}, {"decz": 40.0}, None, "image", {})]
    (s, seq) = api.build_simple_optical_system(blist)
    (recs, _) = _flatten(s, seq, zoo.DLINE)
    assert recs[1]["shape"]["type"] not in ("conic",)
    elem = next(iter(s.elements.values()))
    combo = elem.surfaces["freeform"].shape
    assert [sh.kind for sh in combo.list_shapes] == ["shape_Asphere", "shape_XYPolynomials"]
    assert combo.annotations["list_shape_coefficients"] == [1.0, 0.5]
    # by hand: the same parts constructed in the surface's frame
    lc = combo.lc
    byhand = api.LinearCombination.p(lc, list_of_coefficients_and_shapes=[
        (1.0, api.Asphere.p(lc, curv=-1. / 80., cc=-0.6, coefficients=[0.0, 3e-6])),
        (0.5, api.XYPolynomials.p(lc, normradius=10.0, coefficients=[(2, 0, 0.02), (0, 2, -0.015), (2, 1, 0.004)]))])
    assert json.dumps(st.describe_shape(byhand), sort_keys=True) == json.dumps(recs[1]["shape"], sort_keys=True)
    st.pack_table(recs)          # and it packs
    # the caller's dictionaries are left as they were (the reference pops from them)
    assert parts[0][1]["shape"] == "Asphere" and blist[1][0]["shape"] == "LinearCombination"


def test_host_bundles_equal_reference_bundles(api):
    """collimated_bundle / divergent_bundle with the reference's signature == the reference's own bundles
    (tests/golden/bundles.json: every deterministic raster, in air and in n = 1.33): origins bit for bit,
    wave vectors to 4e-16 (the reference gets k from a per-ray eigenproblem, this package from k = n * unit
    vector); and the rasters' device tables reproduce getGrid bit for bit as outer products"""
    from pyrate_amd.sampling2d import raster
    ref = json.load(open(os.path.join(_golden.GOLDEN_DIR, "bundles.json")))
    objs = {"rect_60": raster.RectGrid(), "hex_45": raster.HexGrid(), "meridional_9": raster.MeridionalFan(),
            "sagital_8": raster.SagitalFan(), "circular_49": raster.CircularGrid()}
    for (key, case) in ref.items():
        assert abs(case["index"] - (1.33 if key.endswith("n133") else 1.0)) < 1e-15
        (xs, ys) = ([], [])
        for (xa, xb, ya, yb, clip) in objs[case["raster"]].device_tables(case["nray"]):
            (x, y) = ((xb[:, None] * xa[None, :]).reshape(-1), (yb[:, None] * ya[None, :]).reshape(-1))
            keep = (x * x + y * y <= 1) if clip else np.ones(x.shape, dtype=bool)
            xs.append(x[keep])
            ys.append(y[keep])
        (gx, gy) = objs[case["raster"]].getGrid(case["nray"])
        assert np.array_equal(np.hstack(xs), gx) and np.array_equal(np.hstack(ys), gy), key
        assert gx.shape[0] == np.array(case["x"]).shape[1], key
    assert objs["rect_60"].device_tables(60) is not None and raster.RandomGrid().device_tables(60) is None
    assert raster.ChiefAndComa().device_tables(6) is None and raster.Single().device_tables(1) is None

    # the tables belong to the class that defines them: a subclass that overrides ONLY getGrid has its own samples
    # (the reference always calls getGrid) and must not get its parent's tables -- for every raster, not only RectGrid
    for base in (raster.RectGrid, raster.HexGrid, raster.MeridionalFan, raster.SagitalFan, raster.CircularGrid):
        class Own(base):
            def getGrid(self, nray):
                return (np.zeros(3), np.linspace(0, 0.5, 3))
        assert raster.device_tables_of(base(), 40) is not None, base
        assert raster.device_tables_of(Own(), 40) is None, base

        class Both(base):                      # overrides both: its tables are its own again
            def getGrid(self, nray):
                return (np.zeros(3), np.linspace(0, 0.5, 3))

            def device_tables(self, nray):
                return [(np.zeros(1), np.ones(3), np.ones(1), np.linspace(0, 0.5, 3), False)]
        assert raster.device_tables_of(Both(), 40) is not None
    assert raster.device_tables_of(object(), 3) is None


def test_newton_cap_annotation_reaches_the_table(api):
    """``shape.annotations["newton_maxit"]`` is the device-side Newton cap of an explicit shape (the reference's
    ``iterations`` annotation never reaches its solver, surface_shape.py:457, and is not used as the cap);
    absent / 0 leaves the record -- and with it the table flattened from unmodified reference objects -- as it was"""
    (s, seq) = api.build_simple_optical_system(systems.asphere_builduplist())
    (recs, _) = _flatten(s, seq, zoo.DLINE)
    assert all("newton_maxit" not in r for r in recs)
    elem = next(iter(s.elements.values()))
    elem.surfaces["back"].shape.annotations["newton_maxit"] = 7
    elem.surfaces["back"].shape.annotations["iterations"] = 3
    (recs, _) = _flatten(s, seq, zoo.DLINE)
    assert [r.get("newton_maxit", 0) for r in recs] == [0, 0, 7, 0]
    table = st.pack_table(recs)
    assert [table[i].newton_maxit for i in range(4)] == [0, 0, 7, 0]


def test_bench_watchdog_prints_one_json_error_line_and_exits_3():
    """bench.py's watchdog: a process that makes no progress ends with ONE JSON line that has
    an ``error`` field and the stage it was in, on the saved stdout, and exit code 3 -- instead of a hang"""
    import json, os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib.util, sys, threading, time
        sys.argv = ["bench.py"]
        spec = importlib.util.spec_from_file_location("bench", %r)
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        w = b.Watchdog(0.3, int(sys.stdin.readline()), [1], {"metric": "ray_surface_ops_per_s", "n_gpus": 2})
        w.stage = "timed region"
        threading.Event().wait(30)          # a collective that never returns
        print("not reached")
    """ % os.path.join(root, "bench.py"))
    for (rank, n_lines) in ((0, 1), (1, 0)):
        r = subprocess.run([sys.executable, "-c", code], input="%d\n" % rank, capture_output=True, text=True,
                           timeout=120)
        assert r.returncode == 3
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == n_lines and "not reached" not in r.stdout
        if rank == 0:
            line = json.loads(lines[0])
            assert line["value"] is None and "timed region" in line["error"] and line["n_gpus"] == 2
        assert "threading.py" in r.stderr or "Thread" in r.stderr       # the stack dump for the post-mortem
    # and a run that finishes first is left alone
    code_ok = code.replace("threading.Event().wait(30)", "w.done()").replace("not reached", "finished")
    r = subprocess.run([sys.executable, "-c", code_ok], input="0\n", capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "finished"


def test_complex_epsilon_tables_stay_inside_crystals():
    """a complex (absorbing) epsilon tensor is packed like any other (class GENERAL, eps_im filled), a complex index
    goes into eps_im[0]; a table in which an isotropic medium follows an absorbing one BEFORE the last surface is
    refused with the reason (no parity target there)"""
    import copy
    from pyrate_amd import surface_table
    case = _golden.load_case("aniso_absorbing_two_crystals")
    assert surface_table.has_complex_eps(case.table)
    assert not surface_table.has_complex_eps(_golden.load_case("aniso_doublet_biaxial").table)
    table = surface_table.pack_table(case.table)
    assert table[1].aniso_class == surface_table.ANISO_GENERAL
    assert abs(table[1].eps_im[0] - case.table[1]["material"]["eps_im"][0][0]) == 0.0 and table[1].eps_im[0] != 0.0
    # an isotropic medium behind the LAST surface is fine (its complex k is unique); before it, it is not
    ok = copy.deepcopy(case.table)
    ok[-1]["material"] = {"type": "isotropic", "n": 1.5}
    surface_table.pack_table(ok)
    ok[-1]["material"] = {"type": "isotropic", "n": 3.9, "n_im": 0.02}
    assert surface_table.pack_table(ok)[len(ok) - 1].eps_im[0] == 0.02
    bad = copy.deepcopy(case.table)
    bad[2]["material"] = {"type": "isotropic", "n": 1.5}
    with pytest.raises(surface_table.UnsupportedError, match="behind the last surface only"):
        surface_table.pack_table(bad)
    det = _golden.load_case("absorbing_detector").table
    assert surface_table.has_complex_eps(det) and det[-1]["material"]["n_im"] == 0.02
    bad = copy.deepcopy(det)
    (bad[1], bad[-1]) = (bad[-1], bad[1])                       # the absorbing glass in the middle
    with pytest.raises(surface_table.UnsupportedError, match="absorbing isotropic medium"):
        surface_table.pack_table(bad)


def test_record_memo_follows_every_mutation_path_of_the_mirror_classes():
    """surface_table.surface_record_cached hands back the record of the last walk while the epochs of the objects it
    was read from stand still (raytracer/variables.py).  Every way this package's classes can be changed must move an
    epoch: after each mutation the memoised walk equals a walk with an empty memo; untouched surfaces keep their
    record OBJECTS (that is what makes the unchanged case cheap); in-place edits of held arrays raise."""
    import random
    from pyrate_amd import surface_table, systems
    from pyrate_amd.builders import build_simple_optical_system
    from pyrate_amd.raytracer.aperture import CircularAperture
    from pyrate_amd.raytracer.surface_shape import Conic
    wave = 0.5876e-3
    (s, seq) = build_simple_optical_system(systems.doublet_builduplist() + [
        ({"shape": "Asphere", "curv": -0.01, "cc": -1.0, "coefficients": [1e-4, 1e-7]}, {"decz": 3.0, "tiltx": 0.01},
         {"eps": systems.uniaxial_eps(1.6, 1.5, (0, 0, 1))}, "extra", {})])
    elem = s.elements["stdelem"]

    def fresh():
        surface_table._RECORD_MEMO.clear()
        return surface_table.flatten_sequence(s, seq, wave)[0]

    def memoised():
        return surface_table.flatten_sequence(s, seq, wave)[0]
    base = memoised()
    again = memoised()
    assert all(a is b for (a, b) in zip(base, again))                       # nothing changed: the same objects
    front = elem.surfaces["front"]
    extra = elem.surfaces["extra"]
    mutations = [
        lambda: front.shape.curvature.set_value(front.shape.curvature() * 1.01),
        lambda: extra.shape.params["A4"].set_value(2e-7),
        lambda: extra.shape.annotations.__setitem__("newton_maxit", 17),
        lambda: (front.shape.lc.decz.set_value(front.shape.lc.decz() + 0.1), s.rootcoordinatesystem.update()),
        lambda: (extra.shape.lc.tiltx.set_value(0.02), s.rootcoordinatesystem.update()),
        lambda: front.aperture.annotations.__setitem__("maxradius", 11.0),
        lambda: setattr(elem.surfaces["cement"], "aperture", CircularAperture.p(elem.surfaces["cement"].shape.lc, maxradius=9.0)),
        lambda: setattr(elem.surfaces["rear"], "shape", Conic.p(elem.surfaces["rear"].shape.lc, curv=-0.004)),
        lambda: elem.materials[elem.annotations["surf_mat_connection"]["front"][1]].n.set_value(1.6),
        lambda: setattr(elem.materials[elem.annotations["surf_mat_connection"]["extra"][1]], "epstensor",
                        systems.uniaxial_eps(1.7, 1.5, (0, 1, 0))),
    ]
    rng = random.Random(3)
    for trial in range(40):
        before = memoised()
        which = rng.randrange(len(mutations))
        mutations[which]()
        after = memoised()
        truth = fresh()
        assert after == truth, (trial, which)
        memoised()
        changed = [i for (i, (a, b)) in enumerate(zip(before, truth)) if a != b]
        kept = [i for (i, (a, b)) in enumerate(zip(before, after)) if a is b]
        assert not (set(kept) & set(changed)), (trial, which)
        if which in (0, 1, 2, 5, 8):                  # one object touched, no frame update: every other record is kept
            assert len(kept) >= len(truth) - 2, (trial, which, kept)
    with pytest.raises(ValueError):
        front.shape.lc.globalcoordinates[2] = 5.0               # held arrays are read-only: no silent stale table
    with pytest.raises(ValueError):
        elem.materials[elem.annotations["surf_mat_connection"]["extra"][1]].epstensor[0, 0] = 3.0


def test_whole_walk_is_reused_only_while_nothing_can_have_changed():
    """OpticalSystem._flattened hands back the records of the last call only if the mutation epoch stands still, the
    wavelength is the same and the sequence equals, entry by entry, the one of that call (the caller's list may be
    edited in place); a system that holds an object WITHOUT mutation epochs (a duck-typed stand-in, a class of the
    reference) is walked every time"""
    from pyrate_amd import systems
    from pyrate_amd.builders import build_simple_optical_system
    wave = 0.5876e-3
    (s, seq) = build_simple_optical_system(systems.doublet_builduplist())
    (r1, _) = s._flattened(seq, wave)
    (r2, _) = s._flattened(seq, wave)
    assert r2 is r1                                                 # the same list: no walk
    assert s._flattened([(seq[0][0], list(seq[0][1]))], wave)[0] is r1   # an equal sequence in another list
    seq[0][1][2] = (seq[0][1][2][0], {"is_mirror": True})          # the caller edits his sequence in place
    (r3, _) = s._flattened(seq, wave)
    assert r3 is not r1 and r3[2]["interaction"] == "mirror" and r1[2]["interaction"] == "refract"
    assert s._flattened(seq, 0.4861e-3)[0] is not r3                # another wavelength
    (r4, _) = s._flattened(seq, wave)
    s.elements["stdelem"].surfaces["front"].shape.curvature.set_value(0.02)
    (r5, _) = s._flattened(seq, wave)
    assert r5 is not r4 and r5[1]["shape"]["curv"] == 0.02 and r5[0] is r4[0]   # one surface re-read

    class ForeignShape(object):                                     # no epochs: its state can change unseen
        kind = "shape_Conic"

        def __init__(self, lc):
            self.lc = lc
            self.c = 0.01

        def curvature(self):
            return self.c

        def conic(self):
            return 0.0
    front = s.elements["stdelem"].surfaces["front"]
    foreign = ForeignShape(front.shape.lc)
    front.shape = foreign
    (a, _) = s._flattened(seq, wave)
    foreign.c = 0.03                                                # ... and does
    (b, _) = s._flattened(seq, wave)
    assert a[1]["shape"]["curv"] == 0.01 and b[1]["shape"]["curv"] == 0.03


def test_deferred_raypath_is_a_plain_list_once_somebody_looks():
    """RayPath._deferred: the bundle list of a traced path is built on first access and is an ordinary list from
    then on (append / += / assignment / containsSplitted behave like the reference's attribute, ray.py:207-260)"""
    from pyrate_amd.raytracer.ray import RayPath

    class B(object):
        def __init__(self, tag, splitted=False):
            (self.tag, self.splitted) = (tag, splitted)

        def clone(self):
            return B(self.tag, self.splitted)
    calls = []

    def make():
        calls.append(1)
        return [B(0), B(1, True)]
    p = RayPath._deferred(make)
    p.dense = "dense arrays"
    assert calls == []                                   # nothing built by the trace itself
    assert [b.tag for b in p.raybundles] == [0, 1] and type(p.raybundles) is list
    assert p.raybundles is p.raybundles and calls == [1]
    p.appendRayBundle(B(2))
    q = RayPath(B(9))
    q.appendRayPath(p)
    assert [b.tag for b in q.raybundles] == [9, 0, 1, 2] and q.containsSplitted()
    c = RayPath._deferred(make).clone()                  # cloning looks, too
    assert [b.tag for b in c.raybundles] == [0, 1] and calls == [1, 1]
    r = RayPath._deferred(make)
    r.raybundles = []                                    # plain assignment wins over the pending builder
    assert r.raybundles == [] and calls == [1, 1]


def test_polynomial_rotation_and_shift_algebra():
    """polyshape.rotated / shifted: p(R^T (x - d)) evaluated term by term == the polynomial evaluated at the transformed
    point, for random polynomials, angles and offsets; a rotation followed by its inverse is the identity"""
    from pyrate_amd import polyshape
    rng = np.random.RandomState(5)
    (x, y) = (rng.uniform(-3, 3, 40), rng.uniform(-3, 3, 40))

    def evaluate(terms, u, v):
        return sum(c * u ** i * v ** j for ((i, j), c) in terms.items())
    for trial in range(25):
        terms = {(int(i), int(j)): float(rng.uniform(-1, 1)) for (i, j) in rng.randint(0, 6, size=(int(rng.randint(1, 9)), 2))}
        a = float(rng.uniform(-np.pi, np.pi))
        rot = ((np.cos(a), -np.sin(a)), (np.sin(a), np.cos(a)))
        (dx, dy) = (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
        placed = polyshape.shifted(polyshape.rotated(terms, rot), dx, dy)
        (u, v) = (x - dx, y - dy)
        (xs, ys) = (rot[0][0] * u + rot[1][0] * v, rot[0][1] * u + rot[1][1] * v)            # rot^T (u, v)
        want = evaluate(terms, xs, ys)
        assert np.allclose(evaluate(placed, x, y), want, rtol=1e-11, atol=1e-11 * np.abs(want).max())
        assert max(i + j for (i, j) in placed) <= max(i + j for (i, j) in terms)              # the degree stays
        back = polyshape.rotated(polyshape.rotated(terms, rot), ((rot[0][0], rot[1][0]), (rot[0][1], rot[1][1])))
        assert np.allclose(evaluate(back, x, y), evaluate(terms, x, y), rtol=1e-10, atol=1e-10)
    assert polyshape.rotated({(2, 1): 1.0}, ((1.0, 0.0), (0.0, 1.0))) == {(2, 1): 1.0}


def test_a_variable_shared_by_two_objects_invalidates_both():
    """a FloatVariable assigned to two objects (a pickup: two surfaces sharing one curvature) used to
    advance only its LAST owner's epoch -- the other holder's cached record went stale"""
    from pyrate_amd import surface_table, systems
    from pyrate_amd.builders import build_simple_optical_system
    wave = 0.5876e-3
    (s, seq) = build_simple_optical_system(systems.doublet_builduplist())
    elem = s.elements["stdelem"]
    (sa, sb) = (elem.surfaces["front"].shape, elem.surfaces["rear"].shape)
    sb.curvature = sa.curvature                          # one variable, two holders
    surface_table.flatten_sequence(s, seq, wave)
    sa.curvature.set_value(0.05)
    recs = surface_table.flatten_sequence(s, seq, wave)[0]
    surface_table._RECORD_MEMO.clear()
    truth = surface_table.flatten_sequence(s, seq, wave)[0]
    assert recs == truth
    curvs = [r["shape"]["curv"] for r in recs if r["shape"].get("curv") == 0.05]
    assert len(curvs) == 2
    # a holder that is gone is dropped from the owner list, not kept alive by it
    import gc
    from pyrate_amd.raytracer.variables import FloatVariable, Named
    v = FloatVariable(1.0)
    (a, b) = (Named("a"), Named("b"))
    (a.v, b.v) = (v, v)
    assert len(v._owners) == 2
    del b
    gc.collect()
    e0 = a._epoch
    v.set_value(2.0)
    assert a._epoch > e0 and v._owner is a and len([r for r in v._owners if r() is not None]) == 1


def test_deep_copies_and_pickles_keep_their_mutation_tracking():
    """copy.deepcopy / pickle restore an object's attributes without __setattr__; the copy's
    dictionaries and variables must move the COPY's epoch (and not the original's)"""
    import copy
    import pickle
    from pyrate_amd import surface_table, systems
    from pyrate_amd.builders import build_simple_optical_system
    from pyrate_amd.raytracer.variables import TrackedDict
    wave = 0.5876e-3
    (s, seq) = build_simple_optical_system(systems.doublet_builduplist())
    for clone in (copy.deepcopy, lambda o: pickle.loads(pickle.dumps(o))):
        s2 = clone(s)
        el = s2.elements["stdelem"]
        shape = el.surfaces["front"].shape
        assert isinstance(shape.annotations, TrackedDict) and shape.annotations._owner is shape
        assert isinstance(el.surfaces, TrackedDict) and el.surfaces._owner is el
        before = surface_table.flatten_sequence(s2, seq, wave)[0]
        (e_copy, e_orig) = (shape._epoch, s.elements["stdelem"].surfaces["front"].shape._epoch)
        shape.annotations["newton_maxit"] = 11
        assert shape._epoch > e_copy
        e_copy = shape._epoch
        shape.curvature.set_value(0.02)
        assert shape._epoch > e_copy and s.elements["stdelem"].surfaces["front"].shape._epoch == e_orig
        after = surface_table.flatten_sequence(s2, seq, wave)[0]
        surface_table._RECORD_MEMO.clear()
        assert after == surface_table.flatten_sequence(s2, seq, wave)[0] and after != before
        # the original is untouched by all of this
        assert s.elements["stdelem"].surfaces["front"].shape.curvature() != 0.02


def test_frame_tree_update_moves_only_the_frames_that_moved():
    """LocalCoordinates.update() re-assigned its matrices on every frame, so one update() of the root
    per optimiser step invalidated every cached record; now only frames whose numbers changed advance their epoch"""
    from pyrate_amd import surface_table, systems
    from pyrate_amd.builders import build_simple_optical_system
    wave = 0.5876e-3
    (s, seq) = build_simple_optical_system(systems.doublet_builduplist())
    base = surface_table.flatten_sequence(s, seq, wave)[0]
    s.rootcoordinatesystem.update()
    again = surface_table.flatten_sequence(s, seq, wave)[0]
    assert all(a is b for (a, b) in zip(base, again))               # nothing moved: every record object is kept
    rear = s.elements["stdelem"].surfaces["rear"]
    rear.shape.lc.decz.set_value(rear.shape.lc.decz() + 0.25)
    s.rootcoordinatesystem.update()
    moved = surface_table.flatten_sequence(s, seq, wave)[0]
    surface_table._RECORD_MEMO.clear()
    assert moved == surface_table.flatten_sequence(s, seq, wave)[0]
    same = [a is b for (a, b) in zip(base, moved)]
    assert not all(same) and same[0] and same[1]                   # the frames in front of the moved one kept theirs


def test_bench_attributes_march_launches_to_their_configurations():
    """bench.py's live PMC pass (benchmarks/pmc.py): a counter row belongs to a configuration by its place in the
    dispatch order of the kernels that count -- the marches (one launch per trace) and, for the per-surface paths
    (plugin, aniso_chain), the k_propagate / k_interact launches, a known number per trace; the fall-back (a counter file
    without dispatch ids) reads the march's instantiation from the kernel name"""
    import sys
    sys.path.insert(0, ROOT)
    from benchmarks import pmc, workloads
    f = pmc.config_of_march
    assert f("void k_trace_iso<0, true, true, 0, false, true, false>(prt_dev_surface const*)") == "doublegauss"
    assert f("void k_trace_iso<0, true, true, 0, false, false, false>(...)") == "benchmark"
    assert f("void k_trace_iso<0, true, true, 1, false, true, false>(...)") == "asphere"
    assert f("void k_trace_iso<0, true, true, 2, false, true, false>(...)") == "xypoly"
    assert f("void k_trace_general<0, false, true, false, 0>(...)") == "aniso"
    assert f("void k_trace_general<0, true, true, false, 0>(...)") == "aniso_biaxial"
    assert f("void k_propagate(...)") is None
    L = pmc.PMC_LAUNCHES
    rows = [{"Kernel_Name": "void k_trace_iso<0, true, true, 0, false, true, false>()", "Dispatch_Id": str(10 + i),
             "Counter_Name": "WRITE_SIZE", "Counter_Value": "1"} for i in range(L)] + \
           [{"Kernel_Name": "void k_trace_general<0, true, true, false, 0>()", "Dispatch_Id": str(40 + i),
             "Counter_Name": "WRITE_SIZE", "Counter_Value": "2"} for i in range(L)] + \
           [{"Kernel_Name": "void k_rectgrid_mask()", "Dispatch_Id": "5", "Counter_Name": "WRITE_SIZE", "Counter_Value": "9"}]
    got = pmc.rows_by_config(rows, ["doublegauss", "aniso_biaxial"])
    assert sorted(set(c for (c, _, _) in got)) == ["aniso_biaxial", "doublegauss"] and len(got) == 2 * L
    assert all(v == (1.0 if c == "doublegauss" else 2.0) for (c, _, v) in got)
    # a per-surface configuration between two marches: 2 surfaces -> 4 kernels per trace
    sweep = []
    for t in range(L):
        for q in range(2):
            sweep.append({"Kernel_Name": "void k_propagate_rows<true>()", "Dispatch_Id": str(100 + 4 * t + 2 * q),
                          "Counter_Name": "WRITE_SIZE", "Counter_Value": "3"})
            sweep.append({"Kernel_Name": "void k_interact_iso_rows<true>()", "Dispatch_Id": str(101 + 4 * t + 2 * q),
                          "Counter_Name": "WRITE_SIZE", "Counter_Value": "4"})
    tail = [{"Kernel_Name": "void k_trace_general<0, false, true, false, 0>()", "Dispatch_Id": str(500 + i),
             "Counter_Name": "WRITE_SIZE", "Counter_Value": "5"} for i in range(L)]
    got = pmc.rows_by_config(rows[:L] + sweep + tail, ["doublegauss", "plugin", "aniso"], {"plugin": 4})
    by = {}
    for (c, _, v) in got:
        by.setdefault(c, []).append(v)
    assert sorted(by) == ["aniso", "doublegauss", "plugin"]
    assert len(by["plugin"]) == 4 * L and sum(by["plugin"]) == (3 + 4) * 2 * L and set(by["aniso"]) == {5.0}
    # a count that does not add up (another kernel of these names slipped in): only the marches, by their names
    got = pmc.rows_by_config(rows[:L] + sweep[:-1] + tail, ["doublegauss", "plugin", "aniso"], {"plugin": 4})
    assert sorted(set(c for (c, _, _) in got)) == ["aniso", "doublegauss"]
    assert set(workloads.SECONDARY_MARCH_CONFIGS) | set(workloads.SECONDARY_CUSTOM_CONFIGS) == \
        {"aniso_biaxial", "aniso_chain", "plugin", "image_moments"}


def test_bench_condenses_a_full_run_into_a_line_the_driver_can_keep():
    """Round 5's bench line had grown to 25 KB and the driver could not parse it; nothing in the tree could have caught
    that.  ``bench.compact_single`` + ``bench.emit`` on the FULL records of a real default run (profiles/
    r06b_bench_detail.json: nine configurations, scaling point, end to end): the line is one JSON object of < 8 KB with
    the contract's keys, the roofline is reproducible from its own parts, every configuration appears in the summary --
    and a record that does outgrow the limit is cut down (and says so) instead of being printed."""
    import importlib.util
    import json
    import sys
    spec = importlib.util.spec_from_file_location("bench_for_compaction", os.path.join(ROOT, "bench.py"))
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    with open(os.path.join(ROOT, "profiles", "r06b_bench_detail.json")) as f:
        detail = json.load(f)
    base = {"metric": "ray_surface_ops_per_s", "unit": "ray-surface-ops/s", "n_gpus": 1, "steps": 200, "warmup": 20,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    recs = detail["configs"]
    line = bench.compact_single(base, recs[0], recs, detail["scaling_point"], detail["e2e"], detail["arena"], detail["build"],
                                detail["wall_s_by_stage"])
    (r, w) = os.pipe()
    bench.emit(w, line)
    os.close(w)
    text = os.read(r, 1 << 16).decode()
    os.close(r)
    assert text.endswith("\n") and text.count("\n") == 1 and len(text) < 8192 and len(text) < 4096, len(text)
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "scaling_point", "e2e"):
        assert key in d, key
    rf = d["roofline"]
    assert abs(rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9 / rf["peak"] - rf["frac"]) < 1e-4
    assert rf["traffic"] is not None and abs(rf["traffic_ratio_to_algorithmic"] - 1.0) < 0.01 and rf["ops_vs_98B_convention"] > 1
    assert set(d["config"]["configs_summary"]) == {"doublegauss", "asphere", "aniso", "xypoly", "benchmark", "aniso_biaxial",
                                                   "aniso_chain", "plugin", "image_moments", "scaling_point_1e8_rays"}
    assert d["config"]["configs_summary"]["plugin"][3] is not None          # measured traffic of the per-surface path
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["verified"]["ok"]
    # the guarded tenth record of the default run (the fused surface step, same shape as the plugin record)
    import copy
    step = copy.deepcopy([c for c in recs if c["name"] == "plugin"][0])
    step["name"] = "surface_step"
    step["roofline"]["kernel"] = "k_surface_step_rows"
    i_plugin = [c["name"] for c in recs].index("plugin")
    recs10 = recs[:i_plugin + 1] + [step] + recs[i_plugin + 1:]
    line10 = bench.compact_single(base, recs10[0], recs10, detail["scaling_point"], detail["e2e"], detail["arena"], detail["build"],
                                  detail["wall_s_by_stage"])
    (r, w) = os.pipe()
    bench.emit(w, line10)
    os.close(w)
    text10 = os.read(r, 1 << 16).decode()
    os.close(r)
    d10 = json.loads(text10)
    assert len(text10) < 4096 and "surface_step" in d10["config"]["configs_summary"] and "truncated" not in d10
    assert d10["roofline"] == d["roofline"] and d10["value"] == d["value"]
    # a record that outgrows the limit is cut down, not printed
    fat = dict(line, config=dict(line["config"], configs_summary={("cfg%d" % i): [1.0, 0.5, True, 1.0, "x" * 90] for i in range(90)}))
    (r, w) = os.pipe()
    bench.emit(w, fat)
    os.close(w)
    text = os.read(r, 1 << 16).decode()
    os.close(r)
    cut = json.loads(text)
    assert len(text) < 8192 and "truncated" in cut and cut["roofline"] == line["roofline"] and cut["value"] == line["value"]


def test_row_pool_carves_arrays_out_of_buffers_of_a_kind_its_caller_does_not_read(monkeypatch):
    """placed.RowPool (round 6: the arrays of DeviceSystem.propagate / interact on big bundles): pieces are carved front
    to back out of 1-GiB arena buffers, never twice; a request names the kinds of HBM to stay out of (the kinds of the
    arrays the same kernel READS) and gets a piece of another kind -- from a buffer at hand if one has room, else a new
    buffer asked for with exactly that avoid mask.  (A fake arena stands in for the device.)"""
    import torch
    from pyrate_amd import placed

    class MetaArena(object):          # (a meta tensor has the sizes of a 1-GiB buffer without the memory)
        def __init__(self):
            self.calls = []

        def alloc(self, sizes, n_distinct=2, avoid_mask=0, max_hunt_slabs=-1):
            self.calls.append((list(sizes), n_distinct, avoid_mask))
            kind = [q for q in range(3) if not (avoid_mask >> q) & 1][0]
            return [torch.empty(int(s), dtype=torch.uint8, device="meta") for s in sizes], [kind]
    arena = MetaArena()
    monkeypatch.setattr(placed.PlacedArena, "for_device", classmethod(lambda cls, index: arena))
    pool = placed.RowPool()
    dev = torch.device("cuda", 0)            # (only its index is looked at)
    piece = 250 << 20
    (a, ka) = pool.take(dev, piece, avoid_kinds=(0, 1))
    assert ka == 2 and arena.calls == [([1 << 30], 1, 0b011)] and a.numel() == piece
    (b, kb) = pool.take(dev, piece, avoid_kinds=(0, None))          # kind 2's buffer has room: no new buffer
    assert kb == 2 and len(arena.calls) == 1
    assert b.storage_offset() == a.storage_offset() + piece and b.storage_offset() % 4096 == 0      # front to back
    (c, kc) = pool.take(dev, piece, avoid_kinds=(2, 1))
    assert kc == 0 and arena.calls[-1] == ([1 << 30], 1, 0b110)
    for _ in range(2):
        pool.take(dev, piece, avoid_kinds=(0, 1))                      # kind 2's buffer: 4 pieces of 250 MiB fit
    assert len(arena.calls) == 2
    (d, kd) = pool.take(dev, piece, avoid_kinds=(0, 1))                # the fifth does not: a new buffer of that kind
    assert kd == 2 and len(arena.calls) == 3 and d.storage_offset() == 0
    (e, ke) = pool.take(dev, 3 << 30, avoid_kinds=())                  # bigger than a slab: a buffer of its own size
    assert arena.calls[-1][0] == [3 << 30] and e.numel() == 3 << 30
    (f, kf) = pool.take(dev, 1000, avoid_kinds=(ke,))
    assert kf != ke and f.numel() == 4096                              # pieces are whole 4-KiB pages
    pool.clear()
    pool.take(dev, piece, avoid_kinds=())
    assert arena.calls[-1][0] == [1 << 30]
