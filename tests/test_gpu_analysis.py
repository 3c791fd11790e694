"""GPU tests of the analysis layer: device reductions behind
RayBundleAnalysis against the reference's own known answers
(reference tests/test_ray_analysis.py:34-112) and against NumPy on traced bundles;
OpticalSystemAnalysis / raytrace() convenience."""
import math

import numpy as np
import pytest

import _golden
import systems_zoo as zoo
from pyrate_amd import systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rb5(gpu_device):
    from pyrate_amd.raytracer.ray import RayBundle
    k0 = np.zeros((3, 5))
    k0[2, :] = 1
    e0 = np.zeros((3, 5))
    e0[1, :] = 1.
    return RayBundle(x0=np.array([[1., 0, 0, 1, 2], [0, 1., 0, 1, 2], [0, 0, 1., 1, 2]]), k0=k0, Efield0=e0)


def test_reference_known_answers(rb5):
    from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
    ra = RayBundleAnalysis(rb5)
    assert np.allclose(ra.get_centroid_position(), 4. / 5.)
    assert np.isclose(ra.get_rms_spot_size(np.array([0, 0, 0])), math.sqrt(18.0 / 4.0))
    assert np.allclose(ra.get_centroid_direction(), np.array([0, 0, 1]))
    ang = ra.get_rms_angluar_size(np.array([math.sin(math.pi / 180.0), 0, math.cos(math.pi / 180.0)]))
    assert np.isclose(ang, math.pi / 180.0)


def test_arc_length_known_answer(gpu_device):
    from pyrate_amd.raytracer.ray import RayBundle
    from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
    k0 = np.zeros((3, 2))
    e0 = np.zeros((3, 2))
    rb = RayBundle(x0=np.zeros((3, 2)), k0=k0, Efield0=e0)
    valid = np.ones(2, dtype=bool)
    for x in ([[1, 0], [0, 0], [0, 0]], [[1, 1], [1, 1], [0, 0]], [[0, 2], [1, 2], [0, 0]], [[0, 3], [0, 3], [0, 0]]):
        rb.append(np.array(x, dtype=float), k0, e0, valid)
    assert np.allclose(RayBundleAnalysis(rb).get_arc_length(), np.array([4., 3 * np.sqrt(2)]))


def test_spot_statistics_of_traced_bundle_match_numpy(gpu_device):
    """image bundle of the double Gauss (5 deg field): device moments == NumPy, and the
    on-axis RMS radius about the origin reproduces the survey's known answer (SURVEY 8c:
    9917 rays, rpup 5, z0 -10: 0.4450661837097206 mm)."""
    from pyrate_amd.builders import build_rotationally_symmetric_optical_system
    from pyrate_amd.raytracer.ray import RayBundle
    from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
    (s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    (o, k, e0) = systems.double_gauss_bundle(10000)
    assert o.shape[1] == 9917
    img = s.seqtrace(RayBundle(o, k, e0, wave=systems.DLINE), seq)[0].raybundles[-1]
    x = img.x[-1]
    rms_origin = math.sqrt(np.sum(x[0] ** 2 + x[1] ** 2) / x.shape[1])
    assert abs(rms_origin - 0.4450661837097206) < 1e-12
    (o, k, e0) = systems.double_gauss_bundle(200000, field_deg=5.0)
    img = s.seqtrace(RayBundle(o, k, e0, wave=systems.DLINE), seq)[0].raybundles[-1]
    ra = RayBundleAnalysis(img)
    x = img.x[-1]
    cen = np.sum(x, axis=1) / (x.shape[1] + 1e-17)
    assert np.allclose(ra.get_centroid_position(), cen, rtol=1e-13, atol=1e-13)
    rms = math.sqrt(np.sum((x - cen[:, None]) ** 2) / (x.shape[1] - 1 + 1e-17))
    assert abs(ra.get_rms_spot_size_centroid() - rms) < 1e-12 * max(1.0, rms)
    d = np.real(img.k[-1])
    d = d / np.sqrt(np.sum(d ** 2, axis=0))
    com = np.sum(d, axis=1)
    assert np.allclose(ra.get_centroid_direction(), com / np.linalg.norm(com), rtol=0, atol=1e-13)
    ref = com / np.linalg.norm(com)
    ang = math.asin(math.sqrt(np.sum(np.cross(d, ref, axisa=0).T ** 2) / d.shape[1]))
    assert abs(ra.get_rms_angluar_size_centroid() - ang) < 1e-12


def test_masked_moments_on_dense_engine_output(gpu_device):
    """engine level: moments over valid_out of the dense, row-pitched image-plane arrays"""
    from pyrate_amd import engine
    case = _golden.load_case("double_gauss_wide")
    sysd = engine.DeviceSystem(case.table, 0)
    res = sysd.trace(engine.to_device_rays(case.x0, gpu_device), engine.to_device_rays(case.k0, gpu_device),
                     engine.to_device_rays(case.E0, gpu_device))
    (cnt, s1, s2) = engine.bundle_moments(res.x_hit[-1], mask=res.valid_out[-1])
    m = res.valid_out[-1].cpu().numpy().astype(bool)
    x = res.x_hit[-1].cpu().numpy()[:, m]
    assert cnt == m.sum() == 277
    assert np.allclose(s1, x.sum(axis=1), rtol=1e-13) and np.allclose(s2, (x ** 2).sum(axis=1), rtol=1e-13)


def test_raytrace_convenience_readme_example(gpu_device):
    """README singlet (README.md:200-210): raytrace(s, seq, 11, {"radius": 9.0}) -> 12 rays,
    5 bundles; ray 0 image point / k from SURVEY 8c"""
    from pyrate_amd.builders import build_rotationally_symmetric_optical_system, raytrace
    (s, seq) = build_rotationally_symmetric_optical_system(
        [(100., 0, 20., 1.5, "front", {}), (-100., 0, 5., None, "back", {}), (0, 0, 100., None, "image", {})])
    r = raytrace(s, seq, 11, {"radius": 9.0})          # [bundle][ray path], like the reference
    assert len(r) == 1 and len(r[0]) == 1 and len(r[0][0].raybundles) == 5
    img = r[0][0].raybundles[-1]
    assert img.x.shape == (1, 3, 12)
    assert np.allclose(img.x[-1][:, 0], [6.0134358655051567e-02, 1.8040307596514893e-01, 125.0], rtol=0, atol=1e-12)
    assert np.allclose(np.real(img.k[-1][:, 0]), [0.02810922264455102, 0.08432766793365307, 0.9960415232424753],
                       rtol=0, atol=1e-13)


def test_optical_system_analysis_bundles(gpu_device):
    from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
    from pyrate_amd.sampling2d import raster
    api = zoo.mirror_api()
    (s, seq) = zoo.doublet(api)
    osa = OpticalSystemAnalysis(s, seq)
    ref = _golden.load_case("doublet_clipped")       # reference collimated_bundle(300, radius 14.5, anglex 0.03)
    (o, k, e) = osa.collimated_bundle(300, {"startz": -5., "radius": 14.5, "anglex": 0.03}, wave=zoo.DLINE)
    assert np.array_equal(o, ref.x0)
    assert np.allclose(k, np.real(ref.k0), rtol=0, atol=1e-15)
    assert np.allclose(np.sum(e * k, axis=0), 0, atol=1e-15) and np.allclose(np.sum(e * e, axis=0), 1)
    (o, k, e) = osa.divergent_bundle(50, {"radius": 0.1, "raster": raster.MeridionalFan()})
    assert o.shape == (3, 50) and np.allclose(np.sum(k * k, axis=0), 1.0)
    osa.aim(20, {"startz": -5., "radius": 11.43, "raster": raster.MeridionalFan()}, wave=zoo.DLINE)
    r2 = osa.trace()[0]
    assert len(r2[0].raybundles) == 7 and r2[0].raybundles[-1].x.shape == (1, 3, 20)


@pytest.mark.parametrize("nray", [11, 100, 10000, 123457])
def test_device_bundle_generation_is_bit_identical_to_the_numpy_raster(gpu_device, nray):
    """RectGrid + collimated bundle generated on the GPU == the host raster
    (identical pupil samples), whole bundle and arbitrary shards"""
    import torch
    from pyrate_amd import engine
    for field in (0.0, 5.0):
        (o, k, e0) = systems.double_gauss_bundle(nray, field_deg=field)
        (xd, kd, ed, total) = systems.double_gauss_bundle_device(nray, gpu_device, field_deg=field)
        assert total == o.shape[1] == engine.rect_grid_count(nray, gpu_device)[1]
        assert np.array_equal(xd.cpu().numpy(), o)
        assert np.array_equal(kd.cpu().numpy(), k)
        assert np.array_equal(ed.cpu().numpy(), e0)
    (lo, hi) = (total // 3, total // 3 + max(1, total // 2))
    (xs, ks, es, _) = systems.double_gauss_bundle_device(nray, gpu_device, field_deg=5.0, lo=lo, hi=hi)
    assert np.array_equal(xs.cpu().numpy(), o[:, lo:hi]) and np.array_equal(ks.cpu().numpy(), k[:, lo:hi])
    (px, py) = systems.rect_grid(nray)
    assert engine.rect_grid_count(nray, gpu_device) == (int(round(math.sqrt(nray * 4.0 / math.pi))), px.shape[0])


def test_return_k_to_d_and_phase_difference(gpu_device):
    """RayBundle.returnKtoD (complex E, ray.py:136-152) and get_phase_difference
    (ray_analysis.py:149-163) on the device == the reference formulas in NumPy"""
    from pyrate_amd.raytracer.ray import RayBundle
    from pyrate_amd.raytracer.analysis.ray_analysis import RayBundleAnalysis
    rng = np.random.RandomState(11)
    n = 1000
    x0 = rng.rand(3, n)
    k0 = rng.rand(3, n) + 0.2
    e0 = rng.rand(3, n) + 1j * rng.rand(3, n)
    rb = RayBundle(x0, k0, e0)
    x1 = x0 + rng.rand(3, n)
    rb.append(x1, k0, e0, np.ones(n, dtype=bool))
    d = rb.returnKtoD()
    assert d.shape == (2, 3, n)
    absE2 = np.sum(np.conj(e0) * e0, axis=0)
    Ek = np.sum(e0 * k0, axis=0)
    S = np.real(absE2 * k0 - Ek * np.conj(e0))
    dref = S / np.sqrt(np.sum(S ** 2, axis=0))
    assert np.allclose(d[0], dref, rtol=0, atol=1e-14) and np.allclose(d[1], dref, rtol=0, atol=1e-14)
    ra = RayBundleAnalysis(rb)
    assert np.allclose(ra.get_arc_length(), np.sqrt(np.sum((x1 - x0) ** 2, axis=0)), rtol=1e-14)
    assert np.allclose(ra.get_phase_difference(), np.sum(x1 * k0 - x0 * k0, axis=0), rtol=1e-12, atol=1e-14)
    # default E (ey): the reference's quirk d = (kx, 0, kz)/norm
    rb2 = RayBundle(x0, k0, None)
    d2 = rb2.returnKtoD()[0]
    ref2 = np.vstack((k0[0], np.zeros(n), k0[2]))
    assert np.allclose(d2, ref2 / np.sqrt(np.sum(ref2 ** 2, axis=0)), rtol=0, atol=1e-14)


def test_async_spot_statistics_and_gather_single_rank(gpu_device):
    """the device-resident statistics pipeline and the image-plane gather of the multi-GPU bench
    path, on one rank (no process group): same numbers as the synchronous reductions"""
    from pyrate_amd import engine
    from pyrate_amd import distributed as pdist
    case = _golden.load_case("double_gauss_wide")
    sysd = engine.DeviceSystem(case.table, 0)
    res = sysd.trace(engine.to_device_rays(case.x0, gpu_device), engine.to_device_rays(case.k0, gpu_device),
                     engine.to_device_rays(case.E0, gpu_device))
    (x, k, v) = (res.x_hit[-1], res.k_out[-1], res.valid_out[-1])
    st = pdist.SpotStatistics(gpu_device)
    st.start(x, v)
    (cnt, cen, rms) = st.result()
    (cnt2, cen2, rms2) = pdist.global_spot_statistics(x, v)
    assert cnt == cnt2 == 277
    assert np.allclose(cen, cen2, rtol=0, atol=1e-14) and abs(rms - rms2) < 1e-14
    m = v.cpu().numpy().astype(bool)
    xs = x.cpu().numpy()[:, m]
    c = xs.mean(axis=1)
    assert np.allclose(cen, c, rtol=1e-13) and abs(rms - np.sqrt(((xs - c[:, None]) ** 2).sum() / (m.sum() - 1))) < 1e-13
    g = pdist.ImagePlaneGather(x.shape[1], gpu_device)
    g.start(x, k, v)
    (gx, gk, gv) = g.finish()
    import torch
    assert torch.equal(torch.nan_to_num(gx), torch.nan_to_num(x)) and torch.equal(gv, v)
    assert torch.equal(torch.nan_to_num(gk), torch.nan_to_num(k))
    assert pdist.shard_range(x.shape[1], 0, 1) == (0, x.shape[1])


def test_trace_moments_equals_two_pass_statistics(gpu_device):
    """prt_trace_moments: moments from inside the trace launch == RayBundleAnalysis-style two-pass
    statistics of the image plane; identical path outputs; bit-reproducible; unaligned fallback"""
    import torch
    from pyrate_amd import engine, systems, _lib
    recs = systems.double_gauss_records()
    sysd = engine.DeviceSystem(recs, 0)
    (x0, k0, e0, _) = systems.double_gauss_bundle_device(200000, gpu_device, field_deg=4.0, rpup=25.0)
    n = x0.shape[1]
    ws = engine.MomentsWorkspace(gpu_device, n_results=2, n_rays=n)
    for mode in (_lib.MODE_PATH, _lib.MODE_IMAGE):
        bufs = sysd.alloc_outputs(n, mode)
        ref = sysd.alloc_outputs(n, mode)
        sysd.trace_into(x0, k0, ref, e0)
        m = sysd.trace_moments_into(x0, k0, bufs, ws, slot=0, e0_re=e0).cpu().numpy().copy()
        for key in ("x_hit", "k_out"):
            vb = sysd.views(bufs)
            vr = sysd.views(ref)
            assert torch.equal(getattr(vb, key)[-1].contiguous().view(torch.int64),
                               getattr(vr, key)[-1].contiguous().view(torch.int64))       # bitwise (NaN rows too)
        assert torch.equal(sysd.views(bufs).valid_out[-1], sysd.views(ref).valid_out[-1])
        v = sysd.views(ref)
        (cnt, cen, rms) = engine.spot_from_moments(m, sysd.moments_reference())
        valid = v.valid_out[-1].cpu().numpy().astype(bool)
        x = v.x_hit[-1].cpu().numpy()[:, valid]
        assert 0 < cnt == valid.sum() < n                   # the wide bundle is vignetted
        c_ref = x.mean(axis=1)
        rms_ref = np.sqrt(np.sum((x - c_ref[:, None]) ** 2) / (x.shape[1] - 1))
        c_ref = np.array([math.fsum(row) for row in x]) / x.shape[1]          # exactly rounded sums
        rms_ref = math.sqrt(math.fsum(((x - c_ref[:, None]) ** 2).ravel()) / (x.shape[1] - 1))
        assert np.allclose(cen, c_ref, rtol=0, atol=1e-10)
        assert abs(rms - rms_ref) < 1e-10 * rms_ref
        m2 = sysd.trace_moments_into(x0, k0, bufs, ws, slot=1, e0_re=e0).cpu().numpy()
        assert np.array_equal(m, m2)                        # fixed summation order
    # odd ray count with tight (unaligned rows) buffers -> trace + two-kernel reduction inside the library
    n_odd = 9999
    lo = n // 2
    xo = x0[:, lo:lo + n_odd].contiguous()
    ko = k0[:, lo:lo + n_odd].contiguous()
    eo = e0[:, lo:lo + n_odd].contiguous()
    bufs = sysd.alloc_outputs(n_odd, _lib.MODE_PATH, pitch=n_odd)
    m = sysd.trace_moments_into(xo, ko, bufs, ws, slot=0, e0_re=eo).cpu().numpy()
    v = sysd.views(bufs)
    valid = v.valid_out[-1].cpu().numpy().astype(bool)
    x = v.x_hit[-1].cpu().numpy()[:, valid]
    (cnt, cen, rms) = engine.spot_from_moments(m, sysd.moments_reference())
    assert cnt == valid.sum() and np.allclose(cen, x.mean(axis=1), rtol=0, atol=1e-10)
    # the same with packed mask flags (what OpticalSystem.image_moments allocates): the fallback selects by
    # bit 1 of the flags byte; and an empty bundle gives zero moments instead of an error
    pb = sysd.alloc_outputs(n_odd, _lib.MODE_IMAGE, pitch=n_odd, packed_flags=True)
    mp = sysd.trace_moments_into(xo, ko, pb, ws, slot=1, e0_re=eo).cpu().numpy()
    assert np.array_equal(mp, m)
    empty = sysd.alloc_outputs(0, _lib.MODE_IMAGE, packed_flags=True)
    z = torch.zeros((3, 0), dtype=torch.float64, device=gpu_device)
    assert np.array_equal(sysd.trace_moments_into(z, z, empty, ws, slot=1).cpu().numpy(), np.zeros(7))
    # crystals: not offered
    sysa = engine.DeviceSystem(systems.aniso_doublet_records(), 0)
    with pytest.raises(_lib.PrtError):
        sysa.trace_moments_into(xo, ko, sysa.alloc_outputs(n_odd), ws)


def test_ray_path_analysis_opd(gpu_device):
    """RayPathAnalysis (ray_analysis.py:169-213) over a traced path == the reference's formulas
    evaluated on the reference's own bundles of the same system"""
    from pyrate_amd.raytracer.analysis.ray_analysis import RayPathAnalysis
    api = zoo.mirror_api()
    case = _golden.load_case("doublet")
    (s, seq) = zoo.doublet(api)
    rp = s.seqtrace(api.RayBundle(x0=case.x0, k0=case.k0, Efield0=case.E0, wave=case.wave), seq)[0]
    arc = np.zeros(case.x0.shape[1])
    ph = np.zeros(case.x0.shape[1])
    for b in case.raw_bundles:
        (x, k) = (b["x"], np.real(b["k"]))
        arc += np.sum(np.sqrt(np.sum((x[1:] - x[:-1]) ** 2, axis=1)), axis=0)
        ph += np.sum(np.sum(x[1:] * k[1:] - x[:-1] * k[:-1], axis=1), axis=0)
    rpa = RayPathAnalysis(rp)
    assert np.allclose(rpa.get_arc_length(), arc, rtol=1e-13)
    assert np.allclose(rpa.get_phase_difference(), ph, rtol=1e-12, atol=1e-12)
    rel = rpa.get_relative_phase_difference(referenceray=3, wavelength=case.wave)
    assert rel[3] == 0.0 and np.allclose(rel, (ph - ph[3]) / case.wave, rtol=1e-9, atol=1e-6)
    # a path that loses rays on the way cannot be summed ray by ray (the reference fails the same way)
    clipped = _golden.load_case("doublet_clipped")
    rp2 = s.seqtrace(api.RayBundle(x0=clipped.x0, k0=clipped.k0, Efield0=clipped.E0, wave=clipped.wave), seq)[0]
    with pytest.raises(ValueError):
        RayPathAnalysis(rp2).get_arc_length()


def test_optical_system_analysis_convenience_traces(gpu_device):
    """trace_3d_global / trace_3d_local / trace_2d_local / get_spot (optical_system_analysis.py:193-303)"""
    from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
    api = zoo.mirror_api()
    case = _golden.load_case("doublet")
    (s, seq) = zoo.doublet(api)
    osa = OpticalSystemAnalysis(s, seq)
    g = osa.trace_3d_global(case.x0, np.real(case.k0), wave=case.wave)
    assert len(g) == 1 and len(g[0]) == 1 and len(g[0][0]) == len(case.raw_bundles)
    for ((X, K), ref) in zip(g[0][0], case.raw_bundles):
        assert np.allclose(X, ref["x"][0], rtol=0, atol=1e-11) and np.allclose(K, np.real(ref["k"][0]), atol=1e-12)
    loc = osa.trace_3d_local(x0=case.x0, k0=np.real(case.k0), wave=case.wave)
    two = osa.trace_2d_local(x0=case.x0, k0=np.real(case.k0), wave=case.wave)
    surfs = osa._flat_surfaces()
    for (j, ((Xl, Kl), (X, K))) in enumerate(zip(loc[0][0], g[0][0])):
        lc = surfs[j].rootcoordinatesystem
        assert np.allclose(Xl, lc.returnGlobalToLocalPoints(X)) and np.allclose(two[0][0][j][0], Xl[:2])
    rp = osa.trace()[0][0]
    (xy, rms) = osa.get_spot(rp)
    img = case.raw_bundles[-1]["x"][-1]
    c = img.mean(axis=1)
    assert xy.shape == (2, img.shape[1])
    assert abs(rms - np.sqrt(np.sum((img - c[:, None]) ** 2) / (img.shape[1] - 1))) < 1e-12


def test_zernike_shape_evaluation_on_the_device(gpu_device):
    """getSag of ZernikeFringe, ZernikeANSI and a decentred LinearCombination (monomial expansion
    on the device) == the reference's values on scattered points; getGrad == the derivative of the
    reference's sag"""
    import os
    api = zoo.mirror_api()
    z = np.load(os.path.join(_golden.GOLDEN_DIR, "zernike_shapes.npz"))
    (x, y) = (z["x"], z["y"])
    lc = api.LocalCoordinates.p(name="zshape")
    lcz = lc.addChild(api.LocalCoordinates.p(name="zshape_dec", decx=0.7, decy=-1.1))
    shapes = {"fringe": api.ZernikeFringe.p(lc, normradius=9.0, coefficients=zoo.ZERNIKE_FRINGE_COEFFS),
              "ansi": api.ZernikeANSI.p(lc, normradius=9.0, coefficients=zoo.ZERNIKE_ANSI_COEFFS),
              "combination": api.LinearCombination.p(lc, list_of_coefficients_and_shapes=[
                  (0.8, api.Asphere.p(lc, curv=-1. / 90., cc=-0.8, coefficients=[0.0, 2e-6])),
                  (1.3, api.ZernikeFringe.p(lcz, normradius=12.0, coefficients=zoo.ZERNIKE_FRINGE_COEFFS[:12]))])}
    for (key, sh) in shapes.items():
        assert np.allclose(sh.getSag(x, y), z[key + "_sag"], rtol=0, atol=1e-13), key
        g = sh.getGrad(x, y)              # = -d(sag)/dx, -d(sag)/dy, 1 of the reference's sag (see _golden.py)
        assert np.allclose(-g[0], z[key + "_dsag_dx"], rtol=0, atol=1e-12), key
        assert np.allclose(-g[1], z[key + "_dsag_dy"], rtol=0, atol=1e-12), key
        assert np.all(g[2] == 1.0)
    # finite at the origin of the Zernike frame (the reference's polar formula is 0/0 there)
    assert np.all(np.isfinite(shapes["fringe"].getGrad(np.zeros(1), np.zeros(1))))


def test_rotated_combination_shape_evaluation_on_the_device(gpu_device):
    """getSag / getGrad of a LinearCombination whose polynomial part is decentred and rotated about the axis ==
    the reference's values on scattered points (prt_shape_eval on the monomials rotated and shifted on the host)"""
    import os
    api = zoo.mirror_api()
    z = np.load(os.path.join(_golden.GOLDEN_DIR, "rotated_combination_shape.npz"))
    (s, seq) = zoo.rotated_combination_system(api)
    sh = s.elements["rc"].surfaces["front"].shape
    assert np.allclose(sh.getSag(z["x"], z["y"]), z["sag"], rtol=0, atol=2e-14)
    assert np.allclose(sh.getGrad(z["x"], z["y"]), z["grad"], rtol=0, atol=2e-14)


def test_gridsag_shape_evaluation_on_the_device(gpu_device):
    """bicubic B-spline on the device == scipy / FITPACK in the reference: sag and gradient on
    scattered points incl. grid corners and points outside the grid (clamped arguments)"""
    import os
    api = zoo.mirror_api()
    z = np.load(os.path.join(_golden.GOLDEN_DIR, "gridsag_shape.npz"))
    sh = api.GridSag.p(api.LocalCoordinates.p(name="gshape"), zoo.gridsag_data())
    assert np.allclose(sh.getSag(z["x"], z["y"]), z["sag"], rtol=0, atol=2e-15)
    assert np.allclose(sh.getGrad(z["x"], z["y"]), z["grad"], rtol=0, atol=2e-14)


@pytest.mark.parametrize("tag,wave", [("red", 0.700e-3), ("blue", 0.470e-3)])
def test_raytrace_prism_like_the_reference_demo(gpu_device, tag, wave):
    """demos/demo_prism.py through ``raytrace``: MeridionalFan raster, start offset, field angle,
    ModelGlass dispersion -- initial bundle and every traced bundle == the reference's"""
    from pyrate_amd.builders import raytrace
    from pyrate_amd.sampling2d import raster
    from test_gpu_dropin import assert_paths_match
    api = zoo.mirror_api()
    case = _golden.load_case("prism_" + tag)
    (s, seq) = zoo.prism(api)
    rd = dict(zoo.PRISM_RAYS)
    rd["raster"] = raster.MeridionalFan()
    r = raytrace(s, seq, 128, rd, wave=wave)
    rp = r[0][0]
    assert np.allclose(rp.raybundles[0].x[0], case.x0, rtol=0, atol=1e-14)
    assert np.allclose(np.real(rp.raybundles[0].k[0]), np.real(case.k0), rtol=0, atol=1e-15)
    assert_paths_match(rp, case.raw_bundles)


def test_device_bundles_on_every_deterministic_raster_equal_the_reference(gpu_device):
    """collimated AND divergent bundles generated on the GPU (prt_raster_bundle) on RectGrid,
    HexGrid, Meridional / SagitalFan and CircularGrid, in air and in a dense background medium, against
    the reference's own bundles (tests/golden/bundles.json, oracle/make_golden.py): pupil samples and
    origins bit for bit, wave vectors to 2e-15 (device sin / cos vs libm: a few ulp; the reference itself
    gets k from a per-ray eigenproblem), |k| = n, E a unit vector perpendicular to k; shards of the
    raster equal slices of the whole"""
    import json
    import os
    import torch
    from pyrate_amd import engine
    from pyrate_amd.sampling2d import raster
    ref = json.load(open(os.path.join(_golden.GOLDEN_DIR, "bundles.json")))
    objs = {"rect_60": raster.RectGrid(), "hex_45": raster.HexGrid(), "meridional_9": raster.MeridionalFan(),
            "sagital_8": raster.SagitalFan(), "circular_49": raster.CircularGrid()}
    for (key, case) in ref.items():
        pd = case["props"]
        tables = objs[case["raster"]].device_tables(case["nray"])
        (gx, gy) = objs[case["raster"]].getGrid(case["nray"])
        start = (pd["startx"], pd["starty"], pd["startz"])
        kw = dict(radius=pd["radius"], start=start, anglex=pd["anglex"], angley=pd["angley"], index=case["index"])
        if case["bundle"] == "collimated":
            unit = np.array([math.sin(pd["angley"]) * math.cos(pd["anglex"]), math.sin(pd["anglex"]),
                             math.cos(pd["angley"]) * math.cos(pd["anglex"])])
            kw.update(kvec=case["index"] * unit, evec=np.array([1.0, 0.0, 0.0]))
        (x, k, e, total, pup) = engine.raster_bundle_device(tables, case["bundle"], gpu_device, want_pupil=True, **kw)
        (xr, kr) = (np.array(case["x"]), np.array(case["k"]))
        assert total == xr.shape[1] == gx.shape[0], key
        assert np.array_equal(pup.cpu().numpy(), np.vstack((gx, gy))), key
        assert np.array_equal(x.cpu().numpy(), xr), key
        kd = k.cpu().numpy()
        assert np.abs(kd - kr).max() < 2e-15, (key, np.abs(kd - kr).max())
        assert np.abs(np.sqrt((kd ** 2).sum(axis=0)) - case["index"]).max() < 4e-16, key
        if case["bundle"] == "divergent":
            ed = e.cpu().numpy()
            assert np.abs((ed * kd).sum(axis=0)).max() < 1e-15 and np.abs((ed ** 2).sum(axis=0) - 1).max() < 1e-15
        # a shard that straddles the two lattices of the hex raster / an arbitrary slice of the others
        (lo, hi) = (total // 3, total // 3 + max(1, total // 2))
        (xs, ks, es, _) = engine.raster_bundle_device(tables, case["bundle"], gpu_device, lo=lo, hi=hi, **kw)
        assert torch.equal(xs, x[:, lo:hi]) and torch.equal(ks, k[:, lo:hi])


def test_aim_generates_divergent_and_fan_bundles_on_the_device(gpu_device):
    """OpticalSystemAnalysis.aim (optical_system_analysis.py:167-181) with a divergent bundle on a fan: the
    initial bundle is device resident (nothing uploaded) and equals the host form of divergent_bundle;
    tracing it gives the same path as tracing the host bundle"""
    from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
    from pyrate_amd.sampling2d import raster
    api = zoo.mirror_api()
    (s, seq) = zoo.doublet(api)
    osa = OpticalSystemAnalysis(s, seq)
    props = {"startz": -40.0, "radius": 0.2, "raster": raster.MeridionalFan()}
    osa.aim(15, props, bundletype="divergent", wave=zoo.DLINE)
    ib = osa.initial_bundles[0]
    (o, k, e) = osa.divergent_bundle(15, props, wave=zoo.DLINE)
    assert np.array_equal(ib.x[0], o) and np.abs(np.real(ib.k[0]) - k).max() < 1e-15
    paths_dev = osa.trace()[0]
    from pyrate_amd.raytracer.ray import RayBundle
    paths_host = s.seqtrace(RayBundle(x0=o, k0=k, Efield0=e, wave=zoo.DLINE), seq)
    for (a, b) in zip(paths_dev[0].raybundles, paths_host[0].raybundles):
        assert np.allclose(a.x, b.x, rtol=0, atol=1e-12) and np.array_equal(a.valid, b.valid)
