"""
GPU tests of ABI v5 (run with -m gpu on an MI355X), all through the C ABI:

  * the UNIFORM first segment of prt_trace_ex (k0 = NULL: one wave vector and one E field for the whole
    bundle -- the reference's collimated bundles, analysis/optical_system_analysis.py:83-122): results equal
    the array form BIT FOR BIT, at BASELINE's full sizes, in both marches (isotropic, crystals), every kind of
    first-segment direction, aligned and unaligned buffers, path / image mode, with the fused moments;
  * the wrappers prt_trace / prt_trace_fields / prt_trace_moments / prt_trace_timed against prt_trace_ex;
  * RayBundle: host arrays with equal columns are recognised as a uniform bundle, ``aim`` creates uniform
    bundles, and the reference-shaped (P,3,N) views are what they were;
  * the "conics + aspheres + XY polynomials + biconics" instantiation at full size (an XY-polynomial system
    of 1e7 rays: residual on the surface, |k| = n, masks and a sub-sample against the oracle).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import seqtrace_np as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(gpu_device):
    import systems_zoo as zoo
    return zoo.mirror_api()


def _bits(t):
    return t.contiguous().view(torch.int64)


def _same(a, b):
    """two TraceResults, bit for bit (NaNs included)"""
    assert len(a.x_hit) == len(b.x_hit)
    for s in range(len(a.x_hit)):
        assert torch.equal(_bits(a.x_hit[s]), _bits(b.x_hit[s])), s
        assert torch.equal(_bits(a.k_out[s]), _bits(b.k_out[s])), s
        assert torch.equal(a.valid[s], b.valid[s]), s
        if a.valid_out[s] is not None:
            assert torch.equal(a.valid_out[s], b.valid_out[s]), s


def _configs():
    from pyrate_amd import systems
    c = systems.CALCITE_TILTED
    return {
        "doublegauss": (systems.double_gauss_records(), dict(), 10_000_000),
        "asphere": (systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5),
                    dict(rpup=9.0, z0=-5.0, field_deg=5.0), 10_000_000),
        "xypoly": (systems.xypoly_records(), dict(rpup=9.0, z0=-5.0, field_deg=5.0), 10_000_000),
        "aniso": (systems.aniso_doublet_records(
            systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
            systems.uniaxial_eps(1.6727, 1.60, (np.sin(0.2), 0.0, np.cos(0.2)))),
            dict(rpup=11.43, z0=-5.0, field_deg=0.0), 1_000_000),
    }


@pytest.mark.parametrize("config", ["doublegauss", "asphere", "xypoly", "aniso"])
def test_uniform_first_segment_equals_the_array_form_bit_for_bit_at_full_size(config, gpu_device):
    """BASELINE configs[1], [2], [3] and the XY-polynomial system at their full sizes: the bundle generated as
    arrays x0, k0, E0 and as x0 + one (k, E) pair gives the same path arrays, bit for bit, in path and image mode"""
    from pyrate_amd import engine, systems, _lib
    (recs, bargs, nrays) = _configs()[config]
    sysd = engine.DeviceSystem(recs, 0)
    (x0, k0, e0d, n) = systems.double_gauss_bundle_device(nrays, gpu_device, **bargs)
    (xu, uni, none, nu) = systems.double_gauss_bundle_device(nrays, gpu_device, uniform=True, **bargs)
    assert nu == n and none is None and isinstance(uni, engine.UniformFirst)
    assert torch.equal(_bits(xu), _bits(x0))
    assert torch.equal(uni.rows(n, gpu_device, "k"), k0) and torch.equal(uni.rows(n, gpu_device, "e_re"), e0d)
    iso = sysd.all_isotropic
    for mode in (_lib.MODE_PATH, _lib.MODE_IMAGE):
        ref = sysd.trace(x0, k0, e0d, mode=mode, packed_flags=iso)
        got = sysd.trace(xu, None, mode=mode, packed_flags=iso, uniform=uni)
        _same(got, ref)
        del ref, got
    # a trace that really happened: rays arrive
    res = sysd.trace(xu, None, mode=_lib.MODE_IMAGE, uniform=uni)
    assert int(res.valid[-1].sum()) > 0.9 * res.valid[-1].numel()


@pytest.mark.parametrize("kind", ["default_e", "complex_e", "k", "dir"])
def test_every_kind_of_uniform_first_direction(kind, gpu_device):
    """E = ey by default (ray.py:71-73), a complex E, d = k/|k|, d given: each against the arrays that say the same,
    on a tilted bundle through the double Gauss (conic march), an asphere and the crystal doublet; aligned
    (pitched) and unaligned (odd ray count, tight) buffers; fused moments"""
    from pyrate_amd import engine, systems, _lib
    cfg = _configs()
    n = 30001
    (o, k, _) = systems.double_gauss_bundle(n, rpup=4.0, field_deg=4.0)
    n = o.shape[1]
    kvec = k[:, 0].copy()
    if kind == "default_e":
        (e_arr, uni, first_dir) = (None, engine.UniformFirst(kvec), None)
    elif kind == "complex_e":
        evec = np.array([0.3 + 0.2j, 0.5 - 0.1j, 0.05 + 0.4j])      # not perpendicular to k: walk-off direction
        e_arr = np.repeat(evec[:, None], n, axis=1)
        (uni, first_dir) = (engine.UniformFirst(kvec, evec), None)
    elif kind == "k":
        (e_arr, uni, first_dir) = (None, engine.UniformFirst(kvec, kind="k"), _lib.FIRST_K)
    else:
        dvec = np.array([0.01, 0.05, 1.0])
        dvec /= np.linalg.norm(dvec)
        e_arr = np.repeat(dvec[:, None], n, axis=1)
        (uni, first_dir) = (engine.UniformFirst(kvec, dvec, kind="dir"), _lib.FIRST_DIR)
    for pitched in (True, False):
        x0 = engine.to_device_rays(o, gpu_device, pitched=pitched)
        k0 = engine.to_device_rays(k, gpu_device, pitched=pitched)
        e_re = e_im = None
        if e_arr is not None:
            e_re = engine.to_device_rays(np.ascontiguousarray(e_arr.real), gpu_device, pitched=pitched)
            if np.iscomplexobj(e_arr):
                e_im = engine.to_device_rays(np.ascontiguousarray(e_arr.imag), gpu_device, pitched=pitched)
        for name in ("doublegauss", "asphere", "aniso"):
            sysd = engine.DeviceSystem(cfg[name][0], 0)
            if not sysd.all_isotropic and pitched:
                continue
            for mode in (_lib.MODE_PATH, _lib.MODE_IMAGE):
                pitch = None if pitched else n            # tight, odd count: the unaligned instantiations
                bufs_a = sysd.alloc_outputs(n, mode, pitch=pitch if sysd.all_isotropic else None)
                bufs_u = sysd.alloc_outputs(n, mode, pitch=pitch if sysd.all_isotropic else None)
                sysd.trace_into(x0, k0, bufs_a, e_re, e_im, first_dir=first_dir)
                sysd.trace_into(x0, None, bufs_u, uniform=uni)
                _same(sysd.views(bufs_u), sysd.views(bufs_a))
        # the march that also reduces the image-plane moments
        sysd = engine.DeviceSystem(cfg["doublegauss"][0], 0)
        ws = engine.MomentsWorkspace(gpu_device, n_results=2, n_rays=n)
        bufs = sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=True, pitch=None if pitched else n + 1)
        if first_dir is None:
            ma = sysd.trace_moments_into(x0, k0, bufs, ws, 0, e_re, e_im).clone()
            mu = sysd.trace_moments_into(x0, None, bufs, ws, 1, uniform=uni)
            assert torch.equal(_bits(ma), _bits(mu)) and float(ma[0]) > 0


def test_uniform_bundle_through_more_crystals_than_the_fused_walk_parks(gpu_device):
    """nine crystal interfaces: the per-surface march (it reads per-ray arrays; the uniform vectors are
    broadcast once) gives what it gives for arrays"""
    from pyrate_amd import engine, systems
    c = systems.CALCITE_TILTED
    eps = systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"])
    build = [({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {})]
    for q in range(9):
        build.append(({"shape": "Conic", "curv": 0.002 * (q - 4)}, {"decz": 2.0}, {"eps": eps * (1 + 0.01 * q)}, "c%d" % q, {}))
    build.append(({"shape": "Conic"}, {"decz": 5.0}, None, "image", {}))
    recs = systems.simple_system_records(build)
    (o, k, e) = systems.double_gauss_bundle(40, rpup=2.0)
    (x0, k0, e0) = [engine.to_device_rays(a, gpu_device, pitched=False) for a in (o, k, e)]
    sysd = engine.DeviceSystem(recs, 0)
    ref = sysd.trace(x0, k0, e0)
    got = sysd.trace(x0, None, uniform=engine.UniformFirst(k[:, 0], e[:, 0]))
    assert ref.x_hit[-1].shape[1] == o.shape[1] * 2 ** 9
    _same(got, ref)


def test_structural_misuse_of_the_uniform_first_segment_is_an_error_code(gpu_device):
    from pyrate_amd import engine, systems, _lib
    sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
    (o, k, e) = systems.double_gauss_bundle(100)
    (x0, k0, e0) = [engine.to_device_rays(a, gpu_device) for a in (o, k, e)]
    bufs = sysd.alloc_outputs(o.shape[1])
    a = sysd._trace_args(x0, k0, bufs, e0)
    a.k0 = None                                       # E arrays with a uniform k
    assert sysd.lib.prt_trace_ex(sysd._h, ctypes.byref(a)) == -1
    a = sysd._trace_args(x0, k0, bufs)
    a.first_dir = _lib.FIRST_E_UNIFORM                # uniform E with k arrays
    assert sysd.lib.prt_trace_ex(sysd._h, ctypes.byref(a)) == -1
    a = sysd._trace_args(x0, k0, bufs)
    a.first_dir = 17
    assert sysd.lib.prt_trace_ex(sysd._h, ctypes.byref(a)) == -1
    a = sysd._trace_args(x0, k0, bufs)
    a.struct_bytes -= 8
    assert sysd.lib.prt_trace_ex(sysd._h, ctypes.byref(a)) == -1
    # the wrappers keep refusing a NULL k0
    P = engine._ptr
    assert sysd.lib.prt_trace(sysd._h, o.shape[1], x0.stride(0), P(x0), None, None, None, 0, bufs["pitch"],
                              P(bufs["x_hit"]), P(bufs["k_out"]), P(bufs["valid"]), P(bufs["valid_out"]), None,
                              engine._stream_handle(gpu_device)) == -1


def test_wrappers_equal_prt_trace_ex(gpu_device):
    """prt_trace, prt_trace_fields, prt_trace_moments, prt_trace_timed are thin wrappers of prt_trace_ex: same
    bits into the same kind of buffers"""
    from pyrate_amd import engine, systems, _lib
    lib = _lib.load()
    P = engine._ptr
    st = engine._stream_handle(gpu_device)
    (o, k, e) = systems.double_gauss_bundle(5000, field_deg=2.0)
    n = o.shape[1]
    (x0, k0, e0) = [engine.to_device_rays(a, gpu_device) for a in (o, k, e)]
    sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
    ref = sysd.trace(x0, k0, e0, want_nonconv=True)
    bufs = sysd.alloc_outputs(n, want_nonconv=True)
    _lib.check(lib.prt_trace(sysd._h, n, x0.stride(0), P(x0), P(k0), P(e0), None, 0, bufs["pitch"], P(bufs["x_hit"]),
                             P(bufs["k_out"]), P(bufs["valid"]), P(bufs["valid_out"]), P(bufs["nonconv"]), st))
    _same(sysd.views(bufs), ref)
    ms = ctypes.c_double()
    bufs2 = sysd.alloc_outputs(n)
    _lib.check(lib.prt_trace_timed(sysd._h, n, x0.stride(0), P(x0), P(k0), P(e0), None, 0, bufs2["pitch"],
                                   P(bufs2["x_hit"]), P(bufs2["k_out"]), P(bufs2["valid"]), P(bufs2["valid_out"]), st,
                                   3, ctypes.byref(ms)))
    _same(sysd.views(bufs2), ref)
    assert 0 < ms.value < 100 and abs(ms.value - sysd.trace_timed(x0, k0, bufs2, 3, e0)) < 1.0
    ws = engine.MomentsWorkspace(gpu_device, n_results=2, n_rays=n)
    bufs3 = sysd.alloc_outputs(n, packed_flags=True)
    m_ex = sysd.trace_moments_into(x0, k0, bufs3, ws, 0, e0).clone()
    _lib.check(lib.prt_trace_moments(sysd._h, n, x0.stride(0), P(x0), P(k0), P(e0), None, _lib.MODE_FLAGS,
                                     bufs3["pitch"], P(bufs3["x_hit"]), P(bufs3["k_out"]), P(bufs3["valid"]), None,
                                     None, P(ws.out[1]), P(ws.scratch), st))
    assert torch.equal(_bits(ws.out[1]), _bits(m_ex)) and float(m_ex[0]) == float(ref.valid_out[-1].sum())
    # crystals: prt_trace_fields
    c = systems.CALCITE_TILTED
    recs = systems.aniso_doublet_records(systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]))
    sysc = engine.DeviceSystem(recs, 0)
    (xt, kt, et) = [engine.to_device_rays(a, gpu_device, pitched=False) for a in (o, k, e)]
    refc = sysc.trace(xt, kt, et, want_fields=True)
    b = sysc.alloc_outputs(n, want_fields=True, pitch=0)        # the wrapper's layout is the tight one
    _lib.check(lib.prt_trace_fields(sysc._h, n, P(xt), P(kt), P(et), None, 0, P(b["x_hit"]), P(b["k_out"]),
                                    P(b["e_re"]), P(b["e_im"]), P(b["valid"]), P(b["valid_out"]), st))
    got = sysc.views(b)
    _same(got, refc)
    for s in range(len(recs)):
        if recs[s]["material"]["type"] == "anisotropic":
            assert torch.equal(_bits(got.e_out[s][0]), _bits(refc.e_out[s][0]))


def test_collimated_host_arrays_become_a_uniform_bundle(gpu_device, api):
    """RayBundle(x0, k0, E0) with the arrays the reference's collimated_bundle returns (equal columns): recognised,
    nothing but x0 is uploaded, the trace equals the one of a bundle that was forced to keep its arrays, and the
    reference-shaped views k / Efield are unchanged; ``aim`` generates uniform bundles on the device"""
    from pyrate_amd import systems
    from pyrate_amd.raytracer import ray as prt_ray
    (s, seq) = api.build_rotationally_symmetric_optical_system(
        [(r, cc, t, m, name, opts) for (r, cc, t, m, name, opts) in systems.double_gauss_tuples()])
    (o, k, e0) = systems.double_gauss_bundle(20000, field_deg=3.0)
    n = o.shape[1]
    kc = k.astype(complex)                                   # the reference hands over complex128 wave vectors
    uni = api.RayBundle(x0=o, k0=kc, Efield0=e0)
    assert uni._uniform is not None and uni._k[0].stride(1) == 0
    old = prt_ray.UNIFORM_DETECT_MIN_RAYS
    prt_ray.UNIFORM_DETECT_MIN_RAYS = 10 ** 12
    try:
        arr = api.RayBundle(x0=o, k0=kc, Efield0=e0)
    finally:
        prt_ray.UNIFORM_DETECT_MIN_RAYS = old
    assert arr._uniform is None and arr._k[0].stride(1) == 1
    assert uni.k.dtype == arr.k.dtype == np.complex128
    assert np.array_equal(uni.k, arr.k) and np.array_equal(uni.Efield, arr.Efield) and uni.k.shape == (1, 3, n)
    pu = s.seqtrace(uni, seq)[0]
    pa = s.seqtrace(arr, seq)[0]
    assert len(pu.raybundles) == len(pa.raybundles)
    for (bu, ba) in zip(pu.raybundles, pa.raybundles):
        assert np.array_equal(bu.x, ba.x, equal_nan=True) and np.array_equal(bu.k, ba.k, equal_nan=True)
        assert np.array_equal(bu.valid, ba.valid) and np.array_equal(bu.rayID, ba.rayID)
    # a bundle that is not collimated keeps its arrays
    k2 = k.copy()
    k2[0, 5] += 1e-9
    assert api.RayBundle(x0=o, k0=k2, Efield0=e0)._uniform is None
    # aim: collimated bundles are generated uniform; the per-surface (plugin) path accepts them too
    from pyrate_amd.raytracer.analysis.optical_system_analysis import OpticalSystemAnalysis
    osa = OpticalSystemAnalysis(s, seq)
    osa.aim(20000, {"radius": 5.0, "startz": -10.0, "anglex": 0.03}, bundletype="collimated", wave=systems.DLINE)
    b = osa.initial_bundles[0]
    assert b._uniform is not None and b.k.shape == (1, 3, b.x.shape[2]) and b.Efield.shape == b.k.shape
    assert np.all(b.k == b.k[:, :, :1]) and abs(np.linalg.norm(b.k[0, :, 0]) - 1.0) < 1e-15
    assert np.max(np.abs(np.sum(b.k[0] * b.Efield[0], axis=0))) < 1e-15
    fused = osa.trace()[0][0]
    generic = s._seqtrace_generic(b, seq, False)[0]
    for (bf, bg) in zip(fused.raybundles, generic.raybundles):
        assert np.array_equal(bf.valid, bg.valid)
        assert np.allclose(bf.x, bg.x, rtol=0, atol=1e-11, equal_nan=True)
    m = s.image_moments(b, seq)
    assert m[0][0] == fused.raybundles[-1].x.shape[2]


def test_xypoly_march_at_full_size(gpu_device):
    """the instantiation "conics + even aspheres + XY polynomials + biconics" (BASELINE's north_star shapes) at
    1e7 rays: demo_asphere's geometry with a 12-term XY polynomial as the back surface, 5 degree field.  Every
    hit point lies on the polynomial surface to 1e-13 mm, |k| = n behind every surface, nothing ends at the
    Newton cap, masks and a 1e4-ray sub-sample equal the oracle's"""
    from pyrate_amd import engine, systems
    recs = systems.xypoly_records()
    assert len(recs[2]["shape"]["terms"]) == 12
    sysd = engine.DeviceSystem(recs, 0)
    (x0, uni, _, n) = systems.double_gauss_bundle_device(10000000, gpu_device, rpup=9.0, z0=-5.0, field_deg=5.0,
                                                         uniform=True)
    res = sysd.trace(x0, None, packed_flags=True, uniform=uni)
    assert all(int(res.nonconv[s].sum()) == 0 for s in range(4))
    p = res.x_hit[2] - torch.tensor(recs[2]["g_shape"], dtype=torch.float64, device=gpu_device)[:, None]
    F = torch.zeros_like(p[0])
    for (i, j, c) in recs[2]["shape"]["terms"]:
        F = F + c * p[0] ** i * p[1] ** j
    ok = res.valid[2].bool()
    assert int(ok.sum()) == n
    assert float((p[2] - F)[ok].abs().max()) < 1e-13
    for s in range(4):
        m = res.valid_out[s].bool()
        kk = res.k_out[s][:, m]
        assert float(((kk ** 2).sum(dim=0).sqrt() - recs[s]["material"]["n"]).abs().max()) < 1e-14
    idx = torch.arange(0, n, 1000, device=gpu_device)
    o_s = x0[:, idx].cpu().numpy()
    k_s = np.repeat(np.array(uni.k)[:, None], o_s.shape[1], axis=1)
    e_s = np.repeat(np.array(uni.e_re)[:, None], o_s.shape[1], axis=1)
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, o_s, k_s, e_s)
    for s in range(4):
        v = out[s]["valid_out"]
        assert np.array_equal(res.valid_out[s][idx].cpu().numpy().astype(bool), v)
        assert np.abs(res.x_hit[s][:, idx].cpu().numpy()[:, v] - out[s]["x_hit"][:, v]).max() < 1e-10
        assert np.abs(res.k_out[s][:, idx].cpu().numpy()[:, v] - out[s]["k_out"][:, v]).max() < 1e-12


def test_dense_polynomial_in_scalar_registers_equals_the_streamed_evaluation(gpu_device):
    """XY polynomials of total degree <= 4 are evaluated from scalar registers (dense_poly_eval, coefficients
    fetched once per surface); anything bigger streams its Horner rows (xypoly_eval).  Same Horner order: a
    degree-5 term too small to matter (1e-300) sends the same surface down the streaming path -- bit-identical
    path arrays, tilted field, 1e6 rays; and a sparse polynomial (missing rows / columns) against the oracle"""
    from pyrate_amd import engine, systems
    (x0, uni, _, n) = systems.double_gauss_bundle_device(1000000, gpu_device, rpup=9.0, z0=-5.0, field_deg=5.0,
                                                         uniform=True)
    recs = systems.xypoly_records()
    streamed = systems.xypoly_records()
    streamed[2]["shape"]["terms"] = list(streamed[2]["shape"]["terms"]) + [[5, 0, 1e-300], [0, 5, 1e-300]]
    a = engine.DeviceSystem(recs, 0).trace(x0, None, packed_flags=True, uniform=uni)
    b = engine.DeviceSystem(streamed, 0).trace(x0, None, packed_flags=True, uniform=uni)
    _same(a, b)
    assert int(a.valid_out[-1].sum()) == n
    sparse = systems.xypoly_records()
    sparse[2]["shape"]["terms"] = [[2, 0, -1. / 55.], [0, 2, -1. / 65.], [0, 3, 2e-5], [4, 0, -1e-6], [1, 1, 1e-4],
                                   [0, 0, 0.01], [1, 0, 1e-3]]
    res = engine.DeviceSystem(sparse, 0).trace(x0, None, packed_flags=True, uniform=uni)
    idx = torch.arange(0, n, 500, device=gpu_device)
    o_s = x0[:, idx].cpu().numpy()
    k_s = np.repeat(np.array(uni.k)[:, None], o_s.shape[1], axis=1)
    e_s = np.repeat(np.array(uni.e_re)[:, None], o_s.shape[1], axis=1)
    out = oracle.trace(sparse, o_s, k_s, e_s)
    for s in range(4):
        v = out[s]["valid_out"]
        assert v.all() and np.array_equal(res.valid_out[s][idx].cpu().numpy().astype(bool), v)
        assert np.abs(res.x_hit[s][:, idx].cpu().numpy() - out[s]["x_hit"]).max() < 1e-11
        assert np.abs(res.k_out[s][:, idx].cpu().numpy() - out[s]["k_out"]).max() < 1e-13


@pytest.mark.parametrize("n_req", [1000000, 777, 130])
def test_crystal_layout_with_a_ray_pitch_equals_the_tight_layout(n_req, gpu_device):
    """tables with crystals: the concatenated layout with ray pitch P = n rounded up to 128 (rows of every level on
    128-B lines, what alloc_outputs picks) holds the same numbers as the tight layout, bit for bit -- path and
    image mode, E fields included; the padding slots of a branch carry mask 0 and are never written"""
    from pyrate_amd import engine, systems, _lib
    (recs, bargs, _) = _configs()["aniso"]
    sysd = engine.DeviceSystem(recs, 0)
    (x0, uni, _, n) = systems.double_gauss_bundle_device(n_req, gpu_device, uniform=True, **bargs)
    P = int(_lib.load().prt_crystal_pitch(n))
    assert P % 128 == 0 and 0 <= P - n < 128
    for mode in (_lib.MODE_PATH, _lib.MODE_IMAGE):
        for fields in (False, True):
            tight = sysd.alloc_outputs(n, mode, pitch=0, want_fields=fields)
            pitched = sysd.alloc_outputs(n, mode, want_fields=fields)
            assert tight["pitch"] == 0 and pitched["pitch"] == P
            for b in (tight, pitched):
                b["x_hit"].fill_(-7.0)
                b["k_out"].fill_(-7.0)
                sysd.trace_into(x0, None, b, uniform=uni)
            (rt, rp) = (sysd.views(tight), sysd.views(pitched))
            _same(rp, rt)
            if fields:
                for s in range(len(rt.x_hit)):
                    if rt.e_out[s] is not None and recs[s if mode == _lib.MODE_PATH else -1]["material"]["type"] == "anisotropic":
                        assert torch.equal(_bits(rp.e_out[s][0]), _bits(rt.e_out[s][0]))
            if P != n:
                for s in range(len(rt.x_hit)):
                    B = rp.padded.valid_out[s].numel() // P
                    assert int(rp.padded.valid_out[s].view(B, P)[:, n:].sum()) == 0
                    assert int(rp.padded.valid[s].view(-1, P)[:, n:].sum()) == 0
                    assert float(rp.padded.k_out[s].view(3, B, P)[:, :, n:].min()) == -7.0        # untouched
                    assert float(rp.padded.k_out[s].view(3, B, P)[:, :, n:].max()) == -7.0
    # pitched inputs are taken as they are (no tight copy), and give the same bits
    (xa, ka, ea, _) = systems.double_gauss_bundle_device(n_req, gpu_device, **bargs)
    assert n < 512 or xa.stride(0) != n
    _same(sysd.trace(xa, ka, ea), sysd.trace(xa.contiguous(), ka.contiguous(), ea.contiguous()))


@pytest.mark.parametrize("packed", [True, False])
def test_image_plane_redirect_into_the_gathers_receive_buffer(packed, gpu_device):
    """prt_trace_ex's image-plane redirect: the march writes the LAST surface's record into this rank's slot of an
    ImagePlaneGather's receive buffer (own_rows) instead of into its rows of the path arrays -- same bits as the
    plain trace, the skipped rows of the path arrays untouched, the fused moments unchanged; the in-place all-gather
    (a single rank here) then holds the image plane in global ray order"""
    from pyrate_amd import engine, systems, _lib
    from pyrate_amd import distributed as pdist
    sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
    (x0, uni, _, n) = systems.double_gauss_bundle_device(300000, gpu_device, field_deg=9.0, uniform=True)
    S = sysd.n_surfaces
    ref = sysd.trace(x0, None, packed_flags=packed, uniform=uni)
    g = pdist.ImagePlaneGather(n, gpu_device, world=1, rank=0, align=512)
    rows = g.own_rows()
    if not packed:
        rows = rows + (torch.zeros(n, dtype=torch.uint8, device=gpu_device),)
    bufs = sysd.alloc_outputs(n, packed_flags=packed)
    bufs["x_hit"].fill_(-3.0)
    bufs["k_out"].fill_(-3.0)
    bufs["image_rows"] = rows
    ws = engine.MomentsWorkspace(gpu_device, n_results=2, n_rays=n)
    m_red = sysd.trace_moments_into(x0, None, bufs, ws, 0, uniform=uni).clone()
    got = sysd.views(bufs)
    _same(got, ref)
    plain = sysd.alloc_outputs(n, packed_flags=packed)
    m_plain = sysd.trace_moments_into(x0, None, plain, ws, 1, uniform=uni)
    assert torch.equal(_bits(m_red), _bits(m_plain)) and 0 < float(m_red[0]) <= n
    # the last surface's rows of the path arrays were not written
    pitch = bufs["pitch"]
    assert float(bufs["x_hit"][3 * (S - 1) * pitch:].max()) == -3.0 and float(bufs["k_out"][3 * (S - 1) * pitch:].min()) == -3.0
    # without moments, and through the launcher
    bufs2 = sysd.alloc_outputs(n, packed_flags=packed)
    bufs2["image_rows"] = rows
    for r in rows[:2]:
        r.fill_(0.0)
    sysd.launcher(x0, None, bufs2, uniform=uni)()
    _same(sysd.views(bufs2), ref)
    # the gather: in place, one rank -> nothing to move; finish() = the image plane
    g.start_in_place()
    (gx, gk, gv) = g.finish()
    assert torch.equal(_bits(gx), _bits(ref.x_hit[-1])) and torch.equal(_bits(gk), _bits(ref.k_out[-1]))
    assert torch.equal(gv, ref.flags[-1] if packed else ref.valid[-1])
    # structural misuse: image mode, or odd pitch
    bad = sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=packed)
    bad["image_rows"] = rows
    with pytest.raises(_lib.PrtError):
        sysd.trace_into(x0, None, bad, uniform=uni)


@pytest.mark.parametrize("eps_kind", ["uniaxial", "biaxial", "isotropic"])
def test_evanescent_modes_come_back_as_complex_wave_vectors(eps_kind, gpu_device):
    """dense glass -> crystal at steep incidence: transmitted modes turn evanescent.  The reference carries them as
    complex k (material/material.py:407-454); with ``want_fields`` the engine reports the same complex k in the slot
    the reference puts it in (k_out + i k_out_im; of a conjugate pair the root with Im(xi) > 0, the reference's pick
    being its sort's -- compared up to that sign), 0 imaginary part for propagating modes, which equal the oracle's;
    uniaxial (closed forms), biaxial (quartic) and isotropic-as-anisotropic eps; refraction and reflection"""
    from pyrate_amd import engine, systems
    eps = {"uniaxial": systems.uniaxial_eps(1.35, 2.1, (1.0, 0.0, 0.0)),
           "biaxial": np.diag([1.35 ** 2, 1.7 ** 2, 2.1 ** 2]),
           "isotropic": 1.4 ** 2 * np.eye(3)}[eps_kind]
    n = 64
    ang = np.linspace(0.2, 1.2, n)
    x0 = np.vstack((np.zeros(n), np.zeros(n), np.full(n, -1.0)))
    k0 = 1.9 * np.vstack((np.sin(ang) * 0.6, np.sin(ang) * 0.8, np.cos(ang)))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    for mirror in (False, True):
        recs = systems.simple_system_records([
            ({"shape": "Conic"}, {"decz": 0.0}, 1.9, "entry", {}),
            ({"shape": "Conic", "curv": 0.01}, {"decz": 5.0}, {"eps": eps}, "crystal", {}),
            ({"shape": "Conic", "curv": -0.01}, {"decz": 5.0}, {"eps": eps * 1.1}, "crystal2", {"is_mirror": mirror}),
            ({"shape": "Conic"}, {"decz": 5.0 if not mirror else -3.0}, None, "exit", {})], background_n=1.9)
        with np.errstate(all="ignore"):
            out = oracle.trace(recs, x0, k0, e0)
        sysd = engine.DeviceSystem(recs, 0)
        res = sysd.trace(*[engine.to_device_rays(a, gpu_device, pitched=False) for a in (x0, k0, e0)],
                         want_fields=True)
        assert res.k_out_im is not None
        seen = 0
        for s in (1, 2):
            ko = out[s]["k_out"]
            kd = res.k_out[s].cpu().numpy() + 1j * res.k_out_im[s].cpu().numpy()
            # rays that ENTER the surface as evanescent garbage are not compared (the reference traces them on along
            # the interface; the engine stops them)
            parent_ok = np.all(np.isfinite(res.x_hit[s].cpu().numpy()), axis=0)
            parent_ok = np.hstack((parent_ok, parent_ok))
            fin = np.all(np.isfinite(ko), axis=0) & parent_ok
            evan = fin & np.any(np.abs(np.imag(ko)) > 1e-9, axis=0)
            prop = fin & ~evan
            assert np.abs(kd[:, prop] - ko[:, prop]).max() < 1e-11
            assert np.abs(np.imag(kd[:, prop])).max() == 0.0
            tol = 1e-7 if eps_kind == "isotropic" else 1e-10     # (LAPACK's double roots of the degenerate pencil)
            # one mode evanescent, its partner propagating: the slot and the value are the reference's (up to the
            # sign of Im).  Both evanescent: the reference's S.n sort sees four zeros and leaves ANY two of the four
            # roots in the two slots -- no parity target; the engine's pair must solve the dispersion relation
            m = ko.shape[1] // 2
            partner = np.hstack((prop[m:], prop[:m]))
            one = evan & partner
            if one.any():
                assert np.abs(np.real(kd[:, one]) - np.real(ko[:, one])).max() < tol
                assert np.abs(np.abs(np.imag(kd[:, one])) - np.abs(np.imag(ko[:, one]))).max() < tol
            ev_eng = parent_ok & np.any(np.abs(np.imag(kd)) > 1e-9, axis=0)
            assert np.array_equal(ev_eng, evan)               # the same slots are evanescent
            seen += int(ev_eng.sum())
            eps_s = np.asarray(recs[s]["material"]["eps_re"])
            for q in np.nonzero(ev_eng)[0]:
                kq = kd[:, q]
                W = eps_s - np.sum(kq * kq) * np.eye(3) + np.outer(kq, kq)      # bilinear, like material.py:385-392
                assert abs(np.linalg.det(W)) < (1e-6 if eps_kind == "isotropic" else 1e-9) * np.linalg.norm(eps_s) ** 3
        assert seen > 5, (eps_kind, mirror, seen)
    # the reference's own bundle (golden case of the uniaxial slab): slot by slot
    if eps_kind == "uniaxial":
        import _golden
        import systems_zoo as zoo
        case = _golden.load_case("aniso_partial_evanescent")
        (xg, kg, eg) = zoo.evanescent_bundle_arrays()
        rg = engine.DeviceSystem(case.table, 0).trace(*[engine.to_device_rays(a, gpu_device, pitched=False)
                                                        for a in (xg, kg, eg)], want_fields=True)
        kref = case.raw_bundles[3]["k"][0]
        kd = rg.k_out[1].cpu().numpy() + 1j * rg.k_out_im[1].cpu().numpy()
        real_ref = np.all(np.abs(np.imag(kref)) < 1e-12, axis=0)
        assert (~real_ref).sum() > 5
        assert np.abs(kd[:, real_ref] - kref[:, real_ref]).max() < 1e-12
        assert np.abs(np.real(kd[:, ~real_ref]) - np.real(kref[:, ~real_ref])).max() < 1e-12
        assert np.abs(np.abs(np.imag(kd[:, ~real_ref])) - np.abs(np.imag(kref[:, ~real_ref]))).max() < 1e-12
        # ... and through the drop-in layer: RayBundle.k is complex there, like the reference's
        api = zoo.mirror_api()
        (s_sys, seq) = zoo.evanescent_slab(api)
        ib = api.RayBundle(x0=xg, k0=kg, Efield0=eg)
        path = s_sys.seqtrace(ib, seq)[0]
        kb = path.raybundles[3].k[0]
        assert kb.dtype == np.complex128 and kb.shape == kref.shape
        assert np.abs(np.real(kb) - np.real(kref)).max() < 1e-12
        assert np.abs(np.abs(np.imag(kb)) - np.abs(np.imag(kref))).max() < 1e-12
