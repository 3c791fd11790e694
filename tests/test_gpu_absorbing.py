"""``-m gpu``: crystals with a COMPLEX (absorbing) epsilon tensor (reference: material_anisotropic.py:52-56 accepts
one; SURVEY.md 8 a12 "c128") for sequences that stay inside crystals -- the engine (prt_trace_ex -> per-surface march
-> k_interact_aniso_cplx, csrc/prt_aniso_cplx.h) against the reference's own bundles and against the NumPy oracle,
all through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

import _golden
import systems_zoo as zoo
from oracle import seqtrace_np as oracle

pytestmark = pytest.mark.gpu


def _eps_of(case, s):
    m = case.table[s]["material"]
    return np.asarray(m["eps_re"]) + 1j * np.asarray(m["eps_im"])


def _rays(case, dev):
    from pyrate_amd import engine
    return [engine.to_device_rays(a, dev, pitched=False) for a in (case.x0, np.real(case.k0), case.E0)]


@pytest.mark.parametrize("name", _golden.ABSORBING_CASES)
def test_hip_vs_reference_absorbing_crystals(name, gpu_device):
    """hit points (1e-10 relative) and complex wave vectors (1e-10 absolute, real and imaginary part) of every
    surface against the reference's bundles; image mode gives the last surface's record bit for bit; the E fields
    solve the wave equation of their complex k"""
    from pyrate_amd import engine, _lib
    case = _golden.load_case(name)
    sysd = engine.DeviceSystem(case.table, 0)
    assert sysd.complex_eps and not sysd.all_isotropic
    res = sysd.trace(*_rays(case, gpu_device), want_fields=True)
    assert res.k_out_im is not None
    out = _golden.compare_dense_to_reference(case, _golden.dense_from_engine(res, complex_k=True),
                                             rtol_x=1e-10, atol_k=1e-10)
    assert out["n_compared"] >= 4 * case.x0.shape[1] and out["max_abs_k"] < 1e-12
    assert float(res.k_out_im[-1].abs().max()) > 1e-3 and float(res.k_out_im[0].abs().max()) == 0.0
    img = sysd.trace(*_rays(case, gpu_device), mode=_lib.MODE_IMAGE)
    for (a, b) in ((img.x_hit[-1], res.x_hit[-1]), (img.k_out[-1], res.k_out[-1]), (img.k_out_im[-1], res.k_out_im[-1]),
                   (img.valid_out[-1], res.valid_out[-1])):
        assert torch.equal(a, b)
    for s in (1, 2, 3):
        if case.table[s]["material"]["type"] != "anisotropic":
            continue
        eps = _eps_of(case, s)
        k = res.k_out[s].cpu().numpy() + 1j * res.k_out_im[s].cpu().numpy()
        (er, ei) = res.e_out[s]
        E = er.cpu().numpy() + 1j * ei.cpu().numpy()
        Bm = np.asarray(case.table[s]["B_mat"]).reshape(3, 3)
        (kl, El) = (Bm.T @ k, Bm.T @ E)
        # W E = eps E - (k.k) E + k (k.E) = 0, bilinear (material.py:385-392); |E|^2 (1 + |xi|^2) = 1
        resid = eps @ El - np.sum(kl * kl, axis=0) * El + kl * np.sum(kl * El, axis=0)
        assert np.abs(resid).max() < 1e-12 * np.abs(eps).max()
        assert np.all(np.sum(np.abs(El) ** 2, axis=0) < 1.0)


@pytest.mark.parametrize("seed", range(6))
def test_hip_vs_oracle_random_absorbing_crystals(seed, gpu_device):
    """random complex tensors (symmetric and not, weak and strong loss), curved tilted interfaces, steep incidence,
    refraction and reflection inside: engine == NumPy oracle (LAPACK) on every surface, complex k included"""
    from pyrate_amd import engine, systems
    rng = np.random.RandomState(100 + seed)

    def tensor():
        a = rng.uniform(1.3, 2.4, 3) ** 2
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        e = q @ np.diag(a) @ q.T + 1j * (q @ np.diag(rng.uniform(0.0, 0.3, 3) * (10.0 ** -rng.randint(0, 4))) @ q.T)
        if seed % 2:
            e = e + 0.02 * rng.normal(size=(3, 3)) * (1 + 0.5j)          # not symmetric
        return e
    mirror = bool(seed % 3 == 1)
    recs = systems.simple_system_records([
        ({"shape": "Conic"}, {"decz": 0.0}, None, "entry", {}),
        ({"shape": "Conic", "curv": 0.02}, {"decz": 4.0}, {"eps": tensor()}, "front", {}),
        ({"shape": "Conic", "curv": -0.015}, {"decz": 6.0}, {"eps": tensor()}, "rear", {"is_mirror": mirror}),
        ({"shape": "Conic"}, {"decz": -5.0 if mirror else 5.0}, {"eps": tensor()}, "end", {})])
    n = 96
    ang = rng.uniform(-0.9, 0.9, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    x0 = np.vstack((rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.full(n, -1.0)))
    k0 = np.vstack((np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        ref = oracle.trace(recs, x0, k0, e0)
    sysd = engine.DeviceSystem(recs, 0)
    res = sysd.trace(*[engine.to_device_rays(a, gpu_device, pitched=False) for a in (x0, k0, e0)])
    for s in range(4):
        xe = res.x_hit[s].cpu().numpy()
        ke = res.k_out[s].cpu().numpy() + 1j * res.k_out_im[s].cpu().numpy()
        (xr, kr) = (ref[s]["x_hit"], np.asarray(ref[s]["k_out"], dtype=complex))
        ok = np.all(np.isfinite(xr), axis=0) & np.all(np.abs(xr) < 1e6, axis=0)
        assert ok.sum() > 0.8 * ok.size
        assert np.array_equal(np.all(np.isfinite(xe), axis=0) | ~ok, np.ones(ok.size, bool))
        assert np.abs(xe[:, ok] - xr[:, ok]).max() < 1e-9
        ok2 = np.hstack((ok, ok)) if ke.shape[1] == 2 * ok.size else ok
        assert np.abs(ke[:, ok2] - kr[:, ok2]).max() < 1e-9


@pytest.mark.parametrize("name,mirror", [("aniso_absorbing_mirror", True), ("aniso_absorbing_two_crystals", False),
                                         ("aniso_absorbing_exit", None), ("absorbing_detector", None)])
def test_dropin_seqtrace_through_absorbing_crystals(name, mirror, gpu_device):
    """OpticalSystem.seqtrace of the mirror classes: the reference's bundle structure with complex k, bundle by
    bundle -- from the fused dense trace and from the plugin-granular loop"""
    from pyrate_amd import _lib
    api = zoo.mirror_api()
    case = _golden.load_case(name)
    if name == "aniso_absorbing_exit":
        (s, seq) = zoo.crystal_mirror(api, _eps_of(case, 1))
    elif name == "absorbing_detector":
        (s, seq) = zoo.absorbing_detector(api)
    else:
        (s, seq) = zoo.crystal_inside(api, _eps_of(case, 1), mirror=mirror, eps2=None if mirror else _eps_of(case, 2))
    ib = api.RayBundle(x0=case.x0, k0=case.k0, Efield0=case.E0, wave=case.wave)
    rp = s.seqtrace(ib, seq)
    assert len(rp) == 1 and len(rp[0].raybundles) == len(case.raw_bundles)
    for (i, (rb, ref)) in enumerate(zip(rp[0].raybundles, case.raw_bundles)):
        assert rb.x.shape == ref["x"].shape and rb.k.shape == ref["k"].shape, i
        assert np.array_equal(rb.rayID, ref["id"]) and np.array_equal(rb.valid, ref["valid"].astype(bool)), i
        assert np.abs(rb.x - ref["x"]).max() < 1e-10 * max(1.0, np.abs(ref["x"]).max())
        assert np.abs(rb.k - ref["k"]).max() < 1e-10
        if i >= (5 if name == "absorbing_detector" else 3):
            assert np.iscomplexobj(rb.k) and np.abs(np.imag(rb.k)).max() > 1e-3
    # the plugin-granular loop (Material.propagate / refract / reflect per surface, complex wave vectors through
    # prt_interact_cplx -- also for the detector, whose absorbing medium is an ISOTROPIC record, a one-record table
    # each time Material.refract is called on it): the same bundles
    rg = s._seqtrace_generic(ib, seq, False)
    assert len(rg) == 1 and len(rg[0].raybundles) == len(case.raw_bundles)
    for (i, (rb, ref)) in enumerate(zip(rg[0].raybundles, case.raw_bundles)):
        assert rb.x.shape == ref["x"].shape and rb.k.shape == ref["k"].shape, i
        assert np.array_equal(rb.rayID, ref["id"]) and np.array_equal(rb.valid, ref["valid"].astype(bool)), i
        assert np.abs(rb.x - ref["x"]).max() < 1e-10 * max(1.0, np.abs(ref["x"]).max()), i
        assert np.abs(rb.k - ref["k"]).max() < 1e-10, i
    assert np.iscomplexobj(rg[0].raybundles[-1].k) and np.abs(np.imag(rg[0].raybundles[-1].k)).max() > 1e-3
    # ... and that loop is where seqtrace sends what it cannot trace as a whole: a bundle that already carries an
    # invalid ray.  The rays are independent, so what comes out are the reference's bundles without that ray
    # (dropped by the compaction behind the first isotropic interface; crystal interfaces keep every ray)
    ib_bad = api.RayBundle(x0=case.x0, k0=case.k0, Efield0=case.E0, wave=case.wave)
    ib_bad._ensure()
    ib_bad._valid[-1][0] = 0
    ib_bad._valid.append(ib_bad._valid[-1].clone())
    ib_bad._x.append(ib_bad._x[-1])
    ib_bad._k.append(ib_bad._k[-1])
    ib_bad._e.append(ib_bad._e[-1])
    rb_bad = s.seqtrace(ib_bad, seq)
    assert len(rb_bad) == 1 and len(rb_bad[0].raybundles) == len(case.raw_bundles)
    for (i, (rb, ref)) in enumerate(zip(rb_bad[0].raybundles, case.raw_bundles)):
        if i < 2:
            continue                  # (the initial bundle itself, with its extra point)
        keep = ref["id"] != 0
        assert np.array_equal(rb.rayID, ref["id"][keep]), i
        assert np.abs(rb.x - ref["x"][:, :, keep]).max() < 1e-10 * max(1.0, np.abs(ref["x"]).max()), i
        assert np.abs(rb.k - ref["k"][:, :, keep]).max() < 1e-10, i


def test_dropin_splitup_through_absorbing_crystals_forks_eight_paths(gpu_device):
    """splitup=True: 2^3 RayPaths in the reference's order (existing paths take the first solution, the copies with
    the second one are appended, optical_element.py:360-375), N rays per bundle, complex k -- carved out of one
    dense trace"""
    import os
    api = zoo.mirror_api()
    case = _golden.load_case("aniso_absorbing_two_crystals_split")
    (s, seq) = zoo.crystal_inside(api, _eps_of(case, 1), mirror=False, eps2=_eps_of(case, 2))
    ib = api.RayBundle(x0=case.x0, k0=case.k0, Efield0=case.E0, wave=case.wave)
    rpaths = s.seqtrace(ib, seq, splitup=True)
    assert len(rpaths) == case.npaths == 8
    z = np.load(os.path.join(_golden.GOLDEN_DIR, case.name + ".npz"))
    kinds = set()
    for (pi, rp) in enumerate(rpaths):
        pre = "" if pi == 0 else "p%d_" % pi
        nb = int(z[pre + "nb"])
        assert len(rp.raybundles) == nb
        for i in range(nb):
            (rb, xr, kr) = (rp.raybundles[i], z[pre + "b%d_x" % i], z[pre + "b%d_k" % i])
            assert rb.x.shape == xr.shape and rb.k.shape == kr.shape, (pi, i)
            assert np.array_equal(rb.rayID, z[pre + "b%d_id" % i]) and np.array_equal(rb.valid, z[pre + "b%d_valid" % i].astype(bool))
            assert np.abs(rb.x - xr).max() < 1e-10 * max(1.0, np.abs(xr).max()), (pi, i)
            assert np.abs(rb.k - kr).max() < 1e-10, (pi, i)
        kinds.add(tuple(np.round(np.real(rp.raybundles[-1].k[-1][:, 0]), 9)))
    assert len(kinds) == 8            # eight different final wave vectors: the paths really are the eight branches


def test_absorbing_crystal_tables_the_library_refuses(gpu_device):
    """an isotropic medium behind an absorbing crystal BEFORE the last surface: no parity target (the reference's E
    there is an arbitrary null vector) -> UnsupportedError on the host, PRT_ERR_UNSUPPORTED from prt_system_create;
    k_out_im missing ->
    PRT_ERR_INVALID_ARG; prt_interact on such a table -> PRT_ERR_UNSUPPORTED"""
    import copy
    from pyrate_amd import engine, surface_table, _lib
    case = _golden.load_case("aniso_absorbing_mirror")
    bad = copy.deepcopy(case.table)
    bad[2]["material"] = {"type": "isotropic", "n": 1.0}
    with pytest.raises(surface_table.UnsupportedError):
        engine.DeviceSystem(bad, 0)
    lib = _lib.load()
    recs = [surface_table.pack_record(r) for r in bad]
    table = (surface_table.PrtSurface * len(recs))(*recs)
    h = ctypes.c_void_p()
    assert lib.prt_system_create(table, len(recs), 0, ctypes.byref(h)) == _lib.ERR_UNSUPPORTED
    assert b"behind the last surface only" in lib.prt_last_error()
    sysd = engine.DeviceSystem(case.table, 0)
    (x0, k0, e0) = _rays(case, gpu_device)
    bufs = sysd.alloc_outputs(case.x0.shape[1])
    a = sysd._trace_args(x0, k0, bufs, e0)
    a.k_out_im = None
    assert lib.prt_trace_ex(sysd._h, ctypes.byref(a)) == _lib.ERR_INVALID_ARG
    with pytest.raises(_lib.PrtError):
        sysd.interact(1, x0, k0)


@pytest.mark.parametrize("seed", range(8))
def test_hip_vs_oracle_random_sequences_that_end_in_an_isotropic_medium(seed, gpu_device):
    """the last surface of a table with absorbing media: exit from an absorbing crystal into air at steep incidence
    (rays dropped by the complex validity rule), lenses in front of absorbing detectors -- dielectric-like and
    metal-like indices (Re n^2 < 0: NumPy's order of complex numbers drops every ray) -- masks equal, complex k of
    the survivors equal to the LAPACK / NumPy oracle's"""
    from pyrate_amd import engine, systems
    rng = np.random.RandomState(500 + seed)
    n = 128
    ang = rng.uniform(-1.2, 1.2, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    x0 = np.vstack((rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.full(n, -1.0)))
    k0 = np.vstack((np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    if seed % 2 == 0:
        a = rng.uniform(1.3, 2.4, 3) ** 2
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        eps = q @ np.diag(a) @ q.T + 1j * (q @ np.diag(rng.uniform(0.0, 0.3, 3) * (10.0 ** -rng.randint(0, 4))) @ q.T)
        recs = systems.simple_system_records([
            ({"shape": "Conic"}, {"decz": 0.0}, None, "entry", {}),
            ({"shape": "Conic", "curv": 0.02}, {"decz": 4.0}, {"eps": eps}, "front", {}),
            ({"shape": "Conic", "curv": -0.05}, {"decz": 6.0}, {"eps": eps}, "rear", {"is_mirror": bool(seed % 4)}),
            ({"shape": "Conic", "curv": 0.03}, {"decz": -5.0 if seed % 4 else 5.0}, None, "exit", {})])
    else:
        n_abs = complex(rng.uniform(0.2, 4.0), rng.uniform(0.0, 3.0) * (10.0 ** -rng.randint(0, 3)))
        recs = systems.simple_system_records([
            ({"shape": "Conic"}, {"decz": 0.0}, None, "entry", {}),
            ({"shape": "Conic", "curv": 0.02}, {"decz": 4.0}, 1.5168, "front", {}),
            ({"shape": "Conic", "curv": -0.03}, {"decz": 4.0}, None, "back", {}),
            ({"shape": "Conic", "curv": -0.01}, {"decz": 10.0}, 1.0, "detector", {})])
        recs[-1]["material"] = {"type": "isotropic", "n": n_abs.real, "n_im": n_abs.imag}
    with np.errstate(all="ignore"):
        ref = oracle.trace(recs, x0, k0, e0)
    sysd = engine.DeviceSystem(recs, 0)
    assert sysd.complex_eps
    res = sysd.trace(*[engine.to_device_rays(a, gpu_device, pitched=False) for a in (x0, k0, e0)])
    (vo, vr) = (res.valid_out[-1].cpu().numpy().astype(bool), ref[-1]["valid_out"])
    xr = ref[-1]["x_hit"]
    sane = np.all(np.isfinite(xr), axis=0) & np.all(np.abs(xr) < 1e6, axis=0)
    assert np.array_equal(vo[sane], vr[sane])
    keep = sane & vr
    ke = res.k_out[-1].cpu().numpy() + 1j * res.k_out_im[-1].cpu().numpy()
    if keep.any():
        assert np.abs(ke[:, keep] - np.asarray(ref[-1]["k_out"], dtype=complex)[:, keep]).max() < 1e-9
        assert np.abs(res.x_hit[-1].cpu().numpy()[:, keep] - xr[:, keep]).max() < 1e-9
    if seed % 2 == 1 and (n_abs * n_abs).real < 0:
        assert not vr.any()              # metal-like: the reference's validity rule drops every ray


def test_interact_cplx_is_interact_where_nothing_is_complex_and_says_what_it_refuses(gpu_device):
    """prt_interact_cplx on LOSSLESS tables with a real incoming k: the wave vectors, ray directions and fields of
    prt_interact (crystal: both solutions; isotropic: k and mask), imaginary parts zero away from evanescent modes;
    null pointers / a mirror inside an absorbing isotropic medium are refused with the codes the header names"""
    from pyrate_amd import engine, surface_table, _lib
    case = _golden.load_case("aniso_doublet_uniaxial")
    sysd = engine.DeviceSystem(case.table, 0)
    assert not sysd.complex_eps
    (x0, k0, e0) = _rays(case, gpu_device)
    res = sysd.trace(x0, k0, e0)
    crystal = [s for (s, r) in enumerate(case.table) if r["material"]["type"] == "anisotropic"][0]
    n_in = case.x0.shape[1]
    xh = res.x_hit[crystal][:, :n_in].contiguous()
    kin = (res.k_out[crystal - 1] if crystal else k0)[:, :n_in].contiguous()
    (k_a, d_a, v_a, er_a, ei_a) = sysd.interact(crystal, xh, kin, want_e=True)
    (k_b, kim_b, d_b, v_b, er_b, ei_b) = sysd.interact_cplx(crystal, xh, kin, None, want_e=True)
    real_modes = (kim_b.abs().sum(dim=0) < 1e-12)           # (the complex solver leaves rounding-size imaginary parts)
    assert int(real_modes.sum()) > 0.9 * real_modes.numel()
    assert float((k_a - k_b)[:, real_modes].abs().max()) < 1e-12
    assert float((d_a - d_b)[:, real_modes].abs().max()) < 1e-10
    assert torch.equal(v_a, v_b)
    # E is defined up to a sign / phase per solver: compare the projectors E E^H
    Ea = (er_a + 1j * ei_a)[:, real_modes]
    Eb = (er_b + 1j * ei_b)[:, real_modes]
    overlap = (Ea.conj() * Eb).sum(dim=0).abs() / (Ea.abs().pow(2).sum(dim=0).sqrt() * Eb.abs().pow(2).sum(dim=0).sqrt())
    assert float((overlap - 1).abs().max()) < 1e-9
    iso = [s for (s, r) in enumerate(case.table) if r["material"]["type"] == "isotropic" and s > 0][-1]
    xi = res.x_hit[iso]
    ki = res.k_out[iso - 1]
    vin = res.valid[iso]
    (k_c, _, v_c, _, _) = sysd.interact(iso, xi, ki, valid_in=vin)
    (k_d, kim_d, dir_d, v_d, _, _) = sysd.interact_cplx(iso, xi, ki, None, valid_in=vin)
    assert dir_d is None and torch.equal(v_c, v_d)
    ok = v_c.bool()
    assert float((k_c - k_d)[:, ok].abs().max()) < 1e-12 and float(kim_d[:, ok].abs().max()) < 1e-12
    # refusals
    lib = _lib.load()
    n = xi.shape[1]
    buf = torch.empty((3, n), dtype=torch.float64, device=gpu_device)
    args = [ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in (xi.contiguous(), ki.contiguous())]
    rc = lib.prt_interact_cplx(sysd._h, iso, n, args[0], args[1], None, None, ctypes.c_void_p(buf.data_ptr()), None, None, None,
                               None, None, None)
    assert rc == _lib.ERR_INVALID_ARG                      # k_out_im is required
    rc = lib.prt_interact_cplx(sysd._h, crystal, n_in, ctypes.c_void_p(xh.data_ptr()), ctypes.c_void_p(kin.data_ptr()), None, None,
                               ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(buf.data_ptr()), None, None, None, None, None)
    assert rc == _lib.ERR_INVALID_ARG                      # a crystal interface needs dir_out
    assert lib.prt_interact_cplx(sysd._h, 99, n, args[0], args[1], None, None, None, None, None, None, None, None, None) \
        == _lib.ERR_INVALID_ARG


def test_system_layout_is_the_librarys_decision(gpu_device):
    """prt_system_layout: row-pitched for isotropic tables, concatenated with a ray pitch where the fused crystal walk
    runs, tight where the per-surface march does (absorbing media here) -- what DeviceSystem.alloc_outputs asks instead
    of re-deriving it"""
    from pyrate_amd import engine, systems, _lib
    lib = _lib.load()
    iso = engine.DeviceSystem(systems.double_gauss_records(), 0)
    assert lib.prt_system_layout(iso._h) == _lib.LAYOUT_ROW_PITCHED
    crystal = engine.DeviceSystem(_golden.load_case("aniso_doublet_uniaxial").table, 0)
    assert lib.prt_system_layout(crystal._h) == _lib.LAYOUT_CONCATENATED_PITCHED
    assert crystal.alloc_outputs(1000)["pitch"] == lib.prt_crystal_pitch(1000) >= 1000
    absorbing = engine.DeviceSystem(_golden.load_case("aniso_absorbing_mirror").table, 0)
    assert lib.prt_system_layout(absorbing._h) == _lib.LAYOUT_CONCATENATED_TIGHT
    assert absorbing.alloc_outputs(1000)["pitch"] == 0
    assert lib.prt_system_layout(None) == _lib.ERR_INVALID_ARG
