"""Randomised parity: HIP engine vs the (reference-pinned) NumPy oracle on seeded random
systems -- random conic / asphere / biconic / XY shapes, tilted and decentred frames, apertures,
indices, mirrors -- every ray, every surface, masks included."""
import math

import numpy as np
import pytest

import _golden
from oracle import seqtrace_np as oracle

pytestmark = pytest.mark.gpu


def rot(rng, amp):
    (a, b, c) = rng.uniform(-amp, amp, 3)
    rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return rz.dot(ry).dot(rx)


def random_shape(rng, kind):
    c = rng.uniform(-1, 1) / rng.uniform(25, 120)
    if kind == 0:
        return {"type": "conic", "curv": c, "cc": rng.choice([0.0, rng.uniform(-2, 1.5)])}
    if kind == 1:
        return {"type": "asphere", "curv": c, "cc": rng.uniform(-1.5, 0.5),
                "coeffs": [rng.uniform(-1, 1) * 1e-4, rng.uniform(-1, 1) * 1e-7, rng.uniform(-1, 1) * 1e-10]}
    if kind == 2:
        return {"type": "biconic", "curvx": c, "curvy": c * rng.uniform(0.5, 1.5), "ccx": rng.uniform(-1, 0.5),
                "ccy": rng.uniform(-1, 0.5), "coeffs": [[rng.uniform(-1, 1) * 1e-5, rng.uniform(-0.5, 0.5)]]}
    if kind == 4:
        nterm = int(rng.randint(4, 26))
        return {"type": "zernike", "indexing": "fringe" if rng.rand() < 0.5 else "ansi",
                "normradius": float(rng.uniform(8.0, 12.0)),
                "coeffs": [float(rng.uniform(-1, 1) * 0.03 / (1 + j)) for j in range(nterm)]}
    if kind == 5:
        return {"type": "combination", "parts": [
            {"coefficient": float(rng.uniform(0.7, 1.2)), "offset": [0.0, 0.0, 0.0],
             "shape": {"type": "asphere", "curv": c, "cc": rng.uniform(-1.2, 0.3),
                       "coeffs": [0.0, rng.uniform(-1, 1) * 1e-6]}},
            {"coefficient": float(rng.uniform(0.5, 1.5)),
             "offset": [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(-0.05, 0.05))],
             "shape": {"type": "zernike", "indexing": "fringe", "normradius": 10.0,
                       "coeffs": [float(rng.uniform(-1, 1) * 0.02 / (1 + j)) for j in range(9)]}},
            {"coefficient": 1.0, "offset": [float(rng.uniform(-0.5, 0.5)), 0.0, 0.0],
             "shape": {"type": "xypoly", "normradius": 10.0,
                       "terms": [[2, 1, rng.uniform(-0.01, 0.01)], [0, 3, rng.uniform(-0.01, 0.01)]]}}]}
    return {"type": "xypoly", "normradius": 10.0,
            "terms": [[2, 0, rng.uniform(-0.1, 0.1)], [0, 2, rng.uniform(-0.1, 0.1)], [1, 1, rng.uniform(-0.02, 0.02)],
                      [3, 0, rng.uniform(-0.01, 0.01)], [2, 2, rng.uniform(-0.005, 0.005)]]}


def random_table(rng, n_surf, tilted, explicit, mirrors):
    recs = []
    z = 0.0
    n_cur = 1.0
    for s in range(n_surf):
        z += rng.uniform(3.0, 12.0)
        kind = int(rng.randint(0, 6)) if explicit else 0
        Bs = rot(rng, 0.08) if tilted else np.eye(3)
        g = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), z]) if tilted else np.array([0., 0., z])
        own_ap_frame = tilted and rng.rand() < 0.5
        Ba = rot(rng, 0.3) if own_ap_frame else Bs
        ga = g + (np.array([0.2, -0.1, 0.0]) if own_ap_frame else 0.0)
        apk = int(rng.randint(0, 3))
        ap = [{"type": "none"}, {"type": "circular", "minradius": 0.0 if rng.rand() < 0.7 else 0.5,
                                 "maxradius": rng.uniform(5.0, 9.0)},
              {"type": "rectangular", "width": rng.uniform(9, 16), "height": rng.uniform(9, 16)}][apk]
        mirror = mirrors and (s in (1, 2)) and kind == 0
        if not mirror:
            n_cur = 1.0 if (s % 2 == 1) else rng.uniform(1.4, 1.9)
        recs.append({"shape": random_shape(rng, kind), "B_shape": Bs.tolist(), "g_shape": g.tolist(),
                     "aperture": ap, "B_ap": np.asarray(Ba).tolist(), "g_ap": np.asarray(ga).tolist(),
                     "interaction": "mirror" if mirror else "refract",
                     "material": {"type": "isotropic", "n": float(n_cur)},
                     "B_mat": (rot(rng, 0.5) if tilted else np.eye(3)).tolist()})
        if mirror:
            z -= rng.uniform(12.0, 20.0)      # the next surface sits behind the mirror
    return recs


@pytest.mark.parametrize("seed", range(24))
def test_random_systems_match_oracle(gpu_device, seed):
    from pyrate_amd import engine
    rng = np.random.RandomState(1000 + seed)
    tilted = seed % 2 == 1
    explicit = seed % 3 != 0
    mirrors = seed % 4 == 3
    recs = random_table(rng, int(rng.randint(3, 8)), tilted, explicit, mirrors)
    n = int(rng.choice([257, 1000, 1535]))
    x0 = np.vstack((rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-0.12, 0.12, n), rng.uniform(-0.12, 0.12, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy() + 1j * 0.2 * np.cross(
        k0, np.array([0., 1., 0.2]), axisa=0, axisb=0).T
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    sysd = engine.DeviceSystem(recs, 0)
    res = sysd.trace(engine.to_device_rays(x0, gpu_device), engine.to_device_rays(k0, gpu_device),
                     engine.to_device_rays(e0.real, gpu_device), engine.to_device_rays(e0.imag, gpu_device))
    ncmp = 0
    for s in range(len(recs)):
        vo = out[s]["valid"]
        assert np.array_equal(res.valid[s].cpu().numpy().astype(bool), vo), (seed, s)
        wo = out[s]["valid_out"]
        assert np.array_equal(res.valid_out[s].cpu().numpy().astype(bool), wo), (seed, s)
        xo = out[s]["x_hit"][:, vo]
        if xo.shape[1]:
            err = np.abs(res.x_hit[s].cpu().numpy()[:, vo] - xo) / _golden.relative_scale(xo)
            assert err.max() < 1e-10, (seed, s, err.max())
        ko = out[s]["k_out"][:, wo]
        if ko.shape[1]:
            assert np.abs(res.k_out[s].cpu().numpy()[:, wo] - ko).max() < 1e-10, (seed, s)
        ncmp += int(wo.sum())
    assert ncmp > 0


@pytest.mark.parametrize("seed", range(12))
def test_random_crystals_match_oracle(gpu_device, seed):
    """random real symmetric epsilon tensors (isotropic / uniaxial / biaxial, arbitrarily
    rotated) in the doublet: wave vectors of all split rays vs the oracle (scipy.linalg.eig
    like the reference), order of the two transmitted solutions included"""
    from pyrate_amd import engine, systems
    rng = np.random.RandomState(500 + seed)

    def eps():
        kind = seed % 3
        R = rot(rng, 1.0)
        if kind == 0:
            pv = np.full(3, rng.uniform(1.4, 1.8) ** 2)
        elif kind == 1:
            (no, ne) = (rng.uniform(1.45, 1.7), rng.uniform(1.45, 1.7))
            pv = np.array([no ** 2, no ** 2, ne ** 2])
        else:
            pv = np.sort(rng.uniform(1.45, 1.75, 3)) ** 2
        return R.dot(np.diag(pv)).dot(R.T)
    recs = systems.aniso_doublet_records(eps(), eps())
    (o, k, e0) = systems.double_gauss_bundle(300, rpup=10.0, z0=-5.0, field_deg=float(rng.uniform(-4, 4)))
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, o, k, e0)
    sysd = engine.DeviceSystem(recs, 0)
    res = sysd.trace(engine.to_device_rays(o, gpu_device), engine.to_device_rays(k, gpu_device),
                     engine.to_device_rays(e0, gpu_device))
    degenerate = seed % 3 == 0
    for s in range(len(recs)):
        xo = out[s]["x_hit"]
        v = out[s]["valid"] & np.all(np.isfinite(xo), axis=0)
        xd = res.x_hit[s].cpu().numpy()
        kd = res.k_out[s].cpu().numpy()
        ko = np.real(out[s]["k_out"])
        fin = np.all(np.isfinite(ko), axis=0)
        if degenerate:
            # eps = e I: both transmitted solutions share k; hit points agree whatever the order
            assert np.abs(xd[:, v] - xo[:, v]).max() < 1e-9
            assert np.abs(kd[:, fin] - ko[:, fin]).max() < 1e-9
        else:
            assert np.abs(xd[:, v] - xo[:, v]).max() < 1e-9, (seed, s)
            assert np.abs(kd[:, fin] - ko[:, fin]).max() < 1e-10, (seed, s)


def test_many_crystal_interfaces_use_the_per_surface_march(gpu_device):
    """five crystal slabs in a row (32 leaves per input ray): beyond four anisotropic interfaces
    prt_trace switches from the fused kernel to the per-surface march; both against the oracle,
    and the two engine paths against each other on a three-slab stack"""
    from pyrate_amd import engine, systems
    rng = np.random.RandomState(77)

    def eps():
        R = rot(rng, 1.0)
        (no, ne) = (rng.uniform(1.5, 1.7), rng.uniform(1.5, 1.7))
        return R.dot(np.diag([no ** 2, no ** 2, ne ** 2])).dot(R.T)

    def stack(m):
        bl = [({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True})]
        for j in range(m):
            bl.append(({"shape": "Conic", "curv": (0.004 if j % 2 else -0.003)}, {"decz": 4.0},
                       {"eps": eps()}, "slab%d" % j, {}))
        bl.append(({"shape": "Conic", "curv": 0.002}, {"decz": 4.0}, None, "exit", {}))
        bl.append(({"shape": "Conic"}, {"decz": 30.0}, None, "image", {}))
        return systems.simple_system_records(bl)

    (o, k, e0) = systems.double_gauss_bundle(40, rpup=3.0, z0=-5.0, field_deg=2.0)
    dev_rays = [engine.to_device_rays(a, gpu_device, pitched=False) for a in (o, k, e0)]
    for m in (3, 5):
        recs = stack(m)
        with np.errstate(all="ignore"):
            out = oracle.trace(recs, o, k, e0)
        res = engine.DeviceSystem(recs, 0).trace(*dev_rays)
        assert res.x_hit[-1].shape[1] == o.shape[1] * 2 ** m
        for s in range(len(recs)):
            v = out[s]["valid"] & np.all(np.isfinite(out[s]["x_hit"]), axis=0)
            assert np.array_equal(res.valid[s].cpu().numpy().astype(bool)[v], out[s]["valid"][v])
            assert np.abs(res.x_hit[s].cpu().numpy()[:, v] - out[s]["x_hit"][:, v]).max() < 1e-9, (m, s)
            ko = np.real(out[s]["k_out"])
            fin = np.all(np.isfinite(ko), axis=0)
            assert np.abs(res.k_out[s].cpu().numpy()[:, fin] - ko[:, fin]).max() < 1e-10, (m, s)


def test_partially_evanescent_crystal_interface_keeps_the_propagating_mode_in_its_slot(gpu_device):
    """dense glass -> strongly birefringent crystal at steep incidence: one of the two transmitted
    modes is evanescent (complex xi; NaN in the engine), the other propagates.  The reference orders the
    modes by S.n, where an evanescent mode has S.n ~ 0, i.e. sits between the backward and the forward
    propagating ones -- so the propagating transmitted mode is the SECOND of the pair [sol2, sol3]."""
    from pyrate_amd import engine, systems
    eps = systems.uniaxial_eps(1.35, 2.1, (1.0, 0.0, 0.0))
    recs = systems.simple_system_records([
        ({"shape": "Conic"}, {"decz": 0.0}, 1.9, "entry", {}),
        ({"shape": "Conic", "curv": 0.0}, {"decz": 5.0}, {"eps": eps}, "crystal", {}),
        ({"shape": "Conic"}, {"decz": 5.0}, None, "exit", {})], background_n=1.9)
    n = 64
    ang = np.linspace(0.2, 1.1, n)                         # 11 .. 63 degrees inside the glass
    x0 = np.vstack((np.zeros(n), np.zeros(n), np.full(n, -1.0)))
    k0 = 1.9 * np.vstack((np.sin(ang) * 0.6, np.sin(ang) * 0.8, np.cos(ang)))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    res = engine.DeviceSystem(recs, 0).trace(*[engine.to_device_rays(a, gpu_device, pitched=False)
                                               for a in (x0, k0, e0)])
    ko = out[1]["k_out"]
    kd = res.k_out[1].cpu().numpy()
    real_mode = np.all(np.abs(np.imag(ko)) < 1e-12, axis=0) & np.all(np.isfinite(np.real(ko)), axis=0)
    evan = ~real_mode
    # the scenario is there: rays with exactly one propagating transmitted mode, and it is slot [n:]
    one_mode = evan[:n] & real_mode[n:]
    assert one_mode.sum() > 5 and not (real_mode[:n] & evan[n:]).any()
    assert np.abs(kd[:, real_mode] - np.real(ko)[:, real_mode]).max() < 1e-12
    assert np.all(np.isnan(kd[:, evan]))
    # the same against the reference's own bundle (golden case of the same slab)
    import systems_zoo as zoo
    case = _golden.load_case("aniso_partial_evanescent")
    (xg, kg, eg) = zoo.evanescent_bundle_arrays()
    rg = engine.DeviceSystem(case.table, 0).trace(*[engine.to_device_rays(a, gpu_device, pitched=False)
                                                    for a in (xg, kg, eg)])
    kref = case.raw_bundles[3]["k"][0]
    real_ref = np.all(np.abs(np.imag(kref)) < 1e-12, axis=0)
    kdg = rg.k_out[1].cpu().numpy()
    assert np.abs(kdg[:, real_ref] - np.real(kref)[:, real_ref]).max() < 1e-12 and np.all(np.isnan(kdg[:, ~real_ref]))
    # ... and those rays arrive at the exit face where the reference's arrive
    xe = res.x_hit[2].cpu().numpy()
    assert np.abs(xe[:, real_mode] - out[2]["x_hit"][:, real_mode]).max() < 1e-11


# ---------------------------------------------------------------------------------------------
# extreme systems (a slice of the one-off stress fuzzers under scratch/ that found two defects)
# ---------------------------------------------------------------------------------------------
_plain_random_shape = random_shape


def harsh_shape(rng, kind):
    c = rng.uniform(-1, 1) / rng.uniform(6, 40)
    if kind == 0:
        return {"type": "conic", "curv": c, "cc": rng.choice([0.0, rng.uniform(-3, 3)])}
    if kind == 1:
        return {"type": "asphere", "curv": c, "cc": rng.uniform(-2.5, 1.5),
                "coeffs": [rng.uniform(-1, 1) * 1e-3, rng.uniform(-1, 1) * 1e-5, rng.uniform(-1, 1) * 1e-8]}
    return _plain_random_shape(rng, kind)


def compare_with_oracle(recs, x0, k0, e0, device, tight=False, tol=1e-9, newton_slack=3):
    """HIP vs oracle on every surface; rays behind an evanescent crystal mode (complex k in the oracle,
    NaN in the engine) and the few rays whose Newton iteration ends at the cap on one side only are
    excluded from then on.  Returns the number of compared wave vectors."""
    from pyrate_amd import engine
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    res = engine.DeviceSystem(recs, 0).trace(*[engine.to_device_rays(a, device, pitched=not tight)
                                               for a in (x0, k0, e0)])
    taint = np.zeros(x0.shape[1], dtype=bool)
    alive = np.ones(x0.shape[1], dtype=bool)       # rays the reference still carries (not compacted away)
    ncmp = 0
    for s in range(len(recs)):
        xo = out[s]["x_hit"]
        xd = res.x_hit[s].cpu().numpy()
        fin_o = np.all(np.isfinite(xo), axis=0)
        fin_d = np.all(np.isfinite(xd), axis=0)
        border = (fin_o != fin_d) & ~taint & alive
        assert border.sum() <= newton_slack, (s, "finite hit points differ", int(border.sum()))
        taint = taint | border
        v = out[s]["valid"] & fin_o & ~taint
        assert np.array_equal(res.valid[s].cpu().numpy().astype(bool)[~taint], out[s]["valid"][~taint]), (s, "valid")
        if v.any():
            assert (np.abs(xd[:, v] - xo[:, v]) / _golden.relative_scale(xo[:, v])).max() < tol, (s, "x")
        ko = np.real(out[s]["k_out"])
        kd = res.k_out[s].cpu().numpy()
        if ko.shape[1] == 2 * taint.shape[0]:
            taint = np.concatenate((taint, taint))
        alive = out[s]["valid_out"].astype(bool)
        taint = taint | ~np.all(np.abs(np.imag(out[s]["k_out"])) < 1e-12, axis=0)
        wd = res.valid_out[s].cpu().numpy().astype(bool)
        assert np.array_equal(wd[~taint], out[s]["valid_out"][~taint]), (s, "valid_out")
        fin = np.all(np.isfinite(ko), axis=0) & out[s]["valid_out"] & ~taint
        if fin.any():
            assert np.abs(kd[:, fin] - ko[:, fin]).max() < tol, (s, "k")
        ncmp += int(fin.sum())
    return ncmp


def wide_bundle(rng, n, r, ang):
    x0 = np.vstack((rng.uniform(-r, r, n), rng.uniform(-r, r, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-ang, ang, n), rng.uniform(-ang, ang, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    return (x0, k0, np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy())


@pytest.mark.parametrize("seed", range(24))
def test_extreme_systems_match_oracle(gpu_device, seed, monkeypatch):
    """strong curvatures (misses, total internal reflection, NaN domains of the explicit shapes),
    large tilts, bundles far wider than the apertures"""
    import test_gpu_fuzz as this
    rng = np.random.RandomState(7000 + seed)
    monkeypatch.setattr(this, "random_shape", harsh_shape)
    recs = random_table(rng, int(rng.randint(3, 9)), seed % 2 == 1, seed % 3 != 0, seed % 4 == 3)
    monkeypatch.undo()
    compare_with_oracle(recs, *wide_bundle(rng, 777, 9.0, 0.35), gpu_device)


@pytest.mark.parametrize("seed", range(16))
def test_extreme_crystal_stacks_match_oracle(gpu_device, seed):
    """1-3 crystal interfaces with strong birefringence (uniaxial / biaxial), tilted surface and
    material frames, mirrors INSIDE crystals, steep rays (partially evanescent interfaces)"""
    rng = np.random.RandomState(9000 + seed)

    def eps():
        R = rot(rng, 1.5)
        if rng.randint(1, 3) == 1:
            (no, ne) = (rng.uniform(1.3, 2.2), rng.uniform(1.3, 2.2))
            pv = np.array([no ** 2, no ** 2, ne ** 2])
        else:
            pv = np.sort(rng.uniform(1.3, 2.2, 3)) ** 2
        return R.dot(np.diag(pv)).dot(R.T)
    ncry = int(rng.randint(1, 4))
    tilted = seed % 2 == 0
    recs = []
    z = 0.0
    in_crystal = False
    for s in range(ncry + 2):
        z += rng.uniform(3.0, 8.0)
        Bs = rot(rng, 0.15) if tilted else np.eye(3)
        g = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), z]) if tilted else np.array([0., 0., z])
        mirror = in_crystal and rng.rand() < 0.3
        if s < ncry or mirror:
            e = eps() if not mirror else np.asarray(recs[-1]["material"]["eps_re"])
            mat = {"type": "anisotropic", "eps_re": e.tolist(), "eps_im": np.zeros((3, 3)).tolist()}
            in_crystal = True
        else:
            mat = {"type": "isotropic", "n": 1.0 if s == ncry + 1 else float(rng.uniform(1.0, 1.8))}
            in_crystal = False
        recs.append({"shape": {"type": "conic", "curv": rng.uniform(-1, 1) / rng.uniform(15, 80),
                               "cc": float(rng.choice([0.0, rng.uniform(-1.5, 1.0)]))},
                     "B_shape": Bs.tolist(), "g_shape": g.tolist(), "aperture": {"type": "none"},
                     "B_ap": Bs.tolist(), "g_ap": g.tolist(), "interaction": "mirror" if mirror else "refract",
                     "material": mat, "B_mat": (rot(rng, 0.8) if tilted else np.eye(3)).tolist()})
        if mirror:
            z -= rng.uniform(6.0, 14.0)
    assert compare_with_oracle(recs, *wide_bundle(rng, 200, 3.0, 0.3), gpu_device, tight=True, tol=1e-8) > 0
