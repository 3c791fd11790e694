"""one-off fuzz of the analysis / bookkeeping kernels against NumPy: prt_compact (arrays + ids + flags),
prt_bundle_moments (3 modes, masks, pitched views), prt_path_sums (both modes), prt_poynting_dir"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from pyrate_amd import engine
dev = torch.device("cuda", 0)
bad = []; ncase = 0
for seed in range(300):
    rng = np.random.RandomState(91000 + seed)
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 1023, 1024, 1025, 4097, 20000]))
    try:
        # ---- compact
        mask = (rng.rand(n) < rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])).astype(np.uint8)
        arrs = [rng.rand(3, n), rng.rand(int(rng.randint(1, 4)), n)]
        ids = rng.randint(0, 10 ** 9, n).astype(np.int64)
        flags = (rng.rand(n) < 0.5).astype(np.uint8)
        pitched = rng.rand() < 0.5
        d_arrs = [engine.to_device_rays(a, dev, pitched=pitched and a.shape[0] == 3) for a in arrs]
        (out, idc, flc) = engine.compact(torch.from_numpy(mask).to(dev), d_arrs, torch.from_numpy(ids).to(dev),
                                         torch.from_numpy(flags).to(dev))
        m = mask.astype(bool)
        for (o, a) in zip(out, arrs):
            assert np.array_equal(o.cpu().numpy(), a[:, m]), "compact arrays"
        assert np.array_equal(idc.cpu().numpy(), ids[m]) and np.array_equal(flc.cpu().numpy(), flags[m]), "compact ids/flags"
        # ---- moments
        x = rng.uniform(-5, 5, (3, n)); ref = rng.uniform(-1, 1, 3)
        xd = engine.to_device_rays(x, dev, pitched=pitched)
        md = torch.from_numpy(mask).to(dev) if rng.rand() < 0.7 else None
        mm = m if md is not None else np.ones(n, bool)
        (cnt, s1, s2) = engine.bundle_moments(xd, md, ref, 0)
        v = x[:, mm] - ref[:, None]
        assert cnt == mm.sum() and np.allclose(s1, v.sum(axis=1), rtol=1e-12, atol=1e-10) and np.allclose(s2, (v ** 2).sum(axis=1), rtol=1e-12, atol=1e-10), "moments 0"
        u = x[:, mm] / np.sqrt(np.sum(x[:, mm] ** 2, axis=0))
        (cnt, s1, s2) = engine.bundle_moments(xd, md, None, 1)
        assert np.allclose(s1, u.sum(axis=1), rtol=1e-12, atol=1e-10), "moments 1"
        (cnt, s1, s2) = engine.bundle_moments(xd, md, ref, 2)
        c = np.cross(u, ref, axisa=0, axisb=0).T
        assert np.allclose(s2, (c ** 2).sum(axis=1), rtol=1e-12, atol=1e-10), "moments 2"
        # ---- path sums
        P = int(rng.randint(2, 6))
        xs = [rng.uniform(-3, 3, (3, n)) for _ in range(P)]; ks = [rng.uniform(-2, 2, (3, n)) for _ in range(P)]
        dx = [engine.to_device_rays(a, dev, pitched=False) for a in xs]; dk = [engine.to_device_rays(a, dev, pitched=False) for a in ks]
        arc = engine.path_sums(dx, mode=0).cpu().numpy()
        assert np.allclose(arc, sum(np.sqrt(np.sum((xs[p + 1] - xs[p]) ** 2, axis=0)) for p in range(P - 1)), rtol=1e-13), "arc"
        ph = engine.path_sums(dx, dk, mode=1).cpu().numpy()
        assert np.allclose(ph, sum(np.sum(xs[p + 1] * ks[p + 1] - xs[p] * ks[p], axis=0) for p in range(P - 1)), rtol=1e-11, atol=1e-11), "phase"
        # ---- Poynting direction with complex E
        k = rng.uniform(-1, 1, (3, n)) + np.array([[0.], [0.], [2.]]); er = rng.uniform(-1, 1, (3, n)); ei = rng.uniform(-1, 1, (3, n))
        d = engine.poynting_dir(*[engine.to_device_rays(a, dev, pitched=False) for a in (k, er, ei)]).cpu().numpy()
        E = er + 1j * ei
        S = np.real(np.sum(np.conj(E) * E, axis=0) * k - np.sum(E * k, axis=0) * np.conj(E))
        assert np.allclose(d, S / np.sqrt(np.sum(S ** 2, axis=0)), rtol=0, atol=1e-13), "poynting"
        ncase += 1
    except AssertionError as exc:
        bad.append((seed, n, str(exc)[:120]))
    except Exception as exc:
        bad.append((seed, n, "exception " + repr(exc)[:200]))
print("cases:", ncase, " failures:", len(bad))
for b in bad[:20]: print(b)
