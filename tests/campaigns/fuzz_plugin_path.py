"""one-off fuzz: random object graphs (mirror classes) traced by OpticalSystem.seqtrace (one engine
call + lazy bundles) and by the plugin-granular loop (Material.propagate / refract per surface):
same RayPath, bundle by bundle"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import systems_zoo as zoo
api = zoo.mirror_api()
bad = []; nb = 0
for seed in range(250):
    try:
        (s, seq) = zoo.random_object_graph(api, seed)
        rng = np.random.RandomState(71000 + seed)
        n = int(rng.choice([1, 7, 64, 200]))
        x0 = np.vstack((rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), np.full(n, -1.0)))
        u = np.vstack((rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), np.ones(n)))
        k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
        e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
        a = s.seqtrace(api.RayBundle(x0, k0, e0, wave=0.55e-3), seq)[0]
        b = s._seqtrace_generic(api.RayBundle(x0, k0, e0, wave=0.55e-3), seq, False)[0]
        assert len(a.raybundles) == len(b.raybundles)
        for (j, (ba, bb)) in enumerate(zip(a.raybundles, b.raybundles)):
            assert ba.x.shape == bb.x.shape, (j, ba.x.shape, bb.x.shape)
            assert np.array_equal(ba.rayID, bb.rayID), j
            assert np.array_equal(ba.valid, bb.valid), (j, "valid")
            assert np.allclose(ba.x, bb.x, rtol=0, atol=1e-9, equal_nan=True), (j, "x", float(np.nanmax(np.abs(ba.x - bb.x))))
            assert np.allclose(np.real(ba.k), np.real(bb.k), rtol=0, atol=1e-10, equal_nan=True), (j, "k")
            nb += 1
    except AssertionError as exc:
        bad.append((seed, "assert", str(exc)[:200]))
    except Exception as exc:
        bad.append((seed, "exception", repr(exc)[:200]))
print("bundles compared:", nb, " failures:", len(bad))
for b in bad[:25]: print(b)
