"""one-off fuzz: splitup=True (one RayPath per branch, forked on the GPU by Material.refract) vs the
doubled bundles of splitup=False: path L of the fork == slots [L n, (L+1) n) of the stacked bundle"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import systems_zoo as zoo
import test_gpu_fuzz as tf
api = zoo.mirror_api()
bad = []; npaths = 0
for seed in range(120):
    rng = np.random.RandomState(81000 + seed)
    s = api.OpticalSystem.p()
    lc_prev = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=0.0), refname=s.rootcoordinatesystem.name)
    elem = api.OpticalElement.p(lc_prev, name="e")
    ncry = int(rng.randint(1, 4))
    seq = []; last = None
    for j in range(ncry + 2):
        lc = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="s%d" % j, decz=float(rng.uniform(3, 7)),
                                                               tiltx=float(rng.uniform(-0.1, 0.1))), refname=lc_prev.name)
        if j < ncry:
            R = tf.rot(rng, 1.2)
            pv = np.sort(rng.uniform(1.4, 1.9, 3)) ** 2 if rng.rand() < 0.5 else np.array([2.2, 2.2, rng.uniform(2.4, 3.2)])
            mat = "c%d" % j
            elem.addMaterial(mat, api.AnisotropicMaterial.p(lc, R.dot(np.diag(pv)).dot(R.T)))
        elif j == ncry:
            mat = "g"
            elem.addMaterial(mat, api.ConstantIndexGlass.p(lc, 1.5))
        else:
            mat = None
        elem.addSurface("s%d" % j, api.Surface.p(lc, shape=api.Conic.p(lc, curv=float(rng.uniform(-0.02, 0.02)))), (last, mat))
        seq.append(("s%d" % j, {}))
        last = mat; lc_prev = lc
    s.addElement("e", elem)
    seq = [("e", seq)]
    n = int(rng.choice([1, 9, 50]))
    x0 = np.vstack((rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.full(n, -1.0)))
    u = np.vstack((rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    try:
        stacked = s.seqtrace(api.RayBundle(x0, k0, e0, wave=0.55e-3), seq)[0]
        forks = s.seqtrace(api.RayBundle(x0, k0, e0, wave=0.55e-3), seq, splitup=True)
        assert len(forks) == 2 ** ncry, len(forks)
        xs = stacked.raybundles[-1].x[-1]; ks = np.real(stacked.raybundles[-1].k[-1])
        assert xs.shape[1] == n * 2 ** ncry
        for (L, rp) in enumerate(forks):
            xf = rp.raybundles[-1].x[-1]; kf = np.real(rp.raybundles[-1].k[-1])
            assert xf.shape[1] == n and np.array_equal(rp.raybundles[-1].rayID, np.arange(n))
            assert np.allclose(xf, xs[:, L * n:(L + 1) * n], rtol=0, atol=1e-10, equal_nan=True), (L, "x")
            assert np.allclose(kf, ks[:, L * n:(L + 1) * n], rtol=0, atol=1e-11, equal_nan=True), (L, "k")
            npaths += 1
    except AssertionError as exc:
        bad.append((seed, "assert", str(exc)[:200]))
    except Exception as exc:
        bad.append((seed, "exception", repr(exc)[:200]))
print("forked paths compared:", npaths, " failures:", len(bad))
for b in bad[:25]: print(b)
