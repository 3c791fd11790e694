"""harsher one-off fuzz: strong curvatures (misses, total internal reflection, NaN domains), large
tilts, wide bundles; HIP vs oracle on every ray incl. masks"""
import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import _golden
from oracle import seqtrace_np as oracle
from pyrate_amd import engine
import test_gpu_fuzz as tf
dev = torch.device("cuda", 0)
_orig_shape = tf.random_shape

def harsh_shape(rng, kind):
    c = rng.uniform(-1, 1) / rng.uniform(6, 40)
    if kind == 0:
        return {"type": "conic", "curv": c, "cc": rng.choice([0.0, rng.uniform(-3, 3)])}
    if kind == 1:
        return {"type": "asphere", "curv": c, "cc": rng.uniform(-2.5, 1.5),
                "coeffs": [rng.uniform(-1, 1) * 1e-3, rng.uniform(-1, 1) * 1e-5, rng.uniform(-1, 1) * 1e-8]}
    return _orig_shape(rng, kind)

bad = []
ntot = 0
for seed in range(1000):
    rng = np.random.RandomState(7000 + seed)
    orig = tf.random_shape
    tf.random_shape = harsh_shape
    try:
        recs = tf.random_table(rng, int(rng.randint(3, 9)), seed % 2 == 1, seed % 3 != 0, seed % 4 == 3)
    finally:
        tf.random_shape = orig
    n = 777
    x0 = np.vstack((rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-0.35, 0.35, n), rng.uniform(-0.35, 0.35, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    res = engine.DeviceSystem(recs, 0).trace(engine.to_device_rays(x0, dev), engine.to_device_rays(k0, dev),
                                             engine.to_device_rays(e0, dev))
    for s in range(len(recs)):
        vo = out[s]["valid"]; wo = out[s]["valid_out"]
        vd = res.valid[s].cpu().numpy().astype(bool); wd = res.valid_out[s].cpu().numpy().astype(bool)
        if not np.array_equal(vd, vo):
            bad.append((seed, s, "valid", int(np.sum(vd != vo)))); break
        if not np.array_equal(wd, wo):
            bad.append((seed, s, "valid_out", int(np.sum(wd != wo)))); break
        xo = out[s]["x_hit"][:, vo]
        if xo.shape[1]:
            xd = res.x_hit[s].cpu().numpy()[:, vo]
            fin = np.all(np.isfinite(xo), axis=0)
            if not np.array_equal(fin, np.all(np.isfinite(xd), axis=0)):
                bad.append((seed, s, "finite(x)", int(np.sum(fin != np.all(np.isfinite(xd), axis=0))))); break
            err = (np.abs(xd[:, fin] - xo[:, fin]) / _golden.relative_scale(xo[:, fin])).max() if fin.any() else 0.0
            if not err < 1e-9:
                bad.append((seed, s, "x", float(err))); break
        ko = out[s]["k_out"][:, wo]
        if ko.shape[1]:
            ek = np.abs(res.k_out[s].cpu().numpy()[:, wo] - ko).max()
            if not ek < 1e-9:
                bad.append((seed, s, "k", float(ek))); break
        ntot += int(wo.sum())
print("compared ray-surfaces:", ntot, " failures:", len(bad))
for b in bad[:30]: print(b)
