"""one-off fuzz for free-form prism faces: biconics with the LARGE b_n of head-up-display designs (demos/demo_hud.py:
b up to 29), two or three coefficient pairs, hit points 20-40 mm from the axis where (r^2 - b (x^2 - y^2))^n cancels
and the Newton steps stall at the rounding noise of the evaluation; refraction and (total internal) reflection at the
same face; tilted frames.  HIP vs oracle on every ray incl. masks."""
import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import _golden
from oracle import seqtrace_np as oracle
from pyrate_amd import engine
import test_gpu_fuzz as tf
dev = torch.device("cuda", 0)


def face(rng):
    r = rng.uniform(50, 140) * rng.choice([-1, 1])
    pairs = [[0.0, 0.0], [rng.uniform(-1, 1) * 10 ** rng.uniform(-11, -6.5), rng.uniform(-30, 30)]]
    if rng.rand() < 0.6:
        pairs.append([rng.uniform(-1, 1) * 10 ** rng.uniform(-13, -10), rng.uniform(-3, 3)])
    return {"type": "biconic", "curvx": 1.0 / r, "curvy": 1.0 / (r * rng.uniform(0.6, 1.5)),
            "ccx": rng.uniform(-0.3, 0.1), "ccy": rng.uniform(-0.3, 0.1), "coeffs": pairs}


bad = []
ntot = 0
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for seed in range(nseeds):
    rng = np.random.RandomState(31000 + seed)
    recs = []
    z = 0.0
    n_cur = 1.0
    nsurf = int(rng.randint(2, 5))
    for s in range(nsurf):
        z += rng.uniform(10.0, 30.0)
        Bs = tf.rot(rng, 0.25)
        g = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), z])
        last = s == nsurf - 1
        mirror = (not last) and s > 0 and rng.rand() < 0.35
        n_next = n_cur if mirror else (1.0 if last else float(rng.choice([1.492, 1.0, 1.6])))
        recs.append({"shape": {"type": "conic", "curv": 0.0, "cc": 0.0} if last else face(rng),
                     "B_shape": Bs.tolist(), "g_shape": g.tolist(),
                     "aperture": {"type": "circular", "minradius": 0.0, "maxradius": 45.0} if not last else {"type": "none"},
                     "B_ap": Bs.tolist(), "g_ap": g.tolist(), "interaction": "mirror" if mirror else "refract",
                     "material": {"type": "isotropic", "n": n_next}, "B_mat": np.eye(3).tolist()})
        if mirror:
            z -= rng.uniform(20.0, 50.0)
        n_cur = n_next
    n = 512
    x0 = np.vstack((rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), np.full(n, -5.0)))
    u = np.vstack((rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    res = engine.DeviceSystem(recs, 0).trace(engine.to_device_rays(x0, dev), engine.to_device_rays(k0, dev),
                                             engine.to_device_rays(e0, dev), want_nonconv=True)
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
            ek = np.abs(res.k_out[s].cpu().numpy()[:, wo] - np.real(ko)).max()
            if not ek < 1e-9:
                bad.append((seed, s, "k", float(ek))); break
        ntot += int(wo.sum())
print("compared ray-surfaces:", ntot, " failures:", len(bad))
for b in bad[:30]: print(b)
