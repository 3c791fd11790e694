"""one-off fuzz, three more dimensions: (a) mirrors on explicit shapes, (b) explicit shapes as crystal
interfaces, (c) the same random systems scaled by 1e-4 and 1e+4 in length"""
import sys, math, copy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import _golden
from oracle import seqtrace_np as oracle
from pyrate_amd import engine
import test_gpu_fuzz as tf
dev = torch.device("cuda", 0)

def compare(recs, x0, k0, e0, tag, bad, tolx=1e-9, tolk=1e-9, crystal=False):
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    res = engine.DeviceSystem(recs, 0).trace(*[engine.to_device_rays(a, dev, pitched=not crystal) for a in (x0, k0, e0)])
    n = x0.shape[1]
    taint = np.zeros(n, dtype=bool)
    cnt = 0
    for s in range(len(recs)):
        xo = out[s]["x_hit"]; xd = res.x_hit[s].cpu().numpy()
        v = out[s]["valid"] & np.all(np.isfinite(xo), axis=0) & ~taint
        vd = res.valid[s].cpu().numpy().astype(bool)
        if not np.array_equal(vd[v], out[s]["valid"][v]) or not np.all(np.isfinite(xd[:, v])):
            bad.append((tag, s, "valid/finite")); return cnt
        if v.any():
            ex = (np.abs(xd[:, v] - xo[:, v]) / _golden.relative_scale(xo[:, v])).max()
            if not ex < tolx:
                bad.append((tag, s, "x", float(ex))); return cnt
        ko = np.real(out[s]["k_out"]); kd = res.k_out[s].cpu().numpy()
        if ko.shape[1] == 2 * taint.shape[0]:
            taint = np.concatenate((taint, taint))
        taint = taint | ~np.all(np.abs(np.imag(out[s]["k_out"])) < 1e-12, axis=0)
        wo = out[s]["valid_out"] & ~taint
        wd = res.valid_out[s].cpu().numpy().astype(bool)
        # rays whose hit point is NaN on one side only are the borderline-Newton category
        if not crystal and not np.array_equal(wd[~taint], out[s]["valid_out"][~taint]):
            nd = int(np.sum(wd[~taint] != out[s]["valid_out"][~taint]))
            if nd > 3:
                bad.append((tag, s, "valid_out", nd)); return cnt
            taint = taint | (wd != out[s]["valid_out"])
            wo = wo & ~taint
        fin = np.all(np.isfinite(ko), axis=0) & wo
        if fin.any():
            ek = np.abs(kd[:, fin] - ko[:, fin]).max()
            if not ek < tolk:
                bad.append((tag, s, "k", float(ek))); return cnt
        cnt += int(fin.sum())
    return cnt

def bundle(rng, n, r=5.0, ang=0.15):
    x0 = np.vstack((rng.uniform(-r, r, n), rng.uniform(-r, r, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-ang, ang, n), rng.uniform(-ang, ang, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    return x0, k0, e0

bad = []; tot = 0
# (a) mirrors on any shape
for seed in range(200):
    rng = np.random.RandomState(13000 + seed)
    recs = tf.random_table(rng, int(rng.randint(3, 7)), seed % 2 == 1, True, False)
    z = recs[1]["g_shape"][2]
    recs[1]["interaction"] = "mirror"
    recs[1]["material"] = dict(recs[0]["material"])
    for r in recs[2:]:                        # fold the rest of the system back behind the mirror
        r["g_shape"][2] = 2 * z - r["g_shape"][2]; r["g_ap"][2] = 2 * z - r["g_ap"][2]
    tot += compare(recs, *bundle(rng, 300), ("mirror-any-shape", seed), bad)
# (b) explicit shapes as crystal interfaces
src = open('tests/campaigns/fuzz_crystal_stress.py').read()
crystal_body = src[src.index("    rng = np.random.RandomState(9000 + seed)"):src.index("    n = 200\n")]
for seed in range(150):
    exec("if True:\n" + crystal_body)
    rng2 = np.random.RandomState(15000 + seed)
    for r in recs:
        kind = int(rng2.randint(0, 6))
        if kind:
            r["shape"] = tf.random_shape(rng2, kind)
    tot += compare(recs, *bundle(rng2, 120, r=3.0, ang=0.2), ("crystal-explicit", seed), bad, crystal=True)
# (c) length scales
for seed in range(120):
    rng = np.random.RandomState(17000 + seed)
    recs = tf.random_table(rng, int(rng.randint(3, 7)), seed % 2 == 1, seed % 3 != 0, False)
    (x0, k0, e0) = bundle(rng, 200)
    for scale in (1e-4, 1e4):
        r2 = copy.deepcopy(recs)
        for r in r2:
            r["g_shape"] = [v * scale for v in r["g_shape"]]; r["g_ap"] = [v * scale for v in r["g_ap"]]
            sh = r["shape"]
            if "curv" in sh: sh["curv"] /= scale
            if sh["type"] == "asphere": sh["coeffs"] = [a / scale ** (2 * q + 1) for (q, a) in enumerate(sh["coeffs"])]
            if sh["type"] == "biconic":
                sh["curvx"] /= scale; sh["curvy"] /= scale
                sh["coeffs"] = [[a / scale ** (2 * q + 1), b] for (q, (a, b)) in enumerate(sh["coeffs"])]
            if sh["type"] in ("xypoly", "zernike"):
                sh["normradius"] *= scale
                key = "terms" if sh["type"] == "xypoly" else "coeffs"
                sh[key] = [[t[0], t[1], t[2] * scale] for t in sh[key]] if key == "terms" else [c * scale for c in sh[key]]
            if sh["type"] == "combination":
                r["shape"] = {"type": "conic", "curv": 0.01 / scale, "cc": 0.0}
            ap = r["aperture"]
            for key in ("minradius", "maxradius", "width", "height"):
                if key in ap: ap[key] *= scale
        tot += compare(r2, x0 * scale, k0, e0, ("scale %g" % scale, seed), bad, tolx=1e-8)
print("compared ray-surfaces:", tot, " failures:", len(bad))
for b in bad[:30]: print(b)
