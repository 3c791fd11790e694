import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'scratch')
import numpy as np, torch
np.set_printoptions(precision=17, linewidth=200)
from oracle import seqtrace_np as oracle
from pyrate_amd import engine
import test_gpu_fuzz as tf
dev = torch.device("cuda", 0)
_orig = tf.random_shape
def harsh_shape(rng, kind):
    c = rng.uniform(-1, 1) / rng.uniform(6, 40)
    if kind == 0:
        return {"type": "conic", "curv": c, "cc": rng.choice([0.0, rng.uniform(-3, 3)])}
    if kind == 1:
        return {"type": "asphere", "curv": c, "cc": rng.uniform(-2.5, 1.5),
                "coeffs": [rng.uniform(-1, 1) * 1e-3, rng.uniform(-1, 1) * 1e-5, rng.uniform(-1, 1) * 1e-8]}
    return _orig(rng, kind)
for (seed, sbad) in ((74, 0), (803, 2), (979, 2)):
    rng = np.random.RandomState(7000 + seed)
    tf.random_shape = harsh_shape
    recs = tf.random_table(rng, int(rng.randint(3, 9)), seed % 2 == 1, seed % 3 != 0, seed % 4 == 3)
    tf.random_shape = _orig
    n = 777
    x0 = np.vstack((rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-0.35, 0.35, n), rng.uniform(-0.35, 0.35, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    res = engine.DeviceSystem(recs, 0).trace(engine.to_device_rays(x0, dev), engine.to_device_rays(k0, dev),
                                             engine.to_device_rays(e0, dev))
    s = sbad
    print("seed", seed, "surface", s, recs[s]["shape"], recs[s]["aperture"], recs[s]["interaction"])
    vo = out[s]["valid"]; vd = res.valid[s].cpu().numpy().astype(bool)
    idx = np.where(vo != vd)[0]
    wo = out[s]["valid_out"]; wd = res.valid_out[s].cpu().numpy().astype(bool)
    widx = np.where(wo != wd)[0]
    kd = res.k_out[s].cpu().numpy(); ko = np.real(out[s]["k_out"])
    print(" valid_out mismatches:", len(widx), " material", recs[s]["material"], " prev material", recs[s-1]["material"] if s else None)
    for i in widx[:4]:
        print("  ray", i, "valid o/h", vo[i], vd[i], " valid_out o/h", wo[i], wd[i])
        print("   x oracle", out[s]["x_hit"][:, i], " x hip", res.x_hit[s].cpu().numpy()[:, i])
        print("   k oracle", ko[:, i], " k hip", kd[:, i])
    xd = res.x_hit[s].cpu().numpy(); xo = out[s]["x_hit"]
    if len(idx) == 0:
        bad = np.where(vo & ~np.all(np.isfinite(xd), axis=0))[0]
        bad2 = np.where(vo & ~np.all(np.isfinite(xo), axis=0))[0]
        print(" valid equal; HIP non-finite at valid:", bad[:5], " oracle non-finite at valid:", bad2[:5])
        idx = np.concatenate((bad, bad2))[:3]
    for i in idx[:4]:
        print("  ray", i, "oracle valid", vo[i], "hip valid", vd[i])
        print("   x oracle", xo[:, i], " x hip", xd[:, i])
        if s > 0:
            print("   prev valid_out oracle/hip", out[s-1]["valid_out"][i], bool(res.valid_out[s-1][i]),
                  " prev x", out[s-1]["x_hit"][:, i], " prev k", np.real(out[s-1]["k_out"][:, i]))
        p = oracle.g2l_points(np.asarray(recs[s]["B_shape"]), np.asarray(recs[s]["g_shape"]), xo[:, i:i+1])
        print("   local hit (oracle)", p.ravel(), " ap frame:", oracle.g2l_points(np.asarray(recs[s]["B_ap"]), np.asarray(recs[s]["g_ap"]), xo[:, i:i+1]).ravel())
