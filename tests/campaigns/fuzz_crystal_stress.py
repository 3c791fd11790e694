"""one-off crystal stress fuzz: random stacks of 1-3 crystal interfaces (uniaxial / biaxial, strong
birefringence, mirrors inside crystals, tilted surface and material frames), HIP vs oracle"""
import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import seqtrace_np as oracle
from pyrate_amd import engine
import test_gpu_fuzz as tf
dev = torch.device("cuda", 0)
bad = []; ntot = 0
for seed in range(400):
    rng = np.random.RandomState(9000 + seed)
    def eps():
        kind = rng.randint(1, 3)
        R = tf.rot(rng, 1.5)
        if kind == 1:
            (no, ne) = (rng.uniform(1.3, 2.2), rng.uniform(1.3, 2.2))
            pv = np.array([no ** 2, no ** 2, ne ** 2])
        else:
            pv = np.sort(rng.uniform(1.3, 2.2, 3)) ** 2
        return R.dot(np.diag(pv)).dot(R.T)
    ncry = int(rng.randint(1, 4))
    tilted = seed % 2 == 0
    recs = []
    z = 0.0
    cur_aniso = False
    for s in range(ncry + 2):
        z += rng.uniform(3.0, 8.0)
        Bs = tf.rot(rng, 0.15) if tilted else np.eye(3)
        g = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), z]) if tilted else np.array([0., 0., z])
        c = rng.uniform(-1, 1) / rng.uniform(15, 80)
        mirror = cur_aniso and rng.rand() < 0.2
        if s < ncry or mirror:
            mat = {"type": "anisotropic", "eps_re": (eps() if not mirror else np.asarray(recs[-1]["material"]["eps_re"])).tolist(),
                   "eps_im": np.zeros((3, 3)).tolist()}
            cur_aniso = True
        else:
            mat = {"type": "isotropic", "n": 1.0 if s == ncry + 1 else float(rng.uniform(1.0, 1.8))}
            cur_aniso = False
        recs.append({"shape": {"type": "conic", "curv": c, "cc": float(rng.choice([0.0, rng.uniform(-1.5, 1.0)]))},
                     "B_shape": Bs.tolist(), "g_shape": g.tolist(), "aperture": {"type": "none"},
                     "B_ap": Bs.tolist(), "g_ap": g.tolist(), "interaction": "mirror" if mirror else "refract",
                     "material": mat, "B_mat": (tf.rot(rng, 0.8) if tilted else np.eye(3)).tolist()})
        if mirror:
            z -= rng.uniform(6.0, 14.0)
    n = 200
    x0 = np.vstack((rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    try:
        with np.errstate(all="ignore"):
            out = oracle.trace(recs, x0, k0, e0)
        res = engine.DeviceSystem(recs, 0).trace(*[engine.to_device_rays(a, dev, pitched=False) for a in (x0, k0, e0)])
    except Exception as exc:
        bad.append((seed, -1, "exception", repr(exc)[:200])); continue
    taint = np.zeros(n, dtype=bool)     # descendants of evanescent (complex k) modes: NaN in the engine by design
    for s in range(len(recs)):
        xo = out[s]["x_hit"]; xd = res.x_hit[s].cpu().numpy()
        v = out[s]["valid"] & np.all(np.isfinite(xo), axis=0) & ~taint
        vd = res.valid[s].cpu().numpy().astype(bool)
        fin_d = np.all(np.isfinite(xd), axis=0)
        if not np.array_equal(vd[v], out[s]["valid"][v]) or not fin_d[v].all():
            bad.append((seed, s, "valid/finite", int(np.sum(vd[v] != out[s]["valid"][v])) + int((~fin_d[v]).sum()))); break
        if v.any():
            ex = np.abs(xd[:, v] - xo[:, v]).max()
            if not ex < 1e-8:
                bad.append((seed, s, "x", float(ex))); break
        ko = np.real(out[s]["k_out"]); kd = res.k_out[s].cpu().numpy()
        if ko.shape[1] == 2 * taint.shape[0]:
            taint = np.concatenate((taint, taint))
        taint = taint | ~np.all(np.abs(np.imag(out[s]["k_out"])) < 1e-12, axis=0)
        fin = np.all(np.isfinite(ko), axis=0) & ~taint
        if fin.any():
            ek = np.abs(kd[:, fin] - ko[:, fin]).max()
            if not ek < 1e-8:
                bad.append((seed, s, "k", float(ek), recs[s]["material"]["type"], recs[s]["interaction"])); break
        ntot += int(fin.sum())
print("compared ray-surfaces:", ntot, " failures:", len(bad))
for b in bad[:30]: print(b)
