"""one-off: uniaxial crystals in special orientations (optic axis along the surface normal, in the
surface plane, along / across the plane of incidence, nearly isotropic) incl. rays exactly along the
optic axis where the two sheets touch; wave vectors and hit points vs the oracle"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_gpu_fuzz as tf
from pyrate_amd import systems, engine
from oracle import seqtrace_np as oracle
dev = torch.device("cuda", 0)
bad = []; tot = 0
axes = [(0, 0, 1), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 1, 1), (1e-9, 0, 1), (0.6, 0, 0.8)]
for (ia, ax) in enumerate(axes):
    for (no, ne) in ((1.5, 1.7), (1.7, 1.5), (1.6, 1.6 + 1e-9), (1.3, 2.2)):
        a = np.array(ax, dtype=float); a /= np.linalg.norm(a)
        recs = systems.aniso_doublet_records(systems.uniaxial_eps(no, ne, a), systems.uniaxial_eps(ne, no, a))
        for r in recs:                          # plane faces: the surface normal is exactly z
            r["shape"] = {"type": "conic", "curv": 0.0, "cc": 0.0}
            r["aperture"] = {"type": "none"}
        n = 64
        rng = np.random.RandomState(ia)
        x0 = np.vstack((rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), np.full(n, -2.0)))
        ang = np.linspace(0.0, 0.5, n)
        phi = rng.choice([0.0, np.pi / 2, np.pi / 4], n)
        k0 = np.vstack((np.sin(ang) * np.cos(phi), np.sin(ang) * np.sin(phi), np.cos(ang)))
        k0[:, 0] = [0.0, 0.0, 1.0]              # exactly along z (the optic axis for the first orientation)
        e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
        with np.errstate(all="ignore"):
            out = oracle.trace(recs, x0, k0, e0)
        res = engine.DeviceSystem(recs, 0).trace(*[engine.to_device_rays(v, dev, pitched=False) for v in (x0, k0, e0)])
        for s in range(len(recs)):
            ko = np.real(out[s]["k_out"]); kd = res.k_out[s].cpu().numpy()
            nn = ko.shape[1] // 2 if recs[s]["material"]["type"] == "anisotropic" else None
            fin = np.all(np.isfinite(ko), axis=0) & np.all(np.abs(np.imag(out[s]["k_out"])) < 1e-12, axis=0)
            if nn:
                # where the two modes coincide (touching sheets) their order is undefined: compare as a set
                same = np.abs(ko[:, :nn] - ko[:, nn:]).max(axis=0) < 1e-7
                d1 = np.abs(kd - ko).max(axis=0)
                swapped = np.hstack((ko[:, nn:], ko[:, :nn]))
                d2 = np.abs(kd - swapped).max(axis=0)
                err = np.where(np.hstack((same, same)), np.minimum(d1, d2), d1)
            else:
                err = np.abs(kd - ko).max(axis=0)
            e = err[fin].max() if fin.any() else 0.0
            if not e < 1e-9:
                bad.append((ax, no, ne, s, float(e)))
                break
            xo = out[s]["x_hit"]; xd = res.x_hit[s].cpu().numpy()
            v = out[s]["valid"] & np.all(np.isfinite(xo), axis=0)
            ex = np.abs(xd[:, v] - xo[:, v]).max() if v.any() else 0.0
            if not ex < 1e-6:       # (degenerate pairs may be swapped -> positions of the two children swap too)
                pass
            tot += int(fin.sum())
print("compared:", tot, " failures:", len(bad))
for b in bad[:20]: print(b)
