"""random absorbing-crystal sequences: REAL reference vs the NumPy oracle on the table flattened from the same objects"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (build container only: imports the real reference)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
warnings.simplefilter("ignore")
import numpy as np
import make_golden as mg
from oracle import seqtrace_np
from pyrate_amd.surface_table import flatten_sequence
zoo = mg.zoo; api = mg.REFAPI
worst = [0.0, 0.0]; ncmp = 0
for seed in range(40):
    rng = np.random.RandomState(seed)
    def tensor():
        a = rng.uniform(1.3, 2.4, 3) ** 2
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        e = q @ np.diag(a) @ q.T + 1j * (q @ np.diag(rng.uniform(0.0, 0.3, 3) * (10.0 ** -rng.randint(0, 4))) @ q.T)
        if seed % 2:
            e = e + 0.02 * rng.normal(size=(3, 3)) * (1 + 0.5j)
        return e
    mirror = bool(rng.randint(2))
    (s, seq) = zoo.crystal_inside(api, tensor(), tilt_deg=rng.uniform(-15, 15), mirror=mirror, eps2=tensor())
    b = mg.disk_bundle(16, 3.0, -5.0, field_deg=rng.uniform(-20, 20))
    (records, _) = flatten_sequence(s, seq, b.wave)
    with np.errstate(all="ignore"):
        rp = s.seqtrace(b, seq)[0]
        out = seqtrace_np.trace(records, np.array(b.x[0]), np.array(b.k[0]), np.array(b.Efield[0]))
    # bundles: [b0 (2 pts), b1 stop->front, b2 ..]; dense surface s <-> bundle s+1 last point / bundle s+2 first k
    for sidx in range(len(records)):
        xr = rp.raybundles[sidx + 1].x[-1]
        kr = rp.raybundles[sidx + 2].k[0]
        xo = out[sidx]["x_hit"]; ko = np.asarray(out[sidx]["k_out"], dtype=complex)
        ok = np.all(np.isfinite(xr), axis=0) & np.all(np.abs(xr) < 1e6, axis=0)
        if xr.shape != xo.shape or kr.shape != ko.shape:
            print("seed", seed, "shape mismatch", xr.shape, xo.shape); continue
        worst[0] = max(worst[0], float(np.abs(xo[:, ok] - xr[:, ok]).max()))
        ok2 = np.hstack((ok, ok)) if kr.shape[1] == 2 * ok.size else ok
        fin = ok2 & np.all(np.isfinite(kr), axis=0)
        worst[1] = max(worst[1], float(np.abs(ko[:, fin] - kr[:, fin]).max()))
        ncmp += int(ok.sum())
print("systems 40, compared ray-surfaces", ncmp, "max |dx|", worst[0], "max |dk|", worst[1])

# ---- sequences that END in an isotropic medium: the folded beam of an absorbing slab leaving into air (steep
#      incidence: rays dropped by the last refraction), a singlet in front of an absorbing detector (complex index,
#      metal-like ones included: Re n^2 < 0 drops every ray) -- survivors and their complex k, both oracles
from oracle import seqtrace_c
worst = {"np": [0.0, 0.0], "c": [0.0, 0.0]}; ncmp = 0; ninvalid = 0; mask_mismatch = 0
for seed in range(60):
    rng = np.random.RandomState(1000 + seed)
    def tensor():
        a = rng.uniform(1.3, 2.4, 3) ** 2
        q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        return q @ np.diag(a) @ q.T + 1j * (q @ np.diag(rng.uniform(0.0, 0.3, 3) * (10.0 ** -rng.randint(0, 4))) @ q.T)
    if seed % 2 == 0:
        (s, seq) = zoo.crystal_mirror(api, tensor(), tilt_deg=rng.uniform(3, 25))
        b = mg.disk_bundle(16, 3.0, -5.0, field_deg=rng.uniform(-25, 25))
    else:
        n_abs = complex(rng.uniform(0.2, 4.0), rng.uniform(0.0, 3.0) * (10.0 ** -rng.randint(0, 3)))
        (s, seq) = zoo.absorbing_detector(api, n_abs)
        b = mg.disk_bundle(16, 4.0, -5.0, field_deg=rng.uniform(-25, 25))
    (records, _) = flatten_sequence(s, seq, b.wave)
    with np.errstate(all="ignore"):
        rp = s.seqtrace(b, seq)[0]
        outs = {"np": seqtrace_np.trace(records, np.array(b.x[0]), np.array(b.k[0]), np.array(b.Efield[0])),
                "c": seqtrace_c.trace(records, np.array(b.x[0]), np.array(b.k[0]), np.array(b.Efield[0]))}
    # the last bundle of the reference: rays that survived the last (isotropic) refraction, with complex k
    last = rp.raybundles[-1]
    for (key, out) in outs.items():
        o = out[-1]
        v = o["valid_out"]
        ids = o["ray_id"][v] if o["ray_id"].shape[0] == v.shape[0] else None
        kr = last.k[0]
        ko = np.asarray(o["k_out"], dtype=complex)[:, v]
        if ko.shape != kr.shape:
            mask_mismatch += 1
            print("seed", seed, key, "survivor count differs", ko.shape, kr.shape)
            continue
        if kr.shape[1]:
            worst[key][1] = max(worst[key][1], float(np.abs(ko - kr).max()))
            worst[key][0] = max(worst[key][0], float(np.abs(o["x_hit"][:, v if o["x_hit"].shape[1] == v.shape[0] else slice(None)] - last.x[0]).max()))
        if key == "np":
            ncmp += kr.shape[1]; ninvalid += int((~v).sum())
print("systems 60, surviving rays compared", ncmp, "rays dropped by the last refraction", ninvalid, "mask mismatches", mask_mismatch)
print("max |dx| / |dk| numpy oracle", worst["np"], " C oracle", worst["c"])
