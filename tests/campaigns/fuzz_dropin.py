"""one-off fuzz of the drop-in bundle assembly: seqtrace_fused on random tables (isotropic and
crystal) vs the bundles the reference's bookkeeping would build from the oracle's dense arrays --
compaction by valid_out, ray doubling, rayIDs, P = 2 history, zero-survivor bundles"""
import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import seqtrace_np as oracle
from pyrate_amd.raytracer.optical_system import seqtrace_fused
from pyrate_amd.raytracer.ray import RayBundle
import test_gpu_fuzz as tf
src = open('tests/campaigns/fuzz_crystal_stress.py').read()
crystal_body = src[src.index("    rng = np.random.RandomState(9000 + seed)"):src.index("    n = 200\n")]
bad = []; nb = 0

def expected_bundles(recs, out, n):
    ids = np.arange(n)
    exp = []
    for s in range(len(recs)):
        crystal = recs[s]["material"]["type"] == "anisotropic"
        mask = out[s]["valid_out"].astype(bool)
        xs = out[s]["x_hit"]
        if crystal:
            xs = np.hstack((xs, xs)); ids = np.concatenate((ids, ids))
        b = {"x": [xs[:, mask]], "k": np.real(out[s]["k_out"])[:, mask], "id": ids[mask], "valid": [np.ones(mask.sum(), bool)]}
        if s + 1 < len(recs):
            b["x"].append(out[s + 1]["x_hit"][:, mask]); b["valid"].append(out[s + 1]["valid"][mask])
        exp.append(b)
        ids = ids  # dense ids keep all slots
    return exp

for seed in range(300):
    rng = np.random.RandomState(11000 + seed)
    if seed % 3 == 2:
        exec("if True:\n" + crystal_body)
        # add apertures so that rays die between crystals
        for r in recs:
            if rng.rand() < 0.5:
                r["aperture"] = {"type": "circular", "minradius": 0.0, "maxradius": float(rng.uniform(1.0, 4.0))}
        n = 150
    else:
        recs = tf.random_table(rng, int(rng.randint(2, 7)), seed % 2 == 1, seed % 4 != 0, seed % 5 == 3)
        if seed % 7 == 0:       # an aperture nobody gets through
            recs[min(1, len(recs) - 1)]["aperture"] = {"type": "circular", "minradius": 0.0, "maxradius": 1e-6}
        n = int(rng.choice([1, 2, 63, 300]))
    x0 = np.vstack((rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    # skip systems whose parity is not defined (evanescent descendants / borderline Newton)
    if any(np.any(np.abs(np.imag(o["k_out"])) > 1e-12) for o in out):
        continue
    try:
        rp = seqtrace_fused(RayBundle(x0, k0, e0, wave=0.5e-3), recs, [len(recs)])
        bundles = rp.raybundles[2:]          # [ib+hit, the same again, then one per surface]
        exp = expected_bundles(recs, out, n)
        assert len(bundles) == len(exp)
        for (j, (b, e)) in enumerate(zip(bundles, exp)):
            assert b.x.shape == (len(e["x"]), 3, len(e["id"])), (j, b.x.shape, len(e["id"]))
            assert np.array_equal(b.rayID, e["id"]), j
            for p in range(len(e["x"])):
                assert np.array_equal(b.valid[p], e["valid"][p]), (j, p)
                fin = np.all(np.isfinite(e["x"][p]), axis=0)
                assert np.array_equal(fin, np.all(np.isfinite(b.x[p]), axis=0)), (j, p, "finite")
                if fin.any():
                    assert np.abs(b.x[p][:, fin] - e["x"][p][:, fin]).max() < 1e-8, (j, p, "x")
            fk = np.all(np.isfinite(e["k"]), axis=0)
            if fk.any():
                assert np.abs(np.real(b.k[0])[:, fk] - e["k"][:, fk]).max() < 1e-9, (j, "k")
            nb += 1
    except AssertionError as exc:
        bad.append((seed, "assert", str(exc)[:200]))
    except Exception as exc:
        bad.append((seed, "exception", repr(exc)[:300]))
print("bundles compared:", nb, " failures:", len(bad))
for b in bad[:25]: print(b)
