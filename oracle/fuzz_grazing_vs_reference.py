#!/usr/bin/env python
"""
TEST INFRASTRUCTURE -- runs only in the build container (needs /root/reference).

Grazing incidence on explicit shapes (round 6, ADVICE r5 "the Newton stop rule at near-tangent rays"): random
near-hemispherical even aspheres and biconics under tilted bundles whose topmost rays pass a random margin (0.02 .. 1 mm)
below the line that touches the surface -- angles of incidence up to ~85 degrees, Newton's g'(t) = d . grad down to 0.1 --
traced by the REAL reference with its fsolve converged (annotations["tol"] = 1e-14, surface_shape.py:396, 457-458) and by
both oracles (NumPy and C) on the table flattened from the same objects; compared FLAT at 1e-10.  Systems on which the
reference itself does not reach the surface on every ray (residual > 1e-12 mm) are counted and skipped.  Also emulated on
every ray: the HIP kernel's stop rule (done behind a step <= 1e-8 that is <= 1e-3 of the step before it, prt_device.h
explicit_t) against the iteration run to 1e-15 -- the largest difference of the two roots.

    python oracle/fuzz_grazing_vs_reference.py [n_systems] [first_seed] > profiles/<tag>_reference_fuzz_grazing.txt
"""
import math
import os
import sys
import warnings

_n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
_first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg            # noqa: E402  (imports the reference with the NumPy-2 shim)
import numpy as np                  # noqa: E402

sys.path.insert(0, os.path.join(mg.ROOT, "tests"))
from oracle import seqtrace_np as onp          # noqa: E402
from oracle import seqtrace_c                   # noqa: E402


def system(rng):
    R = float(rng.uniform(12.0, 40.0))
    decz = float(rng.uniform(20.0, 40.0))
    tilt = float(rng.uniform(15.0, 40.0))
    rpup = float(rng.uniform(2.0, 6.0))
    margin = float(10 ** rng.uniform(math.log10(0.02), 0.0))
    a4 = float(rng.uniform(-1, 1) * 2e-7 * (20.0 / R) ** 3)
    if rng.rand() < 0.7:
        shape = {"shape": "Asphere", "curv": 1. / R, "cc": float(rng.uniform(-0.2, 0.2)), "coefficients": [0.0, a4, 0.0]}
    else:
        shape = {"shape": "Biconic", "curvx": 1. / R, "curvy": 1. / (R * float(rng.uniform(0.9, 1.1))),
                 "ccx": float(rng.uniform(-0.2, 0.2)), "ccy": float(rng.uniform(-0.2, 0.2)), "coefficients": [(a4, 0.1)]}
    kind = shape["shape"]
    bl = [({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {"is_stop": True}),
          (shape, {"decz": decz}, float(rng.uniform(1.4, 1.9)), "dome", {}),
          ({"shape": "Conic"}, {"decz": float(rng.uniform(20, 60))}, None, "image", {})]
    (s, seq) = mg.build_simple_optical_system(bl)
    # the tangent line of the SPHERE of radius R (the shape differs from it by the conic constant and a4: the margin
    # absorbs that, a bundle that misses the surface is skipped below)
    th = math.radians(tilt)
    (yt, zt) = (R * math.cos(th), decz + R - R * math.sin(th))
    z0 = -5.0
    yc = yt - (zt - z0) * math.tan(th) - margin - rpup
    bundle = mg.disk_bundle(120, rpup, z0, field_deg=tilt, yshift=yc)
    return s, seq, bundle, dict(R=R, tilt=tilt, margin=margin, kind=kind)


def hip_rule_error(rec, r0, d):
    """largest |t(HIP stop rule) - t(converged)| / max(1, |t|) over the rays, scalar Newton from t = 0 on the shape of
    ``rec`` (oracle's own F and gradient), both rules in one pass"""
    shape = rec["shape"]
    out = []
    for rule in ("converged", "hip"):
        t = np.zeros(r0.shape[1])
        done = np.zeros_like(t, dtype=bool)
        dt_prev = np.ones_like(t)
        for it in range(40):
            (px, py) = (r0[0] + t * d[0], r0[1] + t * d[1])
            with np.errstate(all="ignore"):
                g = r0[2] + t * d[2] - onp.shape_sag(shape, px, py)
                grad = onp.shape_grad(shape, px, py)              # (-Fx, -Fy, 1)
            gp = grad[0] * d[0] + grad[1] * d[1] + grad[2] * d[2]
            with np.errstate(all="ignore"):
                dt = g / gp
            tn = t - dt
            scale = np.maximum(1.0, np.abs(tn))
            small15 = ~(np.abs(dt) > 1e-15 * scale)
            if rule == "converged":
                now = small15
            else:
                contracted = (it > 0) & (np.abs(dt) <= 1e-3 * np.abs(dt_prev))
                now = ~(np.abs(dt) > 1e-8 * scale) & (contracted | small15)
            t = np.where(done, t, tn)
            done = done | now
            dt_prev = dt
            if done.all():
                break
        out.append((t, done))
    ok = out[0][1] & out[1][1] & np.isfinite(out[0][0])
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(out[1][0][ok] - out[0][0][ok]) / np.maximum(1.0, np.abs(out[0][0][ok]))))


HOST_BUILD = bool(os.environ.get("PRT_FUZZ_HOST_BUILD"))
if HOST_BUILD:
    sys.path.insert(0, os.path.join(mg.ROOT, "tests"))
    import hostemu                  # noqa: E402
host_worst = [0.0, 0.0, 0]


def main():
    have_grad = True
    (n_ok, n_skipped, n_rays, worst_x, worst_k, worst_inc, worst_rule) = (0, 0, 0, 0.0, 0.0, 0.0, 0.0)
    use_c = True
    for seed in range(_first, _first + _n):
        rng = np.random.RandomState(seed)
        (s, seq, b, info) = system(rng)
        for sf in mg._sequence_surfaces(s, seq):
            if "tol" in sf.shape.annotations:
                sf.shape.annotations["tol"] = mg.TIGHT_TOL
        (records, lengths) = mg.flatten_sequence(s, seq, b.wave)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                rp = s.seqtrace(b, seq)
        rb = rp[0].raybundles
        hit = mg._hit_bundle_indices(lengths)
        dome = mg._sequence_surfaces(s, seq)[1]
        xb = np.array(rb[hit[1]].x[-1])
        p = dome.shape.lc.returnGlobalToLocalPoints(xb)
        with np.errstate(all="ignore"):
            resid = np.abs(p[2] - dome.shape.getSag(p[0], p[1]))
        if not np.all(np.isfinite(resid)) or np.max(resid) > 1e-12 or xb.shape[1] < 20:
            n_skipped += 1           # the reference itself is not on the surface with every ray (or the bundle missed)
            continue
        ids = rb[hit[1]].rayID
        (x0, k0, e0) = (np.array(b.x[0]), np.array(b.k[0]), np.array(b.Efield[0]))
        dn = k0 / np.linalg.norm(k0, axis=0)
        g = dome.shape.getGrad(p[0], p[1])
        g = g / np.linalg.norm(g, axis=0)
        inc = np.degrees(np.arccos(np.clip(np.abs(np.sum(g * dn[:, ids], axis=0)), 0, 1)))
        outs = [onp.trace(records, x0, k0, e0)]
        if use_c and seqtrace_c.supports(records):
            outs.append(seqtrace_c.trace(records, x0, k0, e0))
        if HOST_BUILD:
            # the kernels' own Newton loop (stop rule, gradient moved to the root along the secant) as compiled from
            # pyrate_amd/csrc for the host (tests/hostemu), on the same rays
            hb = hostemu.HostSystem(records).trace(x0, k0, e0, want_nonconv=True)
            for (si, bi) in enumerate(hit):
                rbx = np.array(rb[bi].x[-1])
                rid = rb[bi].rayID
                dx = np.linalg.norm(hb[si]["x_hit"][:, rid] - rbx, axis=0) / np.maximum(np.linalg.norm(rbx, axis=0), 1.0)
                host_worst[0] = max(host_worst[0], float(np.max(dx)))
                host_worst[2] += int(np.count_nonzero(hb[si]["nonconv"][rid]))
                if bi + 1 < len(rb):
                    nb = rb[bi + 1]
                    host_worst[1] = max(host_worst[1], float(np.max(np.abs(hb[si]["k_out"][:, nb.rayID] - np.real(np.array(nb.k[0]))))))
        # every bundle of the reference's path against the dense oracle arrays, joined on rayID
        for o in outs:
            for (si, bi) in enumerate(hit):
                rbx = np.array(rb[bi].x[-1])
                rid = rb[bi].rayID
                dx = np.linalg.norm(o[si]["x_hit"][:, rid] - rbx, axis=0) / np.maximum(np.linalg.norm(rbx, axis=0), 1.0)
                worst_x = max(worst_x, float(np.max(dx)))
                if bi + 1 < len(rb):
                    nb = rb[bi + 1]
                    dk = np.abs(o[si]["k_out"][:, nb.rayID] - np.real(np.array(nb.k[0])))
                    worst_k = max(worst_k, float(np.max(dk)))
        if have_grad:
            t0 = -x0[2] / dn[2]                                   # on the stop plane z = 0
            r0 = dome.shape.lc.returnGlobalToLocalPoints(x0 + t0 * dn)
            dl = dome.shape.lc.returnGlobalToLocalDirections(dn)
            worst_rule = max(worst_rule, hip_rule_error(records[1], r0[:, ids], dl[:, ids]))
        n_ok += 1
        n_rays += int(xb.shape[1])
        worst_inc = max(worst_inc, float(np.max(inc)))
        if (seed - _first) % 20 == 0:
            print("seed %4d %-8s R %.1f tilt %.1f margin %.3f: %d rays on the dome, incidence up to %.1f deg, reference "
                  "residual %.1e" % (seed, info["kind"], info["R"], info["tilt"], info["margin"], xb.shape[1], np.max(inc),
                                     np.max(resid)), flush=True)
    print("grazing fuzz: %d systems compared (%d skipped: the reference not on the surface / bundle missed), %d rays on the "
          "explicit surface, incidence up to %.1f deg" % (n_ok, n_skipped, n_rays, worst_inc))
    print("oracles (NumPy%s) vs the converged reference, FLAT: max rel x %.2e, max abs k %.2e (bar 1e-10)"
          % (" + C" if use_c else "", worst_x, worst_k))
    if have_grad:
        print("HIP stop rule (1e-8 behind an observed contraction) vs the iteration run to 1e-15, emulated on the same rays: "
              "max |dt| / max(1, |t|) = %.2e" % worst_rule)
    if HOST_BUILD:
        print("the kernels' sources (host build, tests/hostemu) vs the converged reference, FLAT, same rays: max rel x %.2e, max "
              "abs k %.2e, rays flagged nonconv: %d" % (host_worst[0], host_worst[1], host_worst[2]))
        assert host_worst[0] < 1e-10 and host_worst[1] < 1e-10
    assert worst_x < 1e-10 and worst_k < 1e-10


if __name__ == "__main__":
    main()
