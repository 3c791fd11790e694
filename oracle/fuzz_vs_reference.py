#!/usr/bin/env python
"""
TEST INFRASTRUCTURE -- runs only in the build container (needs /root/reference).

Random systems built with the REAL reference classes, traced by the reference, against the NumPy
oracle on the surface table flattened from the very same objects: the oracle's pin beyond the
committed golden cases (random tilted frames, apertures, mirrors, ModelGlass, crystals with random
uniaxial / biaxial tensors, consecutive mirrors inside crystals, steep incidence with partially
evanescent crystal modes).

    python oracle/fuzz_vs_reference.py [n_systems] [first_seed]
    PRT_FUZZ_ENGINE=hostbuild ...: the reference against the HOST BUILD OF THE KERNELS' SOURCES (tests/hostemu: libprt's
    prt.hip compiled for x86-64, same C ABI) instead of the oracle -- the reference and the product's own code on the same
    random systems, nothing in between (round 6, when no GPU was to be had)
    PRT_FUZZ_TIGHT=1 ...: every explicit shape of the reference with annotations["tol"] = 1e-14 (its fsolve converged,
    surface_shape.py:396, 457-458) and the comparison FLAT at 1e-10, no allowance (round 5; the largest raw deviation
    is printed)
"""
import sys
import os
import types

_extra = len(sys.argv) > 3
sys.argv = sys.argv[:3]
_n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
_first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg            # noqa: E402  (imports the reference with the NumPy-2 shim)
import numpy as np                  # noqa: E402

sys.path.insert(0, os.path.join(mg.ROOT, "tests"))
import _golden                      # noqa: E402
from oracle import seqtrace_np as oracle          # noqa: E402
from test_oracle_golden import explicit_tolerance  # noqa: E402

A = mg.REFAPI
# python oracle/fuzz_vs_reference.py n first extra: also biconic / XY-polynomial surfaces and wider bundles (another
# random stream than the campaigns on record, which ran without)
EXTRA_SHAPES = _extra or bool(os.environ.get("PRT_FUZZ_EXTRA_SHAPES"))
TIGHT = bool(os.environ.get("PRT_FUZZ_TIGHT"))
ENGINE = os.environ.get("PRT_FUZZ_ENGINE", "oracle")
if ENGINE == "hostbuild":
    import hostemu                  # noqa: E402  (tests/hostemu)


def random_eps(rng):
    (a, b, c) = rng.uniform(-1.5, 1.5, 3)
    lc = A.LocalCoordinates.p(name="e", tiltx=a, tilty=b, tiltz=c)
    lc.update()
    R = lc.localbasis
    if rng.rand() < 0.5:
        (no, ne) = (rng.uniform(1.3, 2.1), rng.uniform(1.3, 2.1))
        pv = [no ** 2, no ** 2, ne ** 2]
    else:
        pv = list(np.sort(rng.uniform(1.3, 2.1, 3)) ** 2)
    return R.dot(np.diag(pv)).dot(R.T)


def random_system(rng, crystals):
    s = A.OpticalSystem.p()
    lc_prev = s.addLocalCoordinateSystem(A.LocalCoordinates.p(name="obj", decz=0.0),
                                         refname=s.rootcoordinatesystem.name)
    elem = A.OpticalElement.p(lc_prev, name="e")
    nsurf = int(rng.randint(2, 6))
    seq = []
    last_mat = None
    in_crystal = False
    for j in range(nsurf):
        tilted = rng.rand() < 0.5
        kw = dict(decz=float(rng.uniform(3, 9)))
        if tilted:
            kw.update(decx=float(rng.uniform(-0.3, 0.3)), tiltx=float(rng.uniform(-0.12, 0.12)),
                      tilty=float(rng.uniform(-0.12, 0.12)), tiltThenDecenter=int(rng.randint(0, 2)))
        lc = s.addLocalCoordinateSystem(A.LocalCoordinates.p(name="s%d" % j, **kw), refname=lc_prev.name)
        curv = float(rng.uniform(-1, 1) / rng.uniform(12, 80))
        kind = rng.rand()
        if (not crystals) and kind < 0.25:
            shape = A.Asphere.p(lc, curv=curv, cc=float(rng.uniform(-1.5, 0.5)),
                                coefficients=[float(rng.uniform(-1, 1) * 1e-4), float(rng.uniform(-1, 1) * 1e-7)])
        elif (not crystals) and EXTRA_SHAPES and kind < 0.40:
            # biconics with the large b_n of free-form prisms (demos/demo_hud.py: b up to 29): noisy far from the axis
            shape = A.Biconic.p(lc, curvx=curv, curvy=float(curv * rng.uniform(0.6, 1.4)),
                                ccx=float(rng.uniform(-0.5, 0.3)), ccy=float(rng.uniform(-0.5, 0.3)),
                                coefficients=[(0.0, 0.0), (float(rng.uniform(-1, 1) * 1e-7), float(rng.uniform(-30, 30))),
                                              (float(rng.uniform(-1, 1) * 1e-10), float(rng.uniform(-3, 3)))])
        elif (not crystals) and EXTRA_SHAPES and kind < 0.50:
            shape = A.XYPolynomials.p(lc, normradius=float(rng.uniform(5, 12)),
                                      coefficients=[(2, 0, float(rng.uniform(-0.1, 0.1))), (0, 2, float(rng.uniform(-0.1, 0.1))),
                                                    (2, 1, float(rng.uniform(-0.05, 0.05))), (0, 3, float(rng.uniform(-0.05, 0.05))),
                                                    (4, 0, float(rng.uniform(-0.02, 0.02)))])
        else:
            shape = A.Conic.p(lc, curv=curv, cc=float(rng.choice([0.0, rng.uniform(-1.5, 1.0)])))
        aper = None
        if rng.rand() < 0.4:
            aper = A.CircularAperture.p(lc, maxradius=float(rng.uniform(2.5, 7.0)))
        surf = A.Surface.p(lc, shape=shape, aperture=aper)
        mirror = j > 0 and rng.rand() < 0.2
        if mirror:
            mat = last_mat
        elif j == nsurf - 1:
            mat = None
        elif crystals and rng.rand() < 0.7:
            mat = "c%d" % j
            lcm = s.addLocalCoordinateSystem(A.LocalCoordinates.p(name="m%d" % j, tiltx=float(rng.uniform(-0.5, 0.5))),
                                             refname=lc.name)
            elem.addMaterial(mat, A.AnisotropicMaterial.p(lcm, random_eps(rng)))
        else:
            mat = "g%d" % j
            elem.addMaterial(mat, A.ConstantIndexGlass.p(lc, float(rng.uniform(1.3, 1.9))) if rng.rand() < 0.7
                             else A.ModelGlass.p(lc))
        elem.addSurface("s%d" % j, surf, (last_mat, mat))
        seq.append(("s%d" % j, {"is_mirror": True} if mirror else {}))
        if mirror:
            # the following surface lies behind the mirror
            pass
        last_mat = mat
        lc_prev = lc
    s.addElement("e", elem)
    if rng.rand() < 0.35:
        # visit surfaces more than once / out of order (folded paths, ghost-like sequences): exercises
        # the material bookkeeping of OpticalElement.seqtrace (optical_element.py:336-375), which
        # switches media by identity and only at refractions
        extra = []
        for _ in range(int(rng.randint(1, 4))):
            (name, _) = seq[int(rng.randint(0, len(seq)))]
            extra.append((name, {"is_mirror": True} if rng.rand() < 0.5 else {}))
        seq = seq + extra
    if rng.rand() < 0.3:
        # a second element sharing the frame tree: the medium starts again from the background
        lc_b = s.addLocalCoordinateSystem(A.LocalCoordinates.p(name="b0", decz=float(rng.uniform(3, 8))), refname=lc_prev.name)
        elem2 = A.OpticalElement.p(lc_b, name="e2")
        elem2.addMaterial("g", A.ConstantIndexGlass.p(lc_b, float(rng.uniform(1.4, 1.8))))
        lc_c = s.addLocalCoordinateSystem(A.LocalCoordinates.p(name="b1", decz=float(rng.uniform(2, 5))), refname=lc_b.name)
        elem2.addSurface("f", A.Surface.p(lc_b, shape=A.Conic.p(lc_b, curv=float(rng.uniform(-0.02, 0.02)))), (None, "g"))
        elem2.addSurface("r", A.Surface.p(lc_c, shape=A.Conic.p(lc_c, curv=float(rng.uniform(-0.02, 0.02)))), ("g", None))
        s.addElement("e2", elem2)
        return (s, [("e", seq), ("e2", [("f", {}), ("r", {})])])
    return (s, [("e", seq)])


def main():
    bad = []
    ncmp = 0
    worst = [0.0, 0.0]
    n_explicit = [0]
    for seed in range(_first, _first + _n):
        rng = np.random.RandomState(21000 + seed)
        crystals = seed % 2 == 1
        try:
            (s, seq) = random_system(rng, crystals)
            n = 24
            half = 6.0 if (EXTRA_SHAPES and not crystals) else 2.5
            x0 = np.vstack((rng.uniform(-half, half, n), rng.uniform(-half, half, n), np.full(n, -1.0)))
            steep = 0.45 if crystals else 0.2
            u = np.vstack((rng.uniform(-steep, steep, n), rng.uniform(-steep, steep, n), np.ones(n)))
            k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
            e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
            bundle = A.RayBundle(x0, k0, e0, wave=0.55e-3)
            (records, lengths) = mg.flatten_sequence(s, seq, bundle.wave)
            if TIGHT:
                for el in s.elements.values():
                    for sf in el.surfaces.values():
                        if "tol" in sf.shape.annotations:
                            sf.shape.annotations["tol"] = mg.TIGHT_TOL
            import warnings
            with np.errstate(all="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                rpaths = s.seqtrace(bundle, seq)
            rb = rpaths[0].raybundles
            case = types.SimpleNamespace(name="fuzz%d" % seed, table=records, n_surfaces=len(records),
                                         x0=x0, k0=k0, E0=e0, elem_lengths=lengths)
            allb = [dict(x=np.array(b.x), k=np.array(b.k), valid=np.array(b.valid), id=np.array(b.rayID))
                    for b in rb]
            # canonical list [b0, b0, b1, ..., bS]: drop the duplicate at every further element boundary
            keep = []
            pos = 0
            for (e, L) in enumerate(lengths):
                if e == 0:
                    keep += [pos, pos + 1]
                pos += 1
                keep += list(range(pos + 1, pos + 1 + L))
                pos += L
            assert pos + 1 == len(allb), (pos, len(allb))
            case.bundles = [allb[i] for i in keep]
            with np.errstate(all="ignore"):
                if ENGINE == "hostbuild":
                    out = hostemu.HostSystem(records).trace(x0, k0, e0)
                else:
                    out = _golden.dense_from_oracle(oracle.trace(records, x0, k0, e0))
            # evanescent descendants have complex k in the reference: compare only while every k is real
            if any(np.any(np.abs(np.imag(b["k"])) > 1e-9) for b in case.bundles):
                kind = "complex-k"
                # still compare the leading part: cut the case at the first bundle with complex k
                first = [i for (i, b) in enumerate(case.bundles) if np.any(np.abs(np.imag(b["k"])) > 1e-9)][0]
                upto = first - 2            # surfaces whose outgoing bundle is still real
                if upto < 1:
                    continue
                case.n_surfaces = upto
                case.table = records[:upto]
                case.bundles = case.bundles[:upto + 2]
                out = out[:upto]
            if TIGHT:
                r = _golden.compare_dense_to_reference(case, out, rtol_x=1e-10, atol_k=1e-10,
                                                       explicit_tol=None)
                if os.environ.get("PRT_FUZZ_VERBOSE") and max(r["raw_rel_x"], r["raw_abs_k"]) > 1e-12:
                    print("seed %d: raw deviation %.2e (x, relative) %.2e (k); shapes %s" % (
                        seed, r["raw_rel_x"], r["raw_abs_k"], [rec["shape"]["type"] for rec in case.table]))
                worst[0] = max(worst[0], r["raw_rel_x"])
                worst[1] = max(worst[1], r["raw_abs_k"])
                n_explicit[0] += any(rec["shape"]["type"] != "conic" for rec in case.table)
            else:
                r = _golden.compare_dense_to_reference(case, out, rtol_x=1e-9, atol_k=1e-9,
                                                       explicit_tol=explicit_tolerance)
            ncmp += r["n_compared"]
        except AssertionError as exc:
            bad.append((seed, "crystals" if crystals else "isotropic", str(exc)[:160]))
        except Exception as exc:
            bad.append((seed, "exception", repr(exc)[:200]))
    print("engine: %s" % ("the host build of libprt's sources (tests/hostemu)" if ENGINE == "hostbuild" else "NumPy oracle"))
    print("systems %d, compared ray-surfaces %d, failures %d" % (_n, ncmp, len(bad)))
    if TIGHT:
        print("tight mode (reference tol = 1e-14, flat 1e-10): %d systems with explicit shapes compared; largest raw "
              "deviation %.2e relative on hit points, %.2e on wave vectors" % (n_explicit[0], worst[0], worst[1]))
    for b in bad[:200]:
        print(b)


if __name__ == "__main__":
    main()
