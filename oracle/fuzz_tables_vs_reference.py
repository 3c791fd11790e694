#!/usr/bin/env python
"""
TEST INFRASTRUCTURE -- build container only.  The same random object graph is built twice, once with
the reference's classes and once with this package's mirror classes (LocalCoordinates trees with both
tilt orders, nested frames, aperture and material frames, all shape classes); the two flattened
surface tables must be identical.

    python oracle/fuzz_tables_vs_reference.py [n_systems]
"""
import json
import os
import sys

sys.argv = sys.argv[:2]
_n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg            # noqa: E402
import numpy as np                  # noqa: E402

sys.path.insert(0, os.path.join(mg.ROOT, "tests"))
import systems_zoo as zoo           # noqa: E402


def build(api, seed):
    rng = np.random.RandomState(31000 + seed)
    s = api.OpticalSystem.p()
    lc_prev = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="obj", decz=float(rng.uniform(0, 2))),
                                         refname=s.rootcoordinatesystem.name)
    elem = api.OpticalElement.p(lc_prev, name="e")
    seq = []
    last = None
    for j in range(int(rng.randint(2, 7))):
        kw = {k: float(rng.uniform(-0.4, 0.4)) for k in ("decx", "decy", "tiltx", "tilty", "tiltz") if rng.rand() < 0.5}
        kw["decz"] = float(rng.uniform(1, 9))
        kw["tiltThenDecenter"] = int(rng.randint(0, 2))
        lc = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="s%d" % j, **kw), refname=lc_prev.name)
        lcs = lc
        if rng.rand() < 0.3:            # shape in its own nested frame
            lcs = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="sh%d" % j, decx=float(rng.uniform(-0.2, 0.2)),
                                                                    tiltz=float(rng.uniform(-0.5, 0.5))), refname=lc.name)
        kind = int(rng.randint(0, 6))
        c = float(rng.uniform(-1, 1) / rng.uniform(15, 90))
        if kind == 0:
            shape = api.Conic.p(lcs, curv=c, cc=float(rng.uniform(-1.5, 1)))
        elif kind == 1:
            shape = api.Asphere.p(lcs, curv=c, cc=float(rng.uniform(-1.5, 1)),
                                  coefficients=[float(v) for v in rng.uniform(-1e-5, 1e-5, int(rng.randint(0, 4)))])
        elif kind == 2:
            shape = api.Biconic.p(lcs, curvx=c, ccx=float(rng.uniform(-1, 0.5)), curvy=c * 0.7, ccy=float(rng.uniform(-1, 0.5)),
                                  coefficients=[(float(rng.uniform(-1e-5, 1e-5)), float(rng.uniform(-0.5, 0.5)))])
        elif kind == 3:
            shape = api.XYPolynomials.p(lcs, normradius=float(rng.uniform(5, 12)),
                                        coefficients=[(int(i), int(k), float(rng.uniform(-0.05, 0.05)))
                                                      for (i, k) in ((2, 0), (0, 2), (1, 2), (3, 1))])
        elif kind == 4:
            shape = api.ZernikeFringe.p(lcs, normradius=float(rng.uniform(6, 12)),
                                        coefficients=[float(v) for v in rng.uniform(-0.02, 0.02, int(rng.randint(1, 12)))])
        else:
            lcz = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="z%d" % j, decx=float(rng.uniform(-1, 1)),
                                                                    decy=float(rng.uniform(-1, 1))), refname=lcs.name)
            shape = api.LinearCombination.p(lcs, list_of_coefficients_and_shapes=[
                (float(rng.uniform(0.5, 1.5)), api.Asphere.p(lcs, curv=c, cc=-0.5, coefficients=[0.0, 1e-6])),
                (float(rng.uniform(0.5, 1.5)), api.ZernikeFringe.p(lcz, normradius=10.0,
                                                                   coefficients=[float(v) for v in rng.uniform(-0.02, 0.02, 6)]))])
        aper = None
        r = rng.rand()
        if r < 0.3:
            aper = api.CircularAperture.p(lc, maxradius=float(rng.uniform(3, 8)), minradius=float(rng.choice([0.0, 0.4])))
        elif r < 0.5:
            lca = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="a%d" % j, decy=0.3, tiltz=0.2), refname=lc.name)
            aper = api.RectangularAperture.p(lca, width=float(rng.uniform(6, 12)), height=float(rng.uniform(6, 12)))
        surf = api.Surface.p(lc, shape=shape, aperture=aper)
        mirror = j > 0 and rng.rand() < 0.2
        mat = last
        if not mirror:
            r = rng.rand()
            if r < 0.25:
                mat = None
            else:
                mat = "m%d" % j
                lcm = s.addLocalCoordinateSystem(api.LocalCoordinates.p(name="mf%d" % j, tiltx=float(rng.uniform(-0.6, 0.6)),
                                                                        tilty=float(rng.uniform(-0.6, 0.6))), refname=lc.name)
                if r < 0.55:
                    elem.addMaterial(mat, api.ConstantIndexGlass.p(lcm, float(rng.uniform(1.3, 1.9))))
                elif r < 0.7:
                    elem.addMaterial(mat, api.ModelGlass.p(lcm))
                else:
                    e = rng.uniform(-0.2, 0.2, (3, 3))
                    elem.addMaterial(mat, api.AnisotropicMaterial.p(lcm, np.eye(3) * 2.4 + e + e.T))
        elem.addSurface("s%d" % j, surf, (last, mat))
        seq.append(("s%d" % j, {"is_mirror": True} if mirror else {}))
        last = mat
        lc_prev = lc
    s.addElement("e", elem)
    return (s, [("e", seq)])


def main():
    from pyrate_amd.surface_table import flatten_sequence
    mine = zoo.mirror_api()
    bad = []
    for seed in range(_n):
        (sr, seqr) = build(mg.REFAPI, seed)
        (sm, seqm) = build(mine, seed)
        wave = 0.5876e-3
        (a, la) = flatten_sequence(sr, seqr, wave)
        (b, lb) = flatten_sequence(sm, seqm, wave)
        if la != lb or json.dumps(a, sort_keys=True) != json.dumps(b, sort_keys=True):
            diff = [k for (ra, rb) in zip(a, b) for k in ra if json.dumps(ra[k]) != json.dumps(rb[k])]
            bad.append((seed, diff[:6]))
    print("systems %d, table mismatches %d" % (_n, len(bad)))
    for b in bad[:20]:
        print(b)


if __name__ == "__main__":
    main()
