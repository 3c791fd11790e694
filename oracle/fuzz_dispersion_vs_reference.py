#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- build container only.  Random refractiveindex.info pages (every dispersion
formula type, random coefficient counts, tabulated n / k / nk data, two-entry pages) through the
reference's CatalogMaterial and this package's: identical index at random wavelengths, identical
out-of-range errors."""
import os
import sys

sys.argv = sys.argv[:2]
_n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg            # noqa: E402
import numpy as np                  # noqa: E402


def random_page(rng):
    typ = int(rng.randint(1, 10))
    lo = float(rng.uniform(0.3, 0.5))
    hi = float(rng.uniform(0.8, 2.0))
    if typ <= 7:
        nco = {1: 1 + 2 * int(rng.randint(1, 4)), 2: 1 + 2 * int(rng.randint(1, 4)), 3: 1 + 2 * int(rng.randint(1, 5)),
               4: int(rng.choice([9, 11, 13])), 5: 1 + 2 * int(rng.randint(1, 4)), 6: 1 + 2 * int(rng.randint(1, 3)),
               7: int(rng.randint(3, 7))}[typ]
        c = rng.uniform(0.001, 1.0, nco)
        if typ in (1, 2):
            c[0] = rng.uniform(0, 0.5); c[2::2] = rng.uniform(0.05, 0.2, len(c[2::2]))
        if typ == 3:
            c[0] = rng.uniform(2, 3); c[1::2] = rng.uniform(-0.01, 0.01, len(c[1::2])); c[2::2] = rng.choice([2, -2, -4, 4], len(c[2::2]))
        if typ == 4:
            c[:9] = [2.7, 0.45, 2, 0.12, 1, 0.9, 2, 9.0, 2]
            if nco > 9: c[9:] = [(-0.01, 2, 0.001, -2)[q] for q in range(nco - 9)]
        if typ == 5:
            c[0] = rng.uniform(1.4, 1.8); c[1::2] = rng.uniform(0, 0.01, len(c[1::2])); c[2::2] = rng.choice([-2, -4, -1], len(c[2::2]))
        if typ == 6:
            c[0] = 0.0; c[1::2] = rng.uniform(0.001, 0.06, len(c[1::2])); c[2::2] = rng.uniform(50, 250, len(c[2::2]))
        if typ == 7:
            c[:] = rng.uniform(-0.01, 0.01, nco); c[0] = rng.uniform(1.4, 1.8)
        entry = {"type": "formula %d" % typ, "wavelength_range": "%r %r" % (lo, hi), "coefficients": " ".join("%r" % float(v) for v in c)}
        data = [entry]
    else:
        w = np.sort(rng.uniform(lo, hi, int(rng.randint(3, 9))))
        kind = ["tabulated n", "tabulated k", "tabulated nk"][typ - 8]
        rows = []
        for wi in w:
            vals = [float(rng.uniform(1.4, 1.8))] if kind != "tabulated nk" else [float(rng.uniform(1.4, 1.8)), float(rng.uniform(0, 0.1))]
            rows.append(" ".join(["%r" % float(wi)] + ["%r" % v for v in vals]))
        data = [{"type": kind, "data": "\n".join(rows) + "\n"}]
        if kind == "tabulated k":       # n from a formula, k from the table
            data = [{"type": "formula 5", "wavelength_range": "%r %r" % (float(w[0]), float(w[-1])), "coefficients": "1.5 0.004 -2"}] + data
        (lo, hi) = (float(w[0]), float(w[-1]))
    return ({"DATA": data}, lo, hi)


def main():
    from pyrateoptics.raytracer.material.material_glasscat import CatalogMaterial as Ref
    from pyrate_amd.raytracer.material.material_glasscat import CatalogMaterial as Mine
    from pyrate_amd.raytracer.localcoordinates import LocalCoordinates
    lr = mg.LocalCoordinates.p(name="d")
    lm = LocalCoordinates.p(name="d")
    bad = []
    for seed in range(_n):
        rng = np.random.RandomState(61000 + seed)
        (page, lo, hi) = random_page(rng)
        try:
            (a, b) = (Ref.p(lr, page), Mine.p(lm, page))
            for w in list(rng.uniform(lo, hi, 4)) + [lo * 0.9, hi * 1.1]:
                try:
                    na = a.get_optical_index(None, w * 1e-3)
                    ea = None
                except Exception as exc:
                    ea = str(exc)
                try:
                    nb = b.get_optical_index(None, w * 1e-3)
                    eb = None
                except Exception as exc:
                    eb = str(exc)
                assert (ea is None) == (eb is None), (w, ea, eb)
                if ea is None:
                    both_nan = np.isnan(complex(na)) and np.isnan(complex(nb))
                    assert both_nan or abs(complex(na) - complex(nb)) < 1e-13, (page["DATA"][0]["type"], w, na, nb)
                else:
                    assert ea == eb, (ea, eb)
        except AssertionError as exc:
            bad.append((seed, str(exc)[:200]))
        except Exception as exc:
            bad.append((seed, "exception " + repr(exc)[:200]))
    print("pages %d, mismatches %d" % (_n, len(bad)))
    for b in bad[:15]:
        print(b)


if __name__ == "__main__":
    main()
