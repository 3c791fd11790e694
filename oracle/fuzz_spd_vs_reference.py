#!/usr/bin/env python
"""
TEST INFRASTRUCTURE -- build container only.  Random WinLens-SPD-shaped files (tests/systems_zoo.py
writes them: 1-5 lens groups of 2-4 surfaces, a stop behind a random group, random radii /
thicknesses / glass indices / paraxial summary rows) parsed by the reference's SPDFile +
ParaxialSystem and by pyrate_amd.raytracer.io.spd: surface lists and first-order numbers must agree.

    python oracle/fuzz_spd_vs_reference.py [n_files]
"""
import contextlib
import io
import os
import sys
import tempfile

sys.argv = sys.argv[:2]
_n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg            # noqa: E402
import numpy as np                  # noqa: E402

sys.path.insert(0, os.path.join(mg.ROOT, "tests"))
import systems_zoo as zoo           # noqa: E402
from test_importers import _spd_numbers as mine_numbers, _assert_spd_numbers   # noqa: E402


def main():
    from pyrateoptics.raytracer.io.spd import SPDParser as RefParser
    from pyrate_amd.raytracer.io.spd import SPDParser
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(_n):
            rng = np.random.RandomState(41000 + seed)
            ng = int(rng.randint(1, 6))
            groups = []
            glass = []
            for g in range(ng):
                ns = int(rng.randint(2, 5))
                groups.append([(float(rng.uniform(-1, 1) * rng.uniform(15, 300)), float(rng.uniform(5, 15))) for _ in range(ns)])
                glass.append([("G%d_%d" % (g, j), float(rng.uniform(1, 8)),
                               tuple(float(v) for v in np.sort(rng.uniform(1.45, 1.85, 3))[[1, 2, 0]]))
                              for j in range(ns - 1)])
            stop_after = int(rng.randint(0, ng))
            gaps = [float(rng.uniform(0.5, 10)) for _ in range(ng + 1)]
            summary = dict(efl=float(rng.uniform(40, 200)), mag=float(-rng.uniform(0.05, 0.5)), obj_dist=float(-rng.uniform(300, 2000)),
                           img_dist=float(rng.uniform(40, 200)), l=float(-rng.uniform(320, 2100)), ldash=float(rng.uniform(60, 250)),
                           track=float(rng.uniform(500, 3000)), stop_rad=float(rng.uniform(2, 8)), entpup_rad=float(rng.uniform(2, 9)),
                           expup_rad=float(rng.uniform(2, 9)), obj_angle=float(rng.uniform(1, 10)), obj_height=float(rng.uniform(10, 120)),
                           img_angle=float(-rng.uniform(1, 10)), img_height=float(-rng.uniform(2, 30)))
            f = os.path.join(tmp, "f%d.spd" % seed)
            zoo.write_synthetic_spd(f, groups, stop_after, glass, [587.6, 486.1, 656.3, 440.0, 700.0], summary, gaps)
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ref = RefParser(f, name="r")
                a = mg._spd_numbers(ref.psys)
                b = mine_numbers(SPDParser(f).psys)
                _assert_spd_numbers(b, a)
            except AssertionError as exc:
                bad.append((seed, "assert", str(exc)[:200]))
            except Exception as exc:
                bad.append((seed, "exception", repr(exc)[:200]))
    print("files %d, mismatches %d" % (_n, len(bad)))
    for b in bad[:20]:
        print(b)


if __name__ == "__main__":
    main()
