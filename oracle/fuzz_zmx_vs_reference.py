#!/usr/bin/env python
"""
TEST INFRASTRUCTURE -- build container only.  Random Zemax prescriptions (STANDARD / EVENASPH /
BICONICX / COORDBRK surfaces, CLAP / SQAP / OBDC apertures, named / model / mirror glasses, a stop)
parsed by the reference's ZMXParser and by pyrate_amd.raytracer.io.zmx: the flattened surface tables
must be identical.

    python oracle/fuzz_zmx_vs_reference.py [n_files]
"""
import json
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


def random_zmx(rng):
    lines = ["VERS 140124 258 36214", "MODE SEQ", "NAME random system %d" % rng.randint(1000), "NOTE 0 generated",
             "UNIT MM X W X CM MR CPMM", "ENPD %.3f" % rng.uniform(5, 20), "FTYP 0 0 2 1 0 0 0",
             "XFLN 0 0 0 0", "YFLN 0 %.2f 0 0" % rng.uniform(1, 5), "WAVM 1 0.55 1", "WAVM 2 0.65 1"]
    nsurf = int(rng.randint(3, 9))
    stop = int(rng.randint(1, nsurf))
    lines += ["SURF 0", "  TYPE STANDARD", "  CURV 0.0 0 0 0 0 \"\"", "  DISZ %.4f" % rng.uniform(5, 30)]
    for j in range(1, nsurf + 1):
        lines.append("SURF %d" % j)
        if j == stop:
            lines.append("  STOP")
        r = rng.rand()
        typ = "STANDARD" if r < 0.4 else ("EVENASPH" if r < 0.65 else ("BICONICX" if r < 0.8 else "COORDBRK"))
        if j == nsurf:
            typ = "STANDARD"
        lines.append("  TYPE " + typ)
        lines.append("  CURV %.12e 0 0 0 0 \"\"" % (0.0 if typ == "COORDBRK" else rng.uniform(-1, 1) / rng.uniform(15, 200)))
        if typ != "COORDBRK" and rng.rand() < 0.5:
            lines.append("  CONI %.6f" % rng.uniform(-2, 1))
        if typ == "EVENASPH":
            for q in range(1, 9):
                lines.append("  PARM %d %.6e" % (q, rng.uniform(-1, 1) * 10.0 ** (-3 - 2 * q) if q < 4 else 0))
        elif typ == "BICONICX":
            lines += ["  PARM 1 %.6f" % (rng.uniform(-1, 1) * rng.uniform(20, 200)), "  PARM 2 %.4f" % rng.uniform(-1, 0.5)]
        elif typ == "COORDBRK":
            for (q, v) in enumerate((rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-5, 5), rng.uniform(-5, 5),
                                     rng.uniform(-5, 5), float(rng.randint(0, 2))), start=1):
                lines.append("  PARM %d %.6f" % (q, v))
        lines.append("  DISZ %.6f" % (rng.uniform(1, 25) if j < nsurf else 0.0))
        if typ != "COORDBRK" and j < nsurf:
            g = rng.rand()
            if g < 0.3:
                lines.append("  GLAS %s 0 0 1.5 40 0 0 0 0 0 0" % rng.choice(["BK7", "SF5"]))
            elif g < 0.45:
                lines.append("  GLAS MODELGL 1 0 %.5f 0.0 0 0 0 0 0 0" % rng.uniform(1.4, 1.9))
            elif g < 0.55:
                lines.append("  GLAS MIRROR 0 0 1.5 40 0 0 0 0 0 0")
            a = rng.rand()
            if a < 0.25:
                lines.append("  CLAP %.4f %.4f" % (rng.choice([0.0, 0.5]), rng.uniform(3, 12)))
            elif a < 0.4:
                lines.append("  SQAP %.4f %.4f" % (rng.uniform(3, 10), rng.uniform(3, 10)))
            if a < 0.4 and rng.rand() < 0.5:
                lines.append("  OBDC %.4f %.4f" % (rng.uniform(-1, 1), rng.uniform(-1, 1)))
        lines.append("  DIAM %.3f 1 0 0 1 \"\"" % rng.uniform(5, 15))
    return "\n".join(lines) + "\n"


def main():
    from pyrateoptics.raytracer.io.zmx import ZMXParser as RefParser
    from pyrate_amd.raytracer.io.zmx import ZMXParser
    from pyrate_amd.surface_table import flatten_sequence
    mine = zoo.mirror_api()
    ref = mg.REFAPI
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(_n):
            rng = np.random.RandomState(51000 + seed)
            f = os.path.join(tmp, "f%d.zmx" % seed)
            with open(f, "w") as fh:
                fh.write(random_zmx(rng))
            try:
                lr = ref.LocalCoordinates.p(name="g")
                lm = mine.LocalCoordinates.p(name="g")
                (sr, seqr) = RefParser(f, name="r").create_optical_system(
                    {"BK7": ref.ConstantIndexGlass.p(lr, 1.5168), "SF5": ref.ConstantIndexGlass.p(lr, 1.6727)})
                zp = ZMXParser(f)
                (sm, seqm) = zp.create_optical_system(
                    {"BK7": mine.ConstantIndexGlass.p(lm, 1.5168), "SF5": mine.ConstantIndexGlass.p(lm, 1.6727)})
                (a, la) = flatten_sequence(sr, seqr, 0.55e-3)
                (b, lb) = flatten_sequence(sm, seqm, 0.55e-3)
                assert la == lb and [n for (n, _) in seqr[0][1]] == [n for (n, _) in seqm[0][1]]
                assert [o for (_, o) in seqr[0][1]] == [o for (_, o) in seqm[0][1]], "options"
                if json.dumps(a, sort_keys=True) != json.dumps(b, sort_keys=True):
                    diff = [(i, k) for (i, (ra, rb)) in enumerate(zip(a, b)) for k in ra if json.dumps(ra[k]) != json.dumps(rb[k])]
                    raise AssertionError("tables differ: %s" % diff[:5])
                rf = RefParser(f, name="r")
                assert json.loads(json.dumps(rf.read_field())) == json.loads(json.dumps(zp.read_field())), "field"
                assert json.loads(json.dumps(rf.create_initial_bundle())) == json.loads(json.dumps(zp.create_initial_bundle()))
            except AssertionError as exc:
                bad.append((seed, "assert", str(exc)[:200]))
            except Exception as exc:
                bad.append((seed, "exception", repr(exc)[:200]))
    print("files %d, mismatches %d" % (_n, len(bad)))
    for b in bad[:20]:
        print(b)


if __name__ == "__main__":
    main()
