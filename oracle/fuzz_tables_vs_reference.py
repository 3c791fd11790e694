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


build = zoo.random_object_graph


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
