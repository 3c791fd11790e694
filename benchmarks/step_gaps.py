#!/usr/bin/env python
"""Idle time between consecutive marches of a `bench.py --force-multi` run, from a rocprofv3 kernel trace (csv):
period of the march launches, their duration, and what lies between the end of one march's step (march + the two
moment kernels) and the start of the next.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --force-multi --steps 50 [...]
    python benchmarks/step_gaps.py <dir>/t_kernel_trace.csv [label] > profiles/<tag>_force_multi_gaps_<label>.json
"""
import csv
import json
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
march = [(s, e) for (s, e, n) in rows if n.startswith("void k_trace_iso")]
# the timed region: the longest run of marches whose period stays within 30 % of the median
per = [march[i + 1][0] - march[i][0] for i in range(len(march) - 1)]
med = sorted(per)[len(per) // 2]
(best, cur) = ((0, 0), None)
for (i, p) in enumerate(per):
    if abs(p - med) < 0.3 * med:
        cur = (cur[0], i + 1) if cur else (i, i + 1)
        if cur[1] - cur[0] > best[1] - best[0]:
            best = cur
    else:
        cur = None
(a, b) = best
others = {}
for i in range(a, b):
    for (s, e, n) in rows:
        if march[i][1] <= s < march[i + 1][0]:
            others[n.split("(")[0]] = others.get(n.split("(")[0], 0.0) + (e - s) / 1e3 / (b - a)
period = sum(per[a:b]) / (b - a) / 1e3
dur = sum(e - s for (s, e) in march[a:b]) / (b - a) / 1e3
print(json.dumps({"label": sys.argv[2] if len(sys.argv) > 2 else None, "launches_in_the_steady_run": b - a,
                  "period_us": round(period, 2), "march_us": round(dur, 2),
                  "other_kernels_between_two_marches_us": {k: round(v, 2) for (k, v) in others.items()},
                  "idle_us": round(period - dur - sum(others.values()), 2)}, indent=1))
