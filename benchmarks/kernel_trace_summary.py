#!/usr/bin/env python
"""Per-launch view of the dominant kernel from a rocprofv3 --kernel-trace run of bench.py:
the `--stats` average runs over ALL launches of the process -- device wake-up, the placement scan
(pairs of output arrays of different speed, listed in the bench line's config.output_placement),
warm-up, the K timed steps and the launches bench.py times with HIP events for roofline.kernel_ms.
This prints the launches in order, grouped the way bench.py issues them, so that the averages of
the timed groups can be compared with the bench line.

    python benchmarks/kernel_trace_summary.py <rocprofv3 output dir> [kernel substring] [steps] > summary.json
"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_trace_iso"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    d = [ns * 1e-6 for (_, ns) in rows]
    if not d:
        raise SystemExit("no launches of %r under %s" % (kernel, root))

    def avg(v):
        return sum(v) / len(v) if v else None
    timed_events = d[-steps:]                    # sysd.trace_timed(..., max(steps, 5)) -> roofline.kernel_ms
    timed_steps = d[-2 * steps:-steps]           # the K timed steps (ms_per_step)
    setup = d[:-2 * steps]
    json.dump({"kernel": kernel, "launches": len(d),
               "avg_ms_all_launches": avg(d), "min_ms": min(d), "max_ms": max(d),
               "avg_ms_last_%d_launches_(kernel_ms_group)" % steps: avg(timed_events),
               "avg_ms_timed_steps_group": avg(timed_steps),
               "avg_ms_setup_launches_(wake-up, placement scan, warm-up)": avg(setup),
               "setup_launches": len(setup),
               "per_launch_ms": [round(t, 4) for t in d]}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
