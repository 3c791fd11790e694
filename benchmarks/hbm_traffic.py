#!/usr/bin/env python
"""HBM bytes per launch of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE and
WRITE_SIZE collected separately, MI355X_MICROARCH.md "HBM traffic"): counters are in KiB-like units
of 1024 B per the guide... (values are reported in KB = 1024 B); gfx950 correction: FETCH_SIZE
counts 32-B requests for 16 B/lane coalesced reads, i.e. half the bytes -> doubled.

    python benchmarks/hbm_traffic.py <dir with pmc_fetch/ and pmc_write/> [kernel substring] > hbm_traffic.json
"""
import csv
import glob
import json
import os
import sys


def counter_average(root, counter, kernel):
    vals = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for kernel %r under %s" % (counter, kernel, root))
    return sum(vals) / len(vals), len(vals)


def main():
    root = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_trace_iso"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 9994476
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    (fetch_kb, nf) = counter_average(os.path.join(root, "pmc_fetch"), "FETCH_SIZE", kernel)
    (write_kb, nw) = counter_average(os.path.join(root, "pmc_write"), "WRITE_SIZE", kernel)
    fetch = 2.0 * fetch_kb * 1024.0
    write = write_kb * 1024.0
    alg_r = n * 72
    alg_w = n * 49 * S            # x_hit 24 + k_out 24 + one byte of packed mask flags per record
    out = {"path_%d" % n: {
        "bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
        "FETCH_SIZE_avg_kb": fetch_kb, "WRITE_SIZE_avg_kb": write_kb,
        "launches_fetch": nf, "launches_write": nw,
        "algorithmic_bytes": alg_r + alg_w, "algorithmic_read": alg_r, "algorithmic_write": alg_w,
        "ratio_to_algorithmic": (fetch + write) / (alg_r + alg_w),
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                "`bench.py --steps 5 --warmup 2 --no-cpu-baseline`; kernel %s; FETCH_SIZE doubled per the "
                "guide's gfx950 correction for 16 B/lane coalesced reads; WRITE_SIZE uncalibrated" % kernel}}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
