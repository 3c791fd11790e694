#!/usr/bin/env python
"""Per-launch averages of rocprofv3 PMC counters for one kernel (name substring), as JSON.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR/fetch -- <cmd>
    python benchmarks/pmc_summary.py DIR k_trace_iso [--skip-first N]

Reads every *counter_collection.csv under DIR (one sub-directory per PMC pass).  Derived figures, when
the counters are there:
  hbm_bytes_per_launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (MI355X_MICROARCH.md "HBM": counters in
                         KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of 16 B/lane coalesced reads
                         at 64 B -> doubled; WRITE_SIZE as reported)
  fp64_flops_per_launch = 64 * (2 * SQ_INSTS_VALU_FMA_F64 + SQ_INSTS_VALU_ADD_F64 + SQ_INSTS_VALU_MUL_F64
                                + SQ_INSTS_VALU_TRANS_F64)          (wave instructions x 64 lanes)
"""
import csv
import glob
import json
import os
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    (root, kernel) = (args[0], args[1])
    skip = 0
    if "--skip-first" in sys.argv:
        skip = int(sys.argv[sys.argv.index("--skip-first") + 1])
    vals = {}
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per_file = {}
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if kernel in row.get("Kernel_Name", ""):
                    per_file.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for (name, v) in per_file.items():
            vals.setdefault(name, []).extend(v[skip:])
    out = {"kernel": kernel, "counters": {k: {"avg": sum(v) / len(v), "launches": len(v)} for (k, v) in vals.items() if v}}
    c = {k: v["avg"] for (k, v) in out["counters"].items()}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["fetch_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0
        out["write_bytes"] = c["WRITE_SIZE"] * 1024.0
        out["hbm_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
    f64 = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64")
    if all(k in c for k in f64):
        out["fp64_flops_per_launch"] = 64.0 * (2.0 * c[f64[0]] + c[f64[1]] + c[f64[2]] + c[f64[3]])
        out["fp64_valu_wave_instructions"] = sum(c[k] for k in f64)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
