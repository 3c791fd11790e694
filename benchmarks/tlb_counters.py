#!/usr/bin/env python
"""Why does the conic march lose 2-3 points of its HBM fraction between 1e7 and 1e8 rays?  Address-translation and
write-path counters of the SAME kernel at both sizes (rocprofv3 PMC passes, kernel trace only, a few counters per
pass; arena-placed path arrays like the bench), per launch and per ray:

    python benchmarks/tlb_counters.py > profiles/<tag>_tlb_counters_1e7_vs_1e8.json

    python benchmarks/tlb_counters.py --inner RAYS      (what runs under the profiler)
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PASSES = [
    ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_TRANSLATION_MISS_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum", "TCP_UTCL1_STALL_MULTI_MISS_sum", "TCP_UTCL1_STALL_INFLIGHT_MAX_sum"],
    ["TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_UTCL1_SERIALIZATION_STALL_sum"],
    ["TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_TAG_STALL_sum", "TCC_BUSY_sum"],
    ["GRBM_GUI_ACTIVE", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum"],
]
LAUNCHES = 6


def inner(rays):
    import torch
    from benchmarks import workloads as bench
    from pyrate_amd import engine, _lib
    dev = torch.device("cuda", 0)
    wl = bench.make_workload("doublegauss", rays, dev)
    sysd = engine.DeviceSystem(wl["records"], 0)
    n = wl["n_local"]
    ob = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True, placement="arena", pitch=engine.recommended_pitch(n))
    for _ in range(LAUNCHES):
        sysd.trace_into(wl["x0"], None, ob, uniform=wl["uniform"])
    torch.cuda.synchronize()
    print("rays", n)


def main():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = {"kernel": "k_trace_iso (double Gauss, path mode, uniform first segment, arena-placed arrays)", "sizes": {}}
    env = dict(os.environ, TMPDIR="/tmp")
    for rays in (10_000_000, 100_000_000):
        rec = {}
        for counters in PASSES:
            tmp = tempfile.mkdtemp(prefix="prt_tlb_", dir="/tmp")
            cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", tmp, "--",
                                                                  sys.executable, os.path.abspath(__file__), "--inner", str(rays)]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            if res.returncode != 0:
                rec["error_" + counters[0]] = res.stderr[-300:]
                shutil.rmtree(tmp, ignore_errors=True)
                continue
            vals = {}
            for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if "k_trace_iso" in row.get("Kernel_Name", ""):
                            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for (k, v) in vals.items():
                rec[k] = sum(v) / len(v)
            shutil.rmtree(tmp, ignore_errors=True)
        out["sizes"][str(rays)] = rec
    (a, b) = (out["sizes"].get("10000000", {}), out["sizes"].get("100000000", {}))
    out["per_ray_ratio_1e8_over_1e7"] = {k: (b[k] / 10.0) / a[k] for k in a if k in b and not k.startswith("error") and a[k]}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--inner":
        inner(int(sys.argv[2]))
    else:
        main()
