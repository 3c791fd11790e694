#!/usr/bin/env python
"""Issue-side picture of the kernels that are NOT HBM-bound (image-mode march, Newton shapes,
crystal solver) from rocprofv3 SQ counters, next to the HBM-bound headline kernel.

    # on an MI355X, from the repo root (two PMC passes, kernel trace only -- never with sys-trace):
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
        --output-format csv -d gpurun_out/valu/a -- python benchmarks/valu_profile.py run
    rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
        --output-format csv -d gpurun_out/valu/b -- python benchmarks/valu_profile.py run
    python benchmarks/valu_profile.py report gpurun_out/valu > gpurun_out/valu/valu_profile.json

``run`` launches each workload a few times; every workload maps to its own kernel instantiation, so
the report keys on the kernel name.  Derived figures (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* and
SQ_WAVE_CYCLES count quad-cycles; 1024 SIMDs; a wave64 FP64 instruction occupies its SIMD's 16-lane
VALU for 4 cycles = one quad-cycle):
    valu_busy   = SQ_ACTIVE_INST_VALU * 4 / (duration * sclk * 1024)    share of all SIMD-cycles issuing VALU
    valu_busy_at_grbm_clock = the same against GRBM_GUI_ACTIVE / 8 cycles (the clock the launch really ran at)
    wave_active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES                    share of resident-wave time issuing
    wave_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES                           share waiting on s_waitcnt (memory)
"""
import csv
import glob
import json
import math
import os
import sys

SCLK_HZ = 2.39e9        # observed shader clock during the march (image mode throttles to ~2.25e9)
N_SIMD = 1024


def run():
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyrate_amd import _lib, engine, systems
    dev = torch.device("cuda", 0)
    reps = 6

    def go(sysd, o, k, e0, mode, packed=False):
        (x0, k0, e0d) = [engine.to_device_rays(a, dev) for a in (o, k, e0)]
        bufs = sysd.alloc_outputs(o.shape[1], mode, packed_flags=packed)
        for _ in range(reps):
            sysd.trace_into(x0, k0, bufs, e0d)
        torch.cuda.synchronize()

    sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
    (o, k, e0) = systems.double_gauss_bundle(int(1e7))
    go(sysd, o, k, e0, _lib.MODE_PATH, packed=True)
    go(sysd, o, k, e0, _lib.MODE_IMAGE, packed=True)
    sysd = engine.DeviceSystem(systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5), 0)
    (o, k, e0) = systems.double_gauss_bundle(int(1e7), rpup=9.0, z0=-5.0, field_deg=5.0)
    go(sysd, o, k, e0, _lib.MODE_PATH, packed=True)
    go(sysd, o, k, e0, _lib.MODE_IMAGE, packed=True)
    c = systems.CALCITE_TILTED
    for (e1, e2) in ((systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
                      systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))),
                     (np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2]), np.diag([1.62 ** 2, 1.66 ** 2, 1.70 ** 2]))):
        sysd = engine.DeviceSystem(systems.aniso_doublet_records(e1, e2), 0)
        (o, k) = systems.collimated_bundle(int(1e6), 11.43, -5.0)
        e0 = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T.copy()
        go(sysd, o, k, e0, _lib.MODE_PATH)


def report(root):
    counters = {}
    durations = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "k_trace" not in name:
                    continue
                counters.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "k_trace" in name:
                    durations.setdefault(name, []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    out = {}
    for (name, cs) in sorted(counters.items()):
        avg = {k: sum(v) / len(v) for (k, v) in cs.items()}
        ds = sorted(durations.get(name, []))
        dur_ns = ds[len(ds) // 2] if ds else float("nan")          # median launch, under the profiler
        entry = {"launches": len(next(iter(cs.values()))), "duration_ms_under_pmc": dur_ns * 1e-6, "counters": avg}
        if "SQ_ACTIVE_INST_VALU" in avg and ds:
            entry["valu_busy"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (dur_ns * 1e-9 * SCLK_HZ * N_SIMD)
        if "GRBM_GUI_ACTIVE" in avg and ds:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = busy cycles of the launch -> the effective
            # shader clock under the profiler, and VALU busy against THOSE cycles (an estimate: the
            # counter's clock domain is not documented for gfx950)
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
            entry["sclk_ghz_from_grbm"] = cyc / dur_ns
            if "SQ_ACTIVE_INST_VALU" in avg:
                entry["valu_busy_at_grbm_clock"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD)
        if "SQ_INSTS_VALU" in avg and ds:
            # floor if every VALU instruction were a full-rate FP64 op: one quad-cycle each
            entry["valu_insts_per_wave"] = avg["SQ_INSTS_VALU"] / max(avg.get("SQ_WAVES", float("nan")), 1.0)
            entry["issue_floor_ms"] = avg["SQ_INSTS_VALU"] * 4.0 / (SCLK_HZ * N_SIMD) * 1e3
        if "SQ_WAVE_CYCLES" in avg:
            for (key, src) in (("wave_active", "SQ_ACTIVE_INST_ANY"), ("wave_parked", "SQ_WAIT_ANY"),
                               ("wave_issue_stalled", "SQ_WAIT_INST_ANY")):
                if src in avg:
                    entry[key] = avg[src] / avg["SQ_WAVE_CYCLES"]
        out[name] = entry
    json.dump({"sclk_hz_assumed": SCLK_HZ, "simds": N_SIMD, "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) > 2 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        raise SystemExit(__doc__)
