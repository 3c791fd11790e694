#!/usr/bin/env python
"""Does the march time follow the clocks the device sustains?  For the double Gauss (conic, store bound) and the asphere
(Newton: part of its arithmetic is exposed) at 1e7 rays: 60 x (50 timed launches, then one look at the device's shader /
memory clock, power and temperature through sysfs, amdgpu pp_dpm_* and hwmon), in one process on the same arrays.

    python benchmarks/clock_sensitivity.py > profiles/<tag>_clock_sensitivity.json
"""
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from benchmarks import workloads as bench
from pyrate_amd import engine, _lib


def sysfs_state():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            continue
        for (key, name) in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk")):
            try:
                for line in open(os.path.join(card, name)):
                    if line.strip().endswith("*"):
                        out[key] = line.split(":")[1].replace("*", "").strip()
            except OSError:
                pass
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for (key, name) in (("power_uW", "power1_average"), ("power_in_uW", "power1_input"), ("temp_mC", "temp1_input"),
                                ("temp_mem_mC", "temp3_input"), ("freq1_Hz", "freq1_input")):
                try:
                    out[key] = int(open(os.path.join(hw, name)).read())
                except (OSError, ValueError):
                    pass
        try:
            out["busy"] = int(open(os.path.join(card, "gpu_busy_percent")).read())
        except (OSError, ValueError):
            pass
        break
    return out


def main():
    dev = torch.device("cuda", 0)
    out = {"what": __doc__.strip().split("\n\n")[0], "configs": {}}
    for config in ("doublegauss", "asphere", "xypoly"):
        wl = bench.make_workload(config, 10_000_000, dev)
        sysd = engine.DeviceSystem(wl["records"], 0)
        n = wl["n_local"]
        bufs = sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True, placement="arena", pitch=engine.recommended_pitch(n))
        for _ in range(3):
            sysd.trace_timed(wl["x0"], wl["k0"], bufs, 50, wl["e0"], uniform=wl["uniform"])
        rows = []
        for it in range(60):
            ms = sysd.trace_timed(wl["x0"], wl["k0"], bufs, 50, wl["e0"], uniform=wl["uniform"])
            st = sysfs_state()
            st["kernel_ms"] = ms
            rows.append(st)
            if it == 29:
                time.sleep(1.0)            # an idle second in the middle: does the device come back at other clocks?
        ms_all = [r["kernel_ms"] for r in rows]
        out["configs"][config] = {"rays": n, "kernel_ms_min_median_max": [min(ms_all), sorted(ms_all)[len(ms_all) // 2], max(ms_all)],
                                  "rows": rows}
        del bufs, sysd, wl
    print(json.dumps(out))


if __name__ == "__main__":
    main()
