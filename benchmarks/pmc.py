"""Live PMC passes of bench.py: the configurations re-run under ``rocprofv3 --kernel-trace --pmc`` (separate passes:
FETCH_SIZE, WRITE_SIZE, the FP64 instruction counters), collected and corrected as MI355X_MICROARCH.md "HBM"
prescribes -- counters in KiB; FETCH_SIZE doubled on gfx950 (the 128-B requests of 16 B/lane coalesced reads are
tallied at 64 B); WRITE_SIZE as reported.

Attribution: ``inner`` (what runs under the profiler: ``bench.py --pmc-inner a,b,...``) launches every configuration
PMC_LAUNCHES times, in the order given, and nothing else of its kernels in between; the kernels that count are the
marches (one launch per trace) and, for the per-surface paths (``plugin``, ``aniso_chain``), the k_propagate /
k_interact_* launches (a known number per trace).  A counter row belongs to the configuration whose slice of that
dispatch order it falls into; per launch = per TRACE (the sum over a trace's kernels).
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64",
               "SQ_INSTS_VALU"))
PMC_LAUNCHES = 6
PER_SURFACE_CONFIGS = ("plugin", "surface_step", "aniso_chain")       # traces that are one or two launches per surface


def inner(args, dev):
    """what runs under rocprofv3: every requested config, PMC_LAUNCHES traces each, nothing else timed"""
    import torch
    from pyrate_amd import engine, _lib
    from . import workloads, configs
    for config in args.pmc_inner.split(","):
        if config in ("plugin", "surface_step"):
            (sweep, _) = configs.plugin_sweep(dev, args.rays_of[config], placement="torch", fused=(config == "surface_step"))
            for _ in range(PMC_LAUNCHES):
                sweep()
            torch.cuda.synchronize()
            del sweep
            torch.cuda.empty_cache()
            continue
        wl = workloads.make_workload(config, args.rays_of[config], dev, first_segment=args.first_segment)
        sysd = engine.DeviceSystem(wl["records"], dev.index)
        iso = sysd.all_isotropic
        packed = iso and not args.two_mask_arrays
        mode = _lib.MODE_PATH if args.mode == "path" else _lib.MODE_IMAGE
        # (placement does not change the bytes a launch moves: plain torch arrays, no arena hunt under the profiler)
        ob = sysd.alloc_outputs(wl["n_local"], mode, packed_flags=packed, placement="torch",
                                pitch=engine.recommended_pitch(wl["n_local"]) if iso else None)
        for _ in range(PMC_LAUNCHES):
            sysd.trace_into(wl["x0"], wl["k0"], ob, wl["e0"], uniform=wl["uniform"])
        torch.cuda.synchronize()
        del ob, wl, sysd
        torch.cuda.empty_cache()


def is_march(kernel_name):
    return "k_trace_general<" in kernel_name or "k_trace_iso<" in kernel_name


def is_per_surface(kernel_name):
    return "k_propagate" in kernel_name or "k_interact_" in kernel_name or "k_surface_step" in kernel_name


def config_of_march(kernel_name):
    """fall-back when the counter file has no dispatch ids: which bench config a march launch belongs to, from its
    instantiation: k_trace_general -> aniso; k_trace_iso<MODE, VEC_IN, VEC_OUT, SHAPES, MOMENTS, UNI, IMG> with
    SHAPES 1 / 2 -> asphere / xypoly, SHAPES 0 -> doublegauss (uniform first segment) or benchmark (arrays)"""
    if "k_trace_general<" in kernel_name:           # <MODE, GENERAL, ...>: GENERAL = the biaxial (quartic) instantiation
        g = re.search(r"k_trace_general<\s*\d+\s*,\s*(\w+)", kernel_name)
        return "aniso_biaxial" if g and g.group(1) in ("1", "true") else "aniso"
    m = re.search(r"k_trace_iso<\s*\d+\s*,\s*\w+\s*,\s*\w+\s*,\s*(\d+)\s*,\s*\w+\s*,\s*(\w+)", kernel_name)
    if m:
        sh = int(m.group(1))
        if sh == 0:
            return "doublegauss" if m.group(2) in ("1", "true") else "benchmark"
        return {1: "asphere", 2: "xypoly"}.get(sh)
    return None


def rows_by_config(rows, configs, launches_per_trace=None):
    """[(config, counter, value)] for the kernels that count.  ``launches_per_trace``: {config: kernels per trace} for
    the per-surface configs (marches: 1).  With dispatch ids the rows are cut along the dispatch order; without
    them the marches are told apart by their instantiation (per-surface rows are then dropped)."""
    per = dict(launches_per_trace or {})
    want = [r for r in rows if is_march(r.get("Kernel_Name", "")) or
            (per and is_per_surface(r.get("Kernel_Name", "")))]
    if want and all(str(r.get("Dispatch_Id", "")).strip().isdigit() for r in want):
        ids = sorted(set(int(r["Dispatch_Id"]) for r in want))
        need = [PMC_LAUNCHES * per.get(c, 1) for c in configs]
        if len(ids) == sum(need):
            owner = {}
            pos = 0
            for (c, k) in zip(configs, need):
                for d in ids[pos:pos + k]:
                    owner[d] = c
                pos += k
            return [(owner[int(r["Dispatch_Id"])], r["Counter_Name"], float(r["Counter_Value"])) for r in want]
    return [(config_of_march(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"]))
            for r in want if is_march(r["Kernel_Name"])]


def measure_live(configs, args, rays_of, timeout_s, launches_per_trace=None, bench_script=None):
    """(traffic, flops): per config the HBM bytes and FP64 flops of one trace, from rocprofv3 PMC passes over
    ``bench.py --pmc-inner``.  Per-surface configs get the traffic passes only."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not found"}, {}
    tmp = tempfile.mkdtemp(prefix="prt_pmc_", dir="/tmp")
    # No arena under the profiler: the bytes and instructions of a launch do not depend on where its arrays lie, and
    # with counters attached every probe launch of a hunt costs milliseconds.
    env = dict(os.environ, TMPDIR="/tmp", PRT_ARENA="off")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    script = bench_script or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    per = dict(launches_per_trace or {})
    sums = {}
    t_end = time.perf_counter() + timeout_s
    try:
        for (p, counters) in enumerate(PMC_PASSES):
            # (the instruction counters of a 24-launch sweep say nothing the march's do not)
            cfgs = list(configs) if p < 2 else [c for c in configs if c not in PER_SURFACE_CONFIGS]
            left = t_end - time.perf_counter()
            if left < 10:
                return {"error": "PMC passes ran out of their time budget (%d s)" % timeout_s}, {}
            out_dir = os.path.join(tmp, "pass%d" % p)
            cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out_dir, "--",
                   sys.executable, script, "--pmc-inner", ",".join(cfgs),
                   "--rays-of", json.dumps(rays_of), "--first-segment", args.first_segment, "--mode", args.mode] + \
                  (["--two-mask-arrays"] if args.two_mask_arrays else [])
            res = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
            if res.returncode != 0:
                return {"error": "rocprofv3 pass %s failed (rc %d): %s"
                                 % ("+".join(counters), res.returncode, res.stderr.decode(errors="replace")[-300:])}, {}
            for path in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as fh:
                    for (cfg, cname, val) in rows_by_config(list(csv.DictReader(fh)), cfgs, per):
                        if cfg in cfgs:
                            sums.setdefault((cfg, cname), []).append(val)
    except (subprocess.TimeoutExpired, OSError) as exc:
        return {"error": "rocprofv3 PMC passes: %s" % exc}, {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic, flops = {}, {}
    for cfg in configs:
        k = per.get(cfg, 1)          # kernels per trace: the per-trace figure is the SUM over them

        def per_trace(name):
            v = sums.get((cfg, name))
            return (sum(v) / (len(v) / float(k)), len(v) // k) if v else (None, 0)
        (f, nf) = per_trace("FETCH_SIZE")
        (w, nw) = per_trace("WRITE_SIZE")
        if f is not None and w is not None:
            traffic[cfg] = {"bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0, "fetch_bytes": 2.0 * f * 1024.0,
                            "write_bytes": w * 1024.0, "launches": [nf, nw], "kernels_per_trace": k,
                            "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc "
                                      "WRITE_SIZE (separate passes, %d traces each) over `bench.py --pmc-inner`; "
                                      "2 x FETCH_SIZE (gfx950 counts the 128-B requests of 16 B/lane coalesced "
                                      "reads at 64 B) + WRITE_SIZE, in KiB" % nf}
        c = {n: per_trace(n)[0] for n in PMC_PASSES[2]}
        if all(v is not None for v in c.values()):
            flops[cfg] = {"flops_per_launch": 64.0 * (2.0 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
                                                      + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]),
                          "valu_wave_instructions": c["SQ_INSTS_VALU"],
                          "source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 "
                                    "SQ_INSTS_VALU over `bench.py --pmc-inner` (2 flop per FMA, 64 lanes per wave "
                                    "instruction)"}
    if not traffic:
        return {"error": "no march launches found in the rocprofv3 counter files"}, flops
    return traffic, flops
