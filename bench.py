#!/usr/bin/env python
"""
bench.py -- ray-surface-ops/s of the sequential-raytrace hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of OpticalSystem.seqtrace over one bundle.  Headline workload: BASELINE.json configs[1], the
12-surface Rudolph double Gauss (12 spherical Conic surfaces, ConstantIndexGlass d-line indices), ~1e7 rays PER GPU
(RectGrid disk raster, collimated on-axis), float64, full path materialised (hit point, outgoing wave vector and
validity at every surface written to HBM).  Inputs are resident in HBM before the timed region.  metric:
ray-surface-ops/s = rays x surfaces / seconds (the reference's own definition, demos/demo_benchmark.py:82-85).

Output contract: the LAST line of stdout is ONE compact JSON record (< 8 KB, asserted) -- metric, value, unit, n_gpus,
steps, warmup, ms_per_step, dtype, config{workload, rays, surfaces, mode, build, configs_summary}, roofline{bound,
achieved, peak, frac, traffic, traffic_ratio_to_algorithmic, kernel, kernel_ms, algorithmic_bytes_per_launch},
cpu_baseline{value, cores, host_cpus, kind, sample}, verified{ok, max_rel_x, max_abs_k, n_checked}, scaling_point{rays,
ms, frac, ok}, e2e{...}.  Everything else (one full record per configuration, prose, arena statistics) goes to
``bench_detail.json`` beside this script (and to gpurun_out/ when that directory exists); nothing else is written to
stdout.

At N = 1 the default run measures, under the same contract, the other single-GPU configurations (configs[2] even
asphere, configs[3] crystal doublet, an XY-polynomial system, the reference's OWN benchmark workload) and the other
shipped paths (biaxial crystals, the per-surface crystal march, the plugin-granular calls, image mode with fused
moments): ``config.configs_summary = {name: [ms per step, fraction of its roof, verified, traffic / algorithmic]}``;
then the 1e8-ray bundle of the multi-GPU protocol on this one GPU (``scaling_point``) and the end-to-end time of the
drop-in call with host arrays in (``e2e``).  Every configuration is VERIFIED on the arrays its timed launches wrote
(benchmarks/verify.py; never inside a timed region); the script exits 4 when a deviation exceeds 1e-10.  HBM traffic
(roofline.traffic) is measured in the same run by rocprofv3 PMC passes (benchmarks/pmc.py).

For N > 1 (configs[4]) ONE bundle of 1e8 rays is sharded by rays over the GPUs (``--scaling strong``; no collective in
the trace), the five prescription wavelengths are cycled over the steps, and every step ends with that wavelength's
spot statistics (one 7-double all-reduce) and its image-plane all-gather (49 B/ray) on a side stream, overlapping the
next trace (benchmarks/multirank.py).  A watchdog prints a JSON line with an ``error`` field instead of hanging.

The parts: benchmarks/workloads.py (configurations), configs.py (measurements), verify.py, cpu_baseline.py, pmc.py,
multirank.py.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The arena's hunt for its kinds of HBM is bounded per call in the library (32 slabs / 50 ms: a product's first call
# must not cost 150 ms); a benchmark wants the steady-state placement from its first allocation on, so it lifts
# the bound -- and says so (bench_detail.json: arena.hunt).  Set before libprt reads it.
os.environ.setdefault("PRT_ARENA_HUNT", "full")

import torch
import torch.distributed as dist

from benchmarks import pmc
from benchmarks.multirank import Watchdog, run_multi
from benchmarks.workloads import (SECONDARY_CUSTOM_CONFIGS, SECONDARY_MARCH_CONFIGS, SHORT_WORKLOAD, SINGLE_GPU_CONFIGS,
                                  VERIFY_TOL)

MAX_LINE_BYTES = 8192          # the driver keeps a bounded tail of stdout: a longer line cannot be parsed (round 5)
T_START = time.perf_counter()


def _flush_c_stdio():
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _stdout_of_other_ranks_to_stderr():
    """under torch.distributed.run every rank shares one stdout pipe; only rank 0 reports, so what
    libraries print on the other ranks (RCCL's banner) goes to stderr instead"""
    if int(os.environ.get("RANK", "0")) != 0:
        sys.stdout.flush()
        os.dup2(2, 1)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (200 steps = 0.2 s of device time at N = 1: a single scheduling hiccup of the host -- one was seen to cost
    # 10 ms of a 50-ms timed region -- no longer moves the rate by more than a few per cent)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=list(SINGLE_GPU_CONFIGS) + list(SECONDARY_MARCH_CONFIGS), default=None,
                    help="measure this configuration alone, as the headline: BASELINE.json configs[1] (doublegauss), "
                         "configs[2] (asphere), configs[3] (aniso), the XY-polynomial system (xypoly) ...")
    ap.add_argument("--headline-only", action="store_true", help="N = 1: do not measure the other configurations")
    ap.add_argument("--configs", default=None, help="N = 1: measure exactly these configurations (comma-separated), the "
                                                    "first one as the headline -- for experiments")
    ap.add_argument("--rays", type=int, default=None,
                    help="requested rays per GPU (default: 1e7 at N = 1; N > 1 with --scaling weak: 1.25e7)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong (default) = ONE bundle of --rays-total rays split over the N GPUs; weak = --rays "
                         "rays per GPU")
    ap.add_argument("--rays-total", type=int, default=None, help="N > 1, --scaling strong: rays of the bundle (default 1e8)")
    ap.add_argument("--no-scaling-point", action="store_true", help="N = 1: skip the 1e8-ray bundle on this one GPU")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the end-to-end timing of the drop-in call")
    ap.add_argument("--first-segment", choices=["uniform", "arrays"], default="uniform",
                    help="how the collimated bundle's k0 / E0 reach the march: one vector each (default; 24 B/ray of "
                         "loads) or per-ray arrays (72 B/ray)")
    ap.add_argument("--exchange", choices=["auto", "gather", "gather-direct", "stats", "final-gather", "none"],
                    default="auto",
                    help="N > 1, what every step ends with: gather = spot-statistics all-reduce + in-place RCCL all-gather "
                         "of the image plane; gather-direct = the same exchange as peer writes into IPC-mapped receive "
                         "buffers; auto (default) = whichever of the two a 3-step probe at start-up finds faster on this "
                         "node (gather where only it can run); stats = the all-reduce only; final-gather = all-reduce "
                         "per step, ONE all-gather after the K steps; none = the bare sharded trace")
    ap.add_argument("--gather-mode", choices=["inplace", "copy"], default="inplace")
    ap.add_argument("--two-pass-stats", action="store_true")
    ap.add_argument("--placement", choices=["arena", "torch"], default="arena",
                    help="where the path arrays come from: the engine's placement-aware arena (default) or torch")
    ap.add_argument("--inputs", choices=["arena", "torch"], default="arena")
    ap.add_argument("--two-mask-arrays", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip aniso_biaxial, aniso_chain, plugin, image_moments")
    ap.add_argument("--cpu-budget", type=float, default=4.0, help="seconds of C-port timing per configuration")
    ap.add_argument("--traffic", choices=["auto", "live", "lookup", "none"], default="auto",
                    help="roofline.traffic: live = rocprofv3 PMC passes in this run (auto: when rocprofv3 is there)")
    ap.add_argument("--traffic-timeout", type=float, default=150.0)
    ap.add_argument("--mode", choices=["path", "image"], default="path")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: dry run of the multi-rank path on ONE GPU (host-staged gather); not a measurement")
    ap.add_argument("--force-multi", action="store_true", help="run the N > 1 code path with whatever world size there is")
    ap.add_argument("--trace-stream", choices=["default", "new", "high"], default="new")
    ap.add_argument("--watchdog", type=float, default=None,
                    help="seconds after which a JSON line with an `error` field is printed and the process exits "
                         "(default: 900 for N > 1, off at N = 1; PRT_BENCH_WATCHDOG overrides)")
    ap.add_argument("--detail", default=None, help="where the full records go (default: bench_detail.json beside bench.py)")
    ap.add_argument("--pmc-inner", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--rays-of", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def _r(v, digits=5):
    """a number as the compact line carries it: 5 significant digits"""
    if not isinstance(v, float):
        return v
    return float("%.*g" % (digits, v)) if v == v and abs(v) != float("inf") else None


def compact_single(base, head, recs, scaling_point, e2e, arena_stats, build, wall):
    """the N = 1 line: numbers only, no prose (the prose is in bench_detail.json)"""
    rf = head["roofline"]
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio_to_algorithmic", "kernel", "kernel_ms",
            "algorithmic_bytes_per_launch", "bytes_per_ray_surface_op", "ops_vs_98B_convention")
    roofline = {k: _r(rf.get(k), 6) for k in keep if k in rf or k in ("traffic", "traffic_ratio_to_algorithmic")}
    sec = rf.get("secondary")
    if sec:
        roofline["secondary"] = {k: _r(sec.get(k)) for k in ("bound", "frac", "valu_issue_frac") if k in sec}
    cb = head.get("cpu_baseline")
    cpu = None
    if cb:
        cpu = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "host_cpus", "kind")}
        cpu["sample"] = str(cb.get("sample", ""))[:160]
        for extra in ("numpy_single_core", "with_svd_efield"):
            if cb.get(extra):
                cpu[extra] = _r(cb[extra]["value"])
    vf = head["verified"]
    verified = {k: _r(vf.get(k)) for k in ("ok", "tolerance", "max_resid", "max_rel_x", "max_abs_k", "n_checked")}
    if vf.get("oracle_sample"):
        verified["oracle_rays"] = vf["oracle_sample"]["rays"]
        verified["mask_mismatches"] = vf["oracle_sample"]["mask_mismatches"]
    summary = {}
    for r in recs:
        if "ms_per_step" not in r:
            continue
        rr = r["roofline"]
        summary[r["name"]] = [_r(r["ms_per_step"], 4), _r(rr.get("frac"), 4), bool(r["verified"]["ok"]),
                              _r(rr.get("traffic_ratio_to_algorithmic"), 4)]
    op = head.get("output_placement") or {}
    cfg = {"workload": SHORT_WORKLOAD.get(head["name"], head["name"]), "rays_per_gpu": head["rays"], "rays_total": head["rays"],
           "surfaces": head["surfaces"], "mode": head["mode"],
           "first_segment": "uniform" if str(head.get("first_segment", "")).startswith("uniform") else "arrays",
           "record_bytes": head.get("record_bytes"), "sharding": "none", "wavelengths": 1,
           "placement": {"policy": op.get("policy"), "kinds": op.get("memory_kinds_of_x_hit_and_k_out"),
                         "input_kind": op.get("memory_kind_of_inputs"), "note": op.get("note")},
           "build": build.get("libprt_sha256_16"),
           "configs_summary_columns": ["ms_per_step", "frac_of_roof", "verified", "traffic_over_algorithmic"],
           "configs_summary": summary, "wall_s": _r(time.perf_counter() - T_START, 4), "detail": "bench_detail.json"}
    line = dict(base, value=_r(head["value"], 6), ms_per_step=_r(head["ms_per_step"], 6), config=cfg, roofline=roofline,
                cpu_baseline=cpu, verified=verified)
    if scaling_point is not None:
        if "error" in scaling_point:
            line["scaling_point"] = {"error": scaling_point["error"][:200]}
        else:
            line["scaling_point"] = {"rays": scaling_point["rays_total"], "ms": _r(scaling_point["ms_per_step"]),
                                     "frac": _r(scaling_point["hbm_frac"]), "ok": bool(scaling_point["verified"]["ok"]),
                                     "value": _r(scaling_point["value"])}
            summary["scaling_point_1e8_rays"] = [_r(scaling_point["ms_per_step"], 4), _r(scaling_point["hbm_frac"], 4),
                                                 bool(scaling_point["verified"]["ok"]), None]
    if e2e is not None:
        if "error" in e2e:
            line["e2e"] = {"error": e2e["error"][:200]}
        else:
            line["e2e"] = {k: _r(e2e[k], 4) for k in ("rays", "first_call_ms", "h2d_ms", "seqtrace_call_ms",
                                                      "seqtrace_device_bundle_ms", "last_bundle_to_host_ms",
                                                      "full_path_to_host_ms")}
            line["e2e"]["small_bundle_call_us"] = {k: _r(v, 4) for (k, v) in e2e["small_bundle_call"].items()}
    return line


def write_detail(path, detail):
    """the full records beside the script (and into gpurun_out/ when the run is a gpurun call)"""
    targets = [path or os.path.join(ROOT, "bench_detail.json")]
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        targets.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    for t in targets:
        try:
            with open(t, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as exc:
            print("bench.py: could not write %s: %s" % (t, exc), file=sys.stderr)


def emit(fd, line):
    """ONE line, last on stdout, short enough for the driver to keep it whole"""
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= MAX_LINE_BYTES:
        # never print a line the driver cannot read: drop the optional parts, say so
        for k in ("e2e", "scaling_point"):
            line.pop(k, None)
        line["config"] = {k: v for (k, v) in line["config"].items() if k in ("workload", "rays_per_gpu", "rays_total",
                                                                             "surfaces", "mode", "build", "detail")}
        line["truncated"] = "the record outgrew %d bytes; see bench_detail.json" % MAX_LINE_BYTES
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < MAX_LINE_BYTES, len(text)
    _flush_c_stdio()
    sys.stdout.flush()
    os.write(fd, (text + "\n").encode())
    sys.stdout.flush()


def main():
    _stdout_of_other_ranks_to_stderr()
    args = parse_args()
    if args.force_multi:
        os.environ["PRT_FORCE_COLLECTIVES"] = "1"
        for (k, v) in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29512")):
            os.environ.setdefault(k, v)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_multi:
        # started as a plain `python bench.py --gpus N`: become the one-process-per-GPU launch the
        # contract describes (torch.distributed.run on this node, rendezvous on 127.0.0.1)
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", os.environ.get("MASTER_PORT", "29577"),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or args.force_multi
    if use_dist and args.config not in (None, "doublegauss"):
        raise SystemExit("--config %s is a 1-GPU workload (BASELINE configs[4] is the double Gauss)" % args.config)
    headline = args.config or "doublegauss"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    try:        # no core file of a process with hundreds of GiB of device memory mapped
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, resource.getrlimit(resource.RLIMIT_CORE)[1]))
    except (ImportError, ValueError, OSError):
        pass
    if args.pmc_inner:
        args.rays_of = json.loads(args.rays_of)
        pmc.inner(args, dev)
        return

    exit_code = 0
    json_fd = [1]
    base_line = {"metric": "ray_surface_ops_per_s", "unit": "ray-surface-ops/s", "n_gpus": world, "steps": args.steps,
                 "warmup": args.warmup, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                 "dtype": "f64", "data": "synthetic"}
    wd_s = args.watchdog if args.watchdog is not None else (900.0 if use_dist else 0.0)
    if os.environ.get("PRT_BENCH_WATCHDOG"):
        wd_s = float(os.environ["PRT_BENCH_WATCHDOG"])
    watchdog = Watchdog(wd_s, rank, json_fd, base_line)
    # Libraries write to STDOUT behind our back (RCCL's version banner whenever a communicator comes up; rocprofv3's
    # tool library): the contract is ONE JSON line, last on stdout.  So file descriptor 1 points at stderr for the whole
    # run and the line goes to a duplicate of the original stdout at the end.
    sys.stdout.flush()
    json_fd[0] = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        watchdog.stage = "init_process_group"
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from pyrate_amd import build as prt_build, _lib
    if use_dist:
        (line, detail) = run_multi(args, dev, world, rank, local_rank, watchdog)
        if line is not None and not (line["verified"]["ok"] and line["verified"]["all_ranks_ok"]):
            line["error"] = "verification failed (deviation above %g)" % VERIFY_TOL
            exit_code = 4
    else:
        from benchmarks import configs
        (recs, scaling_point, e2e, arena_stats, wall) = configs.run_single_gpu(args, dev, watchdog, headline)
        build = prt_build.build_info(_lib.LIB_PATH)
        # (the full records are on disk before anything is condensed: a bug in the condensing loses nothing measured)
        raw = dict(configs=recs, scaling_point=scaling_point, e2e=e2e, arena=arena_stats, build=build, wall_s_by_stage=wall)
        write_detail(args.detail, raw)
        line = compact_single(base_line, recs[0], recs, scaling_point, e2e, arena_stats, build, wall)
        bad = [r["name"] for r in recs if not r["verified"]["ok"]]
        if scaling_point and "error" not in scaling_point and not scaling_point["verified"]["ok"]:
            bad.append("scaling_point")
        if bad:
            line["error"] = "verification failed (deviation above %g or a mask mismatch): %s" % (VERIFY_TOL, ", ".join(bad))
            exit_code = 4
        detail = dict(raw, compact_line=line)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.done()
    if rank == 0:
        write_detail(args.detail, detail)
        emit(json_fd[0], line)
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
