#!/usr/bin/env python
"""
bench.py -- ray-surface-ops/s of the sequential-raytrace hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of OpticalSystem.seqtrace over one bundle: BASELINE.json
configs[1], the 12-surface Rudolph double Gauss (12 spherical Conic surfaces,
ConstantIndexGlass d-line indices), ~1e7 rays PER GPU (RectGrid disk raster,
collimated on-axis), float64, full path materialised (hit point, outgoing wave
vector and validity at every surface written to HBM).  Inputs are resident in
HBM before the timed region.  For N > 1 the bundle of N x 1e7 rays is sharded by
rays (weak scaling, no data-path collective) and every step ends with the
image-plane all-gather (RCCL), issued on a side stream so that it overlaps the
next step's trace; the timed region ends when everything has completed.

metric: ray-surface-ops/s = rays x surfaces / seconds (the reference's own
definition, demos/demo_benchmark.py:82-85).

Prints ONE JSON line (rank 0).
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

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PREWARM_LAUNCHES = 30


def algorithmic_bytes(n_rays, n_surfaces, with_e0=True, record_bytes=49):
    """HBM bytes the fused path-mode march must move per launch:
    read x0, k0 (+E0) once: 48 (+24) B/ray; write per surface x_hit 24 + k_out 24 + one byte holding
    both masks (valid | valid_out << 1) = 49 B/ray -- SURVEY 8d's ray-surface record -- or 50 B with
    the masks in two separate arrays."""
    return n_rays * ((72 if with_e0 else 48) + record_bytes * n_surfaces)


def cpu_baseline(records, o, k, e0, n_all=None, chunk=100_000):
    """CPU restatements of the reference algorithm (test oracles, geometry only -- i.e. WITHOUT
    the reference's SVD E-field step that is 91% of its time) on a bounded sample of the same
    workload, timed on this box's host cores:
      value: C / OpenMP port (oracle/seqtrace_c.c) on all host threads
      numpy_single_core: the NumPy port (oracle/seqtrace_np.py), one process
      with_svd_efield: NumPy port incl. the SVD E-field step, the reference's true cost profile"""
    from oracle import seqtrace_np as oracle
    from oracle import seqtrace_c
    S = len(records)
    n = o.shape[1]
    ws = seqtrace_c.Workspace(S, n)                   # outputs allocated and touched once
    seqtrace_c.trace_arrays(records, o, k, e0, workspace=ws)                            # warm-up
    # the port is memory bound on the host; pick the best of a few thread counts, then time it
    nmax = seqtrace_c.load().seqtrace_c_threads()
    best = (None, 0.0)
    for nt in sorted(set(max(1, nmax // q) for q in (1, 2, 4, 8))):
        t0 = time.perf_counter()
        seqtrace_c.trace_arrays(records, o, k, e0, nthreads=nt, workspace=ws)
        rate = n * S / (time.perf_counter() - t0)
        if rate > best[1]:
            best = (nt, rate)
    reps = 0
    t0 = time.perf_counter()
    while True:
        (_, _, _, _, used) = seqtrace_c.trace_arrays(records, o, k, e0, nthreads=best[0], workspace=ws)
        reps += 1
        dt_c = time.perf_counter() - t0
        if dt_c > 5.0 or reps >= 30:
            break
    m_np = min(n, 1_000_000)
    t1 = time.perf_counter()
    done = 0
    with np.errstate(all="ignore"):
        while done < m_np:
            hi = min(done + chunk, m_np)
            oracle.trace(records, o[:, done:hi], k[:, done:hi], e0[:, done:hi])
            done = hi
    dt_np = time.perf_counter() - t1
    m_e = min(n, 50_000)
    t2 = time.perf_counter()
    with np.errstate(all="ignore"):
        oracle.trace(records, o[:, :m_e], k[:, :m_e], e0[:, :m_e], with_efield=True)
    dt_e = time.perf_counter() - t2
    return {"value": reps * n * S / dt_c, "unit": "ray-surface-ops/s", "cores": used, "kind": "port",
            "sample": "C/OpenMP oracle: %d x (first %d of the %d rays x %d surfaces, path written to "
                      "host RAM), %.1f s" % (reps, n, n_all or n, S, dt_c),
            "numpy_single_core": {"value": m_np * S / dt_np, "sample": "%d rays in chunks of %d, %.1f s"
                                  % (m_np, chunk, dt_np)},
            "with_svd_efield": {"value": m_e * S / dt_e, "sample": "%d rays, %.1f s" % (m_e, dt_e)},
            "host_cpus": os.cpu_count()}


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


def main():
    _stdout_of_other_ranks_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=None,
                    help="requested rays per GPU (default: 1e7 at N = 1 = BASELINE configs[1]; 1.25e7 at N > 1, "
                         "so that 8 GPUs trace the 1e8-ray bundle of configs[4])")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the final image-plane all-gather")
    ap.add_argument("--no-stats", action="store_true", help="N>1: skip the per-step spot statistics all-reduce")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: all-gather the image plane (49 B/ray) after every step instead of once")
    ap.add_argument("--two-pass-stats", action="store_true",
                    help="N>1: per-step spot statistics from two extra passes over the image plane and two "
                         "all-reduces (default: moments reduced inside the trace kernel, one all-reduce)")
    ap.add_argument("--placement-candidates", type=int, default=12,
                    help="output allocations to choose from by timing the march into each (1 = take the first)")
    ap.add_argument("--two-mask-arrays", action="store_true",
                    help="write valid and valid_out as two byte arrays (50 B per record) instead of one "
                         "byte of packed flags (49 B, default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["path", "image"], default="path")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: dry run of the multi-rank path on ONE GPU (all ranks share cuda:0, "
                         "the gather is staged through host memory); not a measurement")
    ap.add_argument("--force-multi", action="store_true",
                    help="run the N>1 code path (5 wavelengths, side-stream all-reduces, final all-gather) "
                         "with whatever world size there is, also 1: RCCL smoke test on a 1-GPU box")
    args = ap.parse_args()
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
    if args.rays is None:
        args.rays = 10_000_000 if (world == 1 and not args.force_multi) else 12_500_000
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_multi
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gpus = world

    from pyrate_amd import engine, systems, _lib
    from pyrate_amd import distributed as pdist

    # N = 1: BASELINE configs[1] (d line).  N > 1: configs[4] -- the same lens at the five
    # wavelengths of the prescription (spd:5), per-wavelength indices from the Conrady fit
    # through the (d, F, C) indices; step i traces wavelength i % 5.
    records = systems.double_gauss_records()
    S = len(records)
    if use_dist:
        record_sets = [systems.double_gauss_records(w) for w in systems.DOUBLE_GAUSS_WAVES_MM]
    else:
        record_sets = [records]
    # the global bundle: n_gpus x rays (RectGrid disk raster, generated on the device,
    # bit-identical to the NumPy raster of the reference); rank r owns a contiguous slice
    (_, n_total) = engine.rect_grid_count(args.rays * n_gpus, dev)
    (lo, hi) = pdist.shard_range(n_total, rank, n_gpus)
    n_local = hi - lo
    (x0, k0, e0d, _) = systems.double_gauss_bundle_device(args.rays * n_gpus, dev, lo=lo, hi=hi)

    sysds = [engine.DeviceSystem(r, local_rank) for r in record_sets]
    sysd = sysds[0]
    mode = _lib.MODE_PATH if args.mode == "path" else _lib.MODE_IMAGE
    multi = use_dist
    # N > 1 (BASELINE configs[4]): every step ends with the image-plane spot statistics of the
    # sharded bundle (an all-reduce of moments instead of moving the image plane).
    # The trace kernel reduces its shard's moments itself (prt_trace_moments, no extra pass over
    # the arrays); the one 7-double all-reduce runs on a side stream and overlaps the next step's
    # trace.  (--two-pass-stats: the generic form, two reduction passes + two all-reduces.)
    # The full image plane is all-gathered ONCE after the K timed steps ("the final
    # image-plane gather"); it is not a step, so it is timed separately and reported as
    # config.image_plane_exchange.final_gather_ms.  --gather-every-step moves the 49 B/ray
    # all-gather into every (timed) step instead.
    do_stats = multi and not args.no_stats
    do_final_gather = multi and not args.no_gather and not args.gather_every_step
    do_step_gather = multi and args.gather_every_step
    fused_stats = do_stats and not args.two_pass_stats
    nbuf = 2 if (do_stats or do_step_gather) else 1          # in-flight side-stream jobs
    # the side stream of the fused form only touches 7-double vectors, so the 6-GB path buffers need
    # no double buffering (alternating between two of them costs ~6 % write bandwidth, measured:
    # scratch/moments_cost.py)
    n_out_bufs = 1 if (fused_stats and not do_step_gather) else nbuf
    packed = not args.two_mask_arrays
    record_bytes = 49 if packed else 50
    # Output placement: the march's write bandwidth depends reproducibly on where x_hit and k_out sit
    # in HBM relative to each other (DESIGN.md section 5 "placement"), so -- like any caller that
    # re-uses its output arrays -- the bench lets the engine choose the pair from a small pool by
    # timing (untimed set-up, after a device wake-up).  --placement-candidates 1 = first allocation.
    placement = None
    if args.placement_candidates > 1 and n_out_bufs == 1:
        warm = sysd.alloc_outputs(n_local, mode, packed_flags=packed)
        for _ in range(PREWARM_LAUNCHES):
            sysd.trace_into(x0, k0, warm, e0d)
        torch.cuda.synchronize()
        del warm
        (b0, placement) = sysd.alloc_outputs_tuned(x0, k0, e0d, mode=mode, packed_flags=packed,
                                                   candidates=args.placement_candidates)
        bufs = [b0]
        torch.cuda.empty_cache()
    else:
        bufs = [sysd.alloc_outputs(n_local, mode, packed_flags=packed) for _ in range(n_out_bufs)]
    host_staged = (args.backend == "gloo")
    stats = [pdist.SpotStatistics(dev, n_rays=n_local) for _ in range(nbuf)] if do_stats else []
    gathers = [pdist.ImagePlaneGather(n_total, dev, stage_on_host=host_staged)
               for _ in range(nbuf if do_step_gather else 1)] if (do_step_gather or do_final_gather) else []
    comm_stream = torch.cuda.Stream(device=dev) if multi else None
    main_stream = torch.cuda.current_stream(dev)
    side_done = [None] * nbuf          # event: side-stream work on buffer pair b has finished

    def step(i):
        b = i % nbuf
        if side_done[b] is not None:
            main_stream.wait_event(side_done[b])      # buffer pair b is free again
        ob = bufs[b % n_out_bufs]
        if fused_stats:
            stats[b].trace_and_start(sysds[i % len(sysds)], x0, k0, ob, e0d)
        else:
            sysds[i % len(sysds)].trace_into(x0, k0, ob, e0d)
        if do_stats or do_step_gather:
            ev = torch.cuda.Event()
            ev.record(main_stream)
            v = sysd.views(ob)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                if fused_stats:
                    stats[b].reduce()
                elif do_stats:
                    stats[b].start(v.x_hit[-1], v.valid_out[-1])
                if do_step_gather:
                    gathers[b].wait()
                    gathers[b].start(v.x_hit[-1], v.k_out[-1], v.valid_out[-1])
                    gathers[b].wait()
                done = torch.cuda.Event()
                done.record(comm_stream)
                side_done[b] = done

    def final_gather(last_step):
        """the one-off image-plane all-gather of the last traced bundle (49 B/ray)"""
        v = sysd.views(bufs[(last_step % nbuf) % n_out_bufs])
        ev = torch.cuda.Event()
        ev.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            gathers[0].start(v.x_hit[-1], v.k_out[-1], v.valid_out[-1])
            gathers[0].wait()
        comm_stream.synchronize()

    def finish(last_step):
        """every step's work (trace + per-step exchange) has completed"""
        if comm_stream is not None:
            comm_stream.synchronize()
        torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    # device wake-up (not one of the W warm-up steps): after idle the first ~25 ms of launches run
    # at ramping clocks (per-launch trace in DESIGN.md section 5); 30 plain launches of the same
    # kernel bring the chip to its steady state before anything is counted
    for _ in range(PREWARM_LAUNCHES):
        sysd.trace_into(x0, k0, bufs[0], e0d)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    finish(args.warmup - 1)
    if do_final_gather and args.warmup > 0:
        final_gather(args.warmup - 1)          # warms the all-gather path too
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish(args.steps - 1)
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=(dev if args.backend == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the final image-plane gather is not one of the K steps: timed on its own, reported beside
    final_gather_ms = None
    if do_final_gather:
        barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        final_gather(args.steps - 1)
        barrier()
        final_gather_ms = (time.perf_counter() - tg) * 1e3
    spot = None
    if do_stats:
        (cnt, cen, rms) = stats[(args.steps - 1) % nbuf].result()
        spot = {"rays": float(cnt), "centroid_mm": [float(c) for c in cen], "rms_spot_mm": rms}

    # dominant kernel: average launch duration from HIP events on the launch stream
    kernel_ms = sysd.trace_timed(x0, k0, bufs[0], max(args.steps, 5), e0d)
    torch.cuda.synchronize()

    ops_total = n_total * S * args.steps
    value = ops_total / elapsed
    if rank == 0:
        if args.mode == "path":
            alg = algorithmic_bytes(n_local, S, record_bytes=record_bytes)
        else:
            alg = n_local * (72 + record_bytes)
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get("%s_%d%s" % (args.mode, n_local, "" if packed else "_two_masks"))
                if ent:
                    traffic = ent["bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "ray_surface_ops_per_s", "value": value, "unit": "ray-surface-ops/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("demo_doublegauss: 12 spherical Conic surfaces (Rudolph 1897 "
                                    "double Gauss, ConstantIndexGlass d-line), RectGrid disk bundle, "
                                    "BASELINE configs[1]") if n_gpus == 1 else
                                   ("demo_doublegauss: 12 spherical Conic surfaces (Rudolph 1897 double "
                                    "Gauss), 5 wavelengths cycled (Conrady indices), RectGrid disk "
                                    "bundle ray-sharded over the GPUs (1.25e7 rays per GPU: the 1e8-ray bundle at "
                                    "8 GPUs), BASELINE configs[4]"),
                       "rays_per_gpu": n_local, "rays_total": n_total, "surfaces": S,
                       "mode": args.mode, "record_bytes": record_bytes,
                       "masks": "valid | valid_out << 1 in one byte" if packed else "two byte arrays",
                       "sharding": "rays" if n_gpus > 1 else "none",
                       "wavelengths": len(sysds), "prewarm_launches": PREWARM_LAUNCHES,
                       "output_placement": ({"policy": "x_hit / k_out pair chosen from a pool of %d arrays by timing "
                                                       "the march during set-up" % args.placement_candidates,
                                             "first_pair_ms": round(placement["first_pair_ms"], 4),
                                             "best_pair_ms": round(placement["best_pair_ms"], 4),
                                             "k_scan_ms": [round(t, 4) for t in placement["k_scan_ms"]],
                                             "x_scan_ms": [round(t, 4) for t in placement["x_scan_ms"]],
                                             "k_rescan_ms": [round(t, 4) for t in placement["k_rescan_ms"]]}
                                            if placement else {"policy": "first allocation"}),
                       "image_plane_exchange": {
                           "per_step": (("spot moments reduced inside the trace kernel + one 7-double all-reduce "
                                         "(side stream)" if fused_stats else
                                         "device spot statistics (two passes) + two 7-double all-reduces, overlapped")
                                        if do_stats else ("image-plane all-gather 49 B/ray" if do_step_gather
                                                          else "none")),
                           "final": ("image-plane all-gather 49 B/ray, once after the K timed steps"
                                     if do_final_gather else "none"),
                           "final_gather_ms": final_gather_ms,
                           "backend": ("rccl" if args.backend == "nccl" else "gloo dry run (host staged)")
                           if multi else "none"},
                       "image_plane_spot": spot},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_trace_iso", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg,
                         "bytes_per_ray_surface_op": alg / (n_local * S),
                         "frac_at_98B_per_op_convention": (n_local * S * 98 / (kernel_ms * 1e-3) / 1e9) / HBM_PEAK_GBS},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            m = min(n_local, 4_000_000)
            out["cpu_baseline"] = cpu_baseline(records, x0[:, :m].cpu().numpy(), k0[:, :m].cpu().numpy(),
                                               e0d[:, :m].cpu().numpy(), n_all=n_local)
        else:
            out["cpu_baseline"] = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio
        # (block-buffered on a pipe, so it would otherwise surface at exit, after the line)
        _flush_c_stdio()
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
