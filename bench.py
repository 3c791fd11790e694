#!/usr/bin/env python
"""
bench.py -- ray-surface-ops/s of the sequential-raytrace hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of OpticalSystem.seqtrace over one bundle.  Default workload:
BASELINE.json configs[1], the 12-surface Rudolph double Gauss (12 spherical Conic
surfaces, ConstantIndexGlass d-line indices), ~1e7 rays PER GPU (RectGrid disk
raster, collimated on-axis), float64, full path materialised (hit point, outgoing
wave vector and validity at every surface written to HBM).  Inputs are resident in
HBM before the timed region.  ``--config asphere`` / ``--config aniso`` run
configs[2] (even asphere, Newton intersection, 1e7 rays) and configs[3]
(anisotropic doublet, 1e6 -> 4e6 rays) under the same contract.  The path arrays
come from the engine's placement-aware arena (x_hit and k_out in two different
kinds of HBM) -- the allocation the product path uses, no scan.

For N > 1 (configs[4]) the bundle of N x 1.25e7 rays is sharded by rays (weak
scaling, no collective in the trace) and the five prescription wavelengths are
cycled over the steps; every step ends with that wavelength's spot statistics (one
7-double all-reduce) AND its image-plane all-gather (49 B/ray, RCCL), both issued
on a side stream so that they overlap the next wavelength's trace (two sets of
path arrays); the timed region ends when everything has completed.  The rate of
the same steps without the gather is measured right after and reported beside it.

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


FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz


def _lookup(fname, key):
    """a per-launch figure measured by rocprofv3 PMC passes of an EARLIER run of the same workload
    (benchmarks/hbm_traffic.py, benchmarks/valu_profile.py) -- looked up, not measured in this run"""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def cpu_baseline_numpy(records, o, k, e0, m, n_all):
    """configs the C port does not cover (explicit shapes, crystals): the NumPy oracle on the first m rays"""
    from oracle import seqtrace_np as oracle
    m = min(m, o.shape[1])
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        oracle.trace(records, o[:, :m], k[:, :m], e0[:, :m])
    dt = time.perf_counter() - t0
    return {"value": m * len(records) / dt, "unit": "ray-surface-ops/s", "cores": 1, "kind": "port",
            "sample": "NumPy oracle (oracle/seqtrace_np.py), first %d of %d rays x %d surfaces, %.1f s"
                      % (m, n_all, len(records), dt),
            "host_cpus": os.cpu_count()}


def main():
    _stdout_of_other_ranks_to_stderr()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (200 steps = 0.2 s of device time at N = 1: a single scheduling hiccup of the host -- one was seen to cost
    # 10 ms of a 50-ms timed region -- no longer moves the rate by more than a few per cent)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["doublegauss", "asphere", "aniso"], default="doublegauss",
                    help="BASELINE.json configs[1] (default, the headline), configs[2], configs[3]")
    ap.add_argument("--rays", type=int, default=None,
                    help="requested rays per GPU (default: 1e7 at N = 1 = BASELINE configs[1]/[2]; 1e6 for "
                         "--config aniso; 1.25e7 at N > 1, so that 8 GPUs trace the 1e8-ray bundle of configs[4])")
    ap.add_argument("--exchange", choices=["gather", "stats", "final-gather", "none"], default="gather",
                    help="N>1, what every step ends with: gather = spot-statistics all-reduce + the image-plane "
                         "all-gather (49 B/ray), overlapped with the next trace (default); stats = the "
                         "all-reduce only; final-gather = all-reduce per step, ONE all-gather after the K "
                         "steps (outside the timed region); none = the bare sharded trace")
    ap.add_argument("--two-pass-stats", action="store_true",
                    help="N>1: per-step spot statistics from two extra passes over the image plane and two "
                         "all-reduces (default: moments reduced inside the trace kernel, one all-reduce)")
    ap.add_argument("--placement", choices=["arena", "torch"], default="arena",
                    help="where the path arrays come from: the engine's placement-aware arena (default; what "
                         "DeviceSystem.trace uses for arrays of this size) or the torch allocator")
    ap.add_argument("--inputs", choices=["arena", "torch"], default="arena",
                    help="where the input arrays x0, k0, E0 live: in a third kind of HBM from the arena (default: "
                         "what engine.ray_rows / RayBundle do for bundles of this size) or in torch-allocated arrays")
    ap.add_argument("--two-mask-arrays", action="store_true",
                    help="write valid and valid_out as two byte arrays (50 B per record) instead of one "
                         "byte of packed flags (49 B, default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["path", "image"], default="path")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: dry run of the multi-rank path on ONE GPU (all ranks share cuda:0, "
                         "the gather is staged through host memory); not a measurement")
    ap.add_argument("--force-multi", action="store_true",
                    help="run the N>1 code path (5 wavelengths, side-stream all-reduce and all-gather) "
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
    use_dist = world > 1 or args.force_multi
    if args.config == "aniso" and use_dist:
        raise SystemExit("--config aniso is BASELINE configs[3], a 1-GPU workload")
    if args.rays is None:
        args.rays = (1_000_000 if args.config == "aniso" else
                     10_000_000 if not use_dist else 12_500_000)
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
    if os.environ.get("PRT_BENCH_WATCHDOG"):
        # debugging aid: dump every thread's stack to stderr and exit if the run takes longer than this
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PRT_BENCH_WATCHDOG"]), exit=True)
    json_fd = 1
    if use_dist:
        # RCCL prints a version banner to STDOUT whenever a communicator comes up (first collective of every
        # process group); the contract is ONE JSON line on stdout.  So file descriptor 1 points at stderr for
        # the whole run and the line goes to a duplicate of the original stdout at the end.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gpus = world

    from pyrate_amd import build as prt_build, engine, placed, systems, _lib
    from pyrate_amd import distributed as pdist

    # ---- workload ---------------------------------------------------------------------------
    # rank r owns a contiguous, equal-stride slice of the global bundle (pdist.shard_range)
    multi = use_dist
    if args.config == "doublegauss":
        # N = 1: BASELINE configs[1] (d line).  N > 1: configs[4] -- the same lens at the five
        # wavelengths of the prescription (spd:5), per-wavelength indices from the Conrady fit
        # through the (d, F, C) indices; step i traces wavelength i % 5.
        records = systems.double_gauss_records()
        record_sets = ([systems.double_gauss_records(w) for w in systems.DOUBLE_GAUSS_WAVES_MM]
                       if multi else [records])
        bundle_args = dict()
        workload = (("demo_doublegauss: 12 spherical Conic surfaces (Rudolph 1897 double Gauss, "
                     "ConstantIndexGlass d-line), RectGrid disk bundle, BASELINE configs[1]") if n_gpus == 1 else
                    ("demo_doublegauss: 12 spherical Conic surfaces (Rudolph 1897 double Gauss), 5 wavelengths "
                     "cycled (Conrady indices), RectGrid disk bundle ray-sharded over the GPUs (1.25e7 rays per "
                     "GPU: the 1e8-ray bundle at 8 GPUs), BASELINE configs[4]"))
    elif args.config == "asphere":
        # configs[2]: demo_asphere.py geometry (stop, plane front, even asphere back, image) with the
        # test-suite coefficient set (tests/test_surf_shape.py:115-127) scaled to stay in-domain, bundle
        # radius 9, 5 degree field: the Newton iteration count varies over the wavefront
        records = systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5)
        record_sets = [records]
        bundle_args = dict(rpup=9.0, z0=-5.0, field_deg=5.0)
        workload = ("demo_asphere: stop, plane, even asphere (curv -1/30, cc -1.5, A2..A6 = 1e-3, -1e-6, 1e-8; "
                    "Newton intersection), image; RectGrid disk bundle r = 9 mm at 5 deg, BASELINE configs[2]")
    else:
        c = systems.CALCITE_TILTED
        records = systems.aniso_doublet_records(
            systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
            systems.uniaxial_eps(1.6727, 1.60, (np.sin(0.2), 0.0, np.cos(0.2))))
        record_sets = [records]
        workload = ("demo_anisotropic_doublet: cemented doublet of two uniaxial crystals (calcite-like, tilted "
                    "axes), k-vector solve + ray doubling at two interfaces (1 -> 2 -> 4 rays), RectGrid disk "
                    "bundle r = 11.43 mm, BASELINE configs[3]")
    S = len(records)
    if args.config == "aniso":
        (o_h, k_h) = systems.collimated_bundle(args.rays, 11.43, -5.0)
        e_h = np.ascontiguousarray(np.cross(k_h, np.array([1., 0., 0.]), axisa=0, axisb=0).T)
        (x0, k0, e0d) = [engine.to_device_rays(a, dev, pitched=False) for a in (o_h, k_h, e_h)]
        n_total = n_local = o_h.shape[1]
        (lo, hi) = (0, n_total)
    else:
        (_, n_total) = engine.rect_grid_count(args.rays * n_gpus, dev)
        (lo, hi) = pdist.shard_range(n_total, rank, n_gpus)
        n_local = hi - lo
        (x0, k0, e0d, _) = systems.double_gauss_bundle_device(args.rays * n_gpus, dev, lo=lo, hi=hi, **bundle_args)

    sysds = [engine.DeviceSystem(r, local_rank) for r in record_sets]
    sysd = sysds[0]
    iso = sysd.all_isotropic
    mode = _lib.MODE_PATH if args.mode == "path" else _lib.MODE_IMAGE
    exchange = args.exchange if multi else "none"
    do_stats = exchange in ("gather", "stats", "final-gather")
    do_step_gather = exchange == "gather"
    do_final_gather = exchange == "final-gather"
    fused_stats = do_stats and not args.two_pass_stats
    # side-stream jobs in flight: the job of step i overlaps the trace of step i+1.  The fused
    # statistics only touch 7-double vectors; the gather reads the image-plane rows of the path
    # arrays, so with it the path arrays are double-buffered.
    nbuf = 2 if (do_stats or do_step_gather) else 1
    n_out_bufs = 2 if (do_step_gather or (do_stats and not fused_stats)) else 1
    packed = iso and not args.two_mask_arrays
    record_bytes = 49 if packed else 50
    placement = args.placement if mode == _lib.MODE_PATH else "torch"
    # one row pitch on every rank: a gathered row is read n_pad elements deep (pdist.ImagePlaneGather)
    pitch = engine.recommended_pitch(pdist.shard_stride(n_total, n_gpus)) if iso else None
    # The input arrays: big bundles are generated straight into arena memory of a kind the path arrays do
    # not use (engine.ray_rows; loads that share a kind of HBM with the march's write streams cost it
    # 5 %).  --inputs torch moves them into torch-allocated arrays instead (A/B).
    if args.inputs == "torch" and iso:
        moved = []
        for t in (x0, k0, e0d):
            buf = torch.empty((3, t.stride(0)), dtype=torch.float64, device=dev)[:, :n_local]
            buf.copy_(t)
            moved.append(buf)
        (x0, k0, e0d) = moved
        torch.cuda.synchronize()
    placement_note = None
    try:
        bufs = [sysd.alloc_outputs(n_local, mode, packed_flags=packed, placement=placement, pitch=pitch)
                for _ in range(n_out_bufs)]
    except RuntimeError as exc:
        if placement != "arena":
            raise
        # no placement control on this device / driver: the run still measures the march, on arrays from the
        # torch allocator, and says so
        placement_note = "arena unavailable (%s): path arrays from the torch allocator" % exc
        print("bench.py: " + placement_note, file=sys.stderr)
        placement = "torch"
        bufs = [sysd.alloc_outputs(n_local, mode, packed_flags=packed, placement=placement, pitch=pitch)
                for _ in range(n_out_bufs)]
    arena_obj = placed.PlacedArena.for_device(local_rank) if placement == "arena" else None
    input_kind = arena_obj.kind_of(x0) if arena_obj is not None else None
    host_staged = (args.backend == "gloo")
    stats = [pdist.SpotStatistics(dev, n_rays=n_local) for _ in range(nbuf)] if do_stats else []
    gathers = ([pdist.ImagePlaneGather(n_total, dev, stage_on_host=host_staged)
                for _ in range(nbuf if do_step_gather else 1)] if (do_step_gather or do_final_gather) else [])
    comm_stream = torch.cuda.Stream(device=dev) if multi else None
    main_stream = torch.cuda.current_stream(dev)
    side_done = [None] * nbuf          # event: side-stream work of slot b has finished

    def image_rows(ob):
        """image-plane rows of a buffer set: (x, k, mask byte row) -- views, nothing is copied"""
        v = sysd.views(ob)
        return v.x_hit[-1], v.k_out[-1], (v.flags[-1] if packed else v.valid_out[-1])

    def step(i, with_gather=True):
        b = i % nbuf
        if side_done[b] is not None:
            main_stream.wait_event(side_done[b])      # slot b (and its path arrays) are free again
        ob = bufs[b % n_out_bufs]
        if fused_stats:
            stats[b].trace_and_start(sysds[i % len(sysds)], x0, k0, ob, e0d)
        else:
            sysds[i % len(sysds)].trace_into(x0, k0, ob, e0d)
        if do_stats or do_step_gather:
            ev = torch.cuda.Event()
            ev.record(main_stream)
            (xi, ki, vi) = image_rows(ob)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                if fused_stats:
                    stats[b].reduce()
                elif do_stats:
                    stats[b].start(xi, sysd.views(ob).valid_out[-1])
                if do_step_gather and with_gather:
                    gathers[b].start(xi, ki, vi)
                    gathers[b].wait()
                done = torch.cuda.Event()
                done.record(comm_stream)
                side_done[b] = done

    def final_gather(last_step):
        """the one-off image-plane all-gather of the last traced bundle (49 B/ray)"""
        (xi, ki, vi) = image_rows(bufs[(last_step % nbuf) % n_out_bufs])
        ev = torch.cuda.Event()
        ev.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            gathers[0].start(xi, ki, vi)
            gathers[0].wait()
        comm_stream.synchronize()

    def finish():
        """every step's work (trace + per-step exchange) has completed"""
        if comm_stream is not None:
            comm_stream.synchronize()
        torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    def timed_region(n_steps, **kw):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i, **kw)
        finish()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=(dev if args.backend == "nccl" else "cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # device wake-up (not one of the W warm-up steps): after idle the first ~25 ms of launches run
    # at ramping clocks (per-launch trace in DESIGN.md section 5); 30 plain launches of the same
    # kernel bring the chip to its steady state before anything is counted
    for _ in range(PREWARM_LAUNCHES):
        sysd.trace_into(x0, k0, bufs[0], e0d)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    finish()
    if do_final_gather and args.warmup > 0:
        final_gather(args.warmup - 1)          # warms the all-gather path too
    elapsed = timed_region(args.steps)
    # N > 1: the same steps without the image-plane all-gather, measured right after (reported beside)
    elapsed_without_gather = timed_region(args.steps, with_gather=False) if do_step_gather else None
    # the final image-plane gather of --exchange final-gather is not one of the K steps: timed on its own
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
        (n_in, n_out) = sysd.ray_counts(n_local)
        if not iso:
            # concatenated layout (real k): read x0, k0, E0; per surface write x_hit 24 B + mask 1 B per
            # entering ray and k_out 24 B + mask 1 B per leaving ray (crystal interfaces double the rays)
            alg = 72 * n_local + 25 * (sum(n_in) + sum(n_out)) if args.mode == "path" \
                else 72 * n_local + 25 * (n_in[-1] + n_out[-1])
        elif args.mode == "path":
            alg = algorithmic_bytes(n_local, S, record_bytes=record_bytes)
        else:
            alg = n_local * (72 + record_bytes)
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        tkey = "%s_%d%s" % (args.mode, n_local, "" if packed else "_two_masks")
        if args.config != "doublegauss":
            tkey = args.config + "_" + tkey
        tent = _lookup("hbm_traffic.json", tkey)
        hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": achieved / HBM_PEAK_GBS,
               "traffic": tent["bytes_per_launch"] if tent else None,
               "traffic_source": ("profiles/hbm_traffic.json[%s]: rocprofv3 PMC passes of an earlier run of this "
                                  "workload, looked up by ray count -- NOT measured in this run" % tkey)
               if tent else None,
               "kernel": "k_trace_iso" if iso else "k_trace_general", "kernel_ms": kernel_ms,
               "algorithmic_bytes_per_launch": alg,
               "bytes_per_ray_surface_op": alg / (n_local * S)}
        if args.config == "doublegauss":
            hbm["frac_at_98B_per_op_convention"] = (n_local * S * 98 / (kernel_ms * 1e-3) / 1e9) / HBM_PEAK_GBS
        roofline = hbm
        if not iso:
            # crystal march: FP64-VALU bound -- flops per launch from SQ instruction
            # counters (benchmarks/valu_profile.py pass c), HBM as the secondary roof
            fent = _lookup("fp64_flops.json", "%s_%s_%d" % (args.config, args.mode, n_local))
            if fent:
                tf = fent["flops_per_launch"] / (kernel_ms * 1e-3) / 1e12
                roofline = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS, "traffic": hbm["traffic"],
                            "flops_per_launch": fent["flops_per_launch"],
                            "flops_source": "profiles/fp64_flops.json: SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 of an "
                                            "earlier PMC run of this workload (2 flop per FMA, 64 lanes per "
                                            "wave instruction), looked up -- NOT measured in this run",
                            "kernel": "k_trace_general", "kernel_ms": kernel_ms, "secondary": hbm,
                            # all VALU wave instructions (selects, compares, address arithmetic included) at one
                            # per 4 cycles and SIMD against the 1024 SIMDs at 2.4 GHz: what the kernel is bound by
                            "valu_issue_frac": (fent["counters"]["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9)
                                                / (kernel_ms * 1e-3)) if "counters" in fent else None}
            else:
                roofline = dict(hbm, note="FP64-VALU bound kernel; no flop count on file for this size "
                                          "(profiles/fp64_flops.json), HBM fraction shown")
        arena_stats = arena_obj.stats() if arena_obj is not None else None
        kinds_out = bufs[0]["placement"].get("kinds")
        if placement_note is None and kinds_out and len(set(kinds_out[:2])) < 2:
            placement_note = ("x_hit and k_out share a kind of HBM (the arena found no second kind within its "
                              "hunt): expect the 5.6 TB/s regime of same-kind write streams")
        elif placement_note is None and arena_obj is not None and iso and input_kind is not None \
                and kinds_out and input_kind in kinds_out[:2]:
            placement_note = "the inputs share a kind of HBM with a path array (no third kind found): about 5 % slower"
        out = {
            "metric": "ray_surface_ops_per_s", "value": value, "unit": "ray-surface-ops/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "rays_per_gpu": n_local, "rays_total": n_total, "surfaces": S,
                       "rays_per_surface": None if iso else {"entering": n_in, "leaving": n_out},
                       "mode": args.mode, "record_bytes": record_bytes if iso else 25,
                       "masks": ("valid | valid_out << 1 in one byte" if packed else "two byte arrays"),
                       "sharding": "rays" if n_gpus > 1 else "none",
                       "wavelengths": len(sysds), "prewarm_launches": PREWARM_LAUNCHES,
                       "output_placement": {"policy": bufs[0]["placement"]["policy"], "note": placement_note,
                                            "memory_kinds_of_x_hit_and_k_out": bufs[0]["placement"].get("kinds"),
                                            "memory_kind_of_inputs": input_kind,
                                            "inputs": "arena" if input_kind is not None else "torch allocator",
                                            "arena": arena_stats},
                       "image_plane_exchange": {
                           "per_step": {"gather": "spot moments from the trace kernel + one 7-double all-reduce, then "
                                                  "image-plane all-gather 49 B/ray (7 row collectives straight into "
                                                  "the [row][global ray] layout), side stream, overlaps the next trace",
                                        "stats": "spot moments from the trace kernel + one 7-double all-reduce (side stream)",
                                        "final-gather": "spot moments + one 7-double all-reduce (side stream)",
                                        "none": "none"}[exchange]
                                       + ("" if fused_stats or not do_stats else " [two-pass statistics]"),
                           "final": ("image-plane all-gather 49 B/ray, once after the K timed steps"
                                     if do_final_gather else "none"),
                           "final_gather_ms": final_gather_ms,
                           "ms_per_step_without_gather": (elapsed_without_gather / args.steps * 1e3
                                                          if elapsed_without_gather else None),
                           "value_without_gather": (ops_total / elapsed_without_gather
                                                    if elapsed_without_gather else None),
                           "backend": ("rccl" if args.backend == "nccl" else "gloo dry run (host staged)")
                           if multi else "none"},
                       "image_plane_spot": spot,
                       "build": prt_build.build_info(_lib.LIB_PATH)},
            "roofline": roofline,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            if args.config == "doublegauss":
                m = min(n_local, 4_000_000)
                out["cpu_baseline"] = cpu_baseline(records, x0[:, :m].cpu().numpy(), k0[:, :m].cpu().numpy(),
                                                   e0d[:, :m].cpu().numpy(), n_all=n_local)
            else:
                m = 4_000_000 if args.config == "asphere" else 64_000
                out["cpu_baseline"] = cpu_baseline_numpy(records, x0[:, :m].cpu().numpy(), k0[:, :m].cpu().numpy(),
                                                         e0d[:, :m].cpu().numpy(), m, n_local)
        else:
            out["cpu_baseline"] = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio
        # (block-buffered on a pipe, so it would otherwise surface at exit, after the line)
        _flush_c_stdio()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        sys.stdout.flush()


if __name__ == "__main__":
    main()
