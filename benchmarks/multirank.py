"""The N > 1 program of bench.py (BASELINE configs[4]; also ``--force-multi`` with one rank): ONE bundle sharded by
rays over the GPUs, one process per GPU, five wavelengths cycled, per-step image-plane exchange on a side stream.
And the watchdog that turns a hang into a JSON line with an ``error`` field.
"""
import json
import os
import sys
import threading
import time

import torch
import torch.distributed as dist

from .verify import verify_outputs
from .workloads import (HBM_PEAK_GBS, PREWARM_LAUNCHES, SHORT_WORKLOAD, STRONG_SCALING_RAYS, algorithmic_bytes,
                        make_workload)


class Watchdog(object):
    def __init__(self, seconds, rank, json_fd_ref, base):
        self.seconds = seconds
        self.stage = "start"
        self._done = threading.Event()
        if seconds > 0:
            t = threading.Thread(target=self._run, args=(rank, json_fd_ref, base), daemon=True)
            t.start()

    def _run(self, rank, json_fd_ref, base):
        if self._done.wait(self.seconds):
            return
        msg = "watchdog: no result after %g s (stage: %s)" % (self.seconds, self.stage)
        try:
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:
            pass
        if rank == 0:
            line = dict(base, value=None, ms_per_step=None, error=msg)
            os.write(json_fd_ref[0], (json.dumps(line) + "\n").encode())
        os._exit(3)

    def done(self):
        self._done.set()


def run_multi(args, dev, world, rank, local_rank, watchdog):
    """-> (compact line, detail) on rank 0, (None, None) elsewhere"""
    from pyrate_amd import build as prt_build, engine, placed, _lib
    from pyrate_amd import distributed as pdist
    n_gpus = world
    strong = args.scaling == "strong"
    if strong and args.rays is not None:
        raise SystemExit("--scaling strong (the default for N > 1) takes --rays-total (rays of the whole bundle); "
                         "--rays (per GPU) goes with --scaling weak")
    rays = args.rays if args.rays is not None else 12_500_000
    total_rays = (args.rays_total if args.rays_total is not None else STRONG_SCALING_RAYS) if strong else None
    watchdog.stage = "bundle generation"
    # shards of one common stride that is a multiple of 512 rays: rank r's slot of a gathered row starts on a 4-KiB
    # boundary, so the march can write its image plane straight into it (ImagePlaneGather.own_rows)
    align = 512
    wl = make_workload("doublegauss", rays, dev, n_gpus=n_gpus, rank=rank, multi=True,
                       first_segment=args.first_segment, align=align, total_rays=total_rays)
    (x0, k0, e0d, uni) = (wl["x0"], wl["k0"], wl["e0"], wl["uniform"])
    (n_total, n_local, S) = (wl["n_total"], wl["n_local"], wl["S"])
    sysds = [engine.DeviceSystem(r, local_rank) for r in wl["record_sets"]]
    sysd = sysds[0]
    mode = _lib.MODE_PATH if args.mode == "path" else _lib.MODE_IMAGE
    rccl = args.backend == "nccl"
    exchange = args.exchange
    if exchange == "auto":
        # the choice between the two forms of the image-plane gather is MEASURED at start-up (below) where both can run:
        # RCCL, more than one rank, path mode, fused statistics
        # (PRT_BENCH_PROBE_DRY=1: also under gloo -- the dry run of this very code with several ranks on ONE GPU)
        dry = os.environ.get("PRT_BENCH_PROBE_DRY", "0") == "1"
        can_probe = (rccl or dry) and world > 1 and args.mode == "path" and not args.two_pass_stats \
            and args.gather_mode == "inplace"
        exchange = "probe" if can_probe else "gather"
    if exchange == "gather-direct" and (not rccl or args.mode != "path" or args.two_pass_stats):
        raise SystemExit("--exchange gather-direct: RCCL backend, path mode, fused statistics")
    do_stats = exchange in ("probe", "gather", "gather-direct", "stats", "final-gather")
    do_step_gather = exchange in ("probe", "gather", "gather-direct")
    do_final_gather = exchange == "final-gather"
    fused_stats = do_stats and not args.two_pass_stats
    # side-stream jobs in flight: the job of step i overlaps the trace of step i+1.  The fused statistics only touch
    # 7-double vectors; the gather reads the image-plane rows of the path arrays, so with it the path arrays are
    # double-buffered.
    nbuf = 2 if (do_stats or do_step_gather) else 1
    n_out_bufs = 2 if (do_step_gather or (do_stats and not fused_stats)) else 1
    packed = not args.two_mask_arrays
    record_bytes = 49 if packed else 50
    placement = args.placement if mode == _lib.MODE_PATH else "torch"
    # one row pitch on every rank: a gathered row is read n_pad elements deep (pdist.ImagePlaneGather)
    pitch = engine.recommended_pitch(pdist.shard_stride(n_total, n_gpus, align))
    watchdog.stage = "output allocation (arena)"
    from .configs import alloc_outputs_or_torch
    (bufs, placement, placement_note) = alloc_outputs_or_torch(sysd, n_local, mode, packed, placement, pitch,
                                                                count=n_out_bufs)
    arena_obj = placed.PlacedArena.for_device(local_rank) if placement == "arena" else None
    input_kind = arena_obj.kind_of(x0) if arena_obj is not None else None
    host_staged = not rccl
    stats = [pdist.SpotStatistics(dev, n_rays=n_local) for _ in range(nbuf)] if do_stats else []

    def build_exchange(kind):
        """the receive buffers of one form of the per-step gather"""
        direct = kind == "gather-direct"
        if direct:
            gathers = [pdist.DirectImagePlaneGather(n_total, dev, align=align) for _ in range(nbuf)]
        elif kind in ("gather", "final-gather"):
            gathers = [pdist.ImagePlaneGather(n_total, dev, stage_on_host=host_staged, align=align)
                       for _ in range(nbuf if kind == "gather" else 1)]
        else:
            gathers = []
        # in place: the march of slot b deposits its image plane in gathers[b]'s receive buffer (prt_trace_ex redirect)
        # (the direct form always: its receive buffers are device buffers whatever the backend)
        inplace = (direct or (kind == "gather" and args.gather_mode == "inplace" and not host_staged)) \
            and mode == _lib.MODE_PATH
        return {"kind": kind, "direct": direct, "gathers": gathers, "inplace": inplace}

    ex = {}

    def use_exchange(e):
        ex.clear()
        ex.update(e)
        for b in range(nbuf):
            ob = dict(bufs[b % n_out_bufs])
            ob.pop("image_rows", None)
            if e["inplace"]:
                ob["image_rows"] = e["gathers"][b].own_rows()
            bufs[b % n_out_bufs] = ob

    comm_stream = torch.cuda.Stream(device=dev)
    if args.trace_stream != "default":
        # the march on a stream of its own (a hardware queue that RCCL's stream and the side stream do not share --
        # their barrier packets otherwise sit between two marches in the same queue: +4 % per step)
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1 if args.trace_stream == "high" else 0))
    main_stream = torch.cuda.current_stream(dev)
    side_done = [None] * nbuf          # event: side-stream work of slot b has finished
    traced = [torch.cuda.Event() for _ in range(nbuf)]          # events are made once and re-recorded every step
    side_events = [torch.cuda.Event() for _ in range(nbuf)]

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
            stats[b].trace_and_start(sysds[i % len(sysds)], x0, k0, ob, e0d, uniform=uni)
        else:
            sysds[i % len(sysds)].trace_into(x0, k0, ob, e0d, uniform=uni)
        if do_stats or do_step_gather:
            ev = traced[b]
            ev.record(main_stream)
            (gathers, direct, inplace) = (ex["gathers"], ex["direct"], ex["inplace"])
            gather_now = do_step_gather and with_gather
            if (gather_now and not inplace) or (do_stats and not fused_stats):
                (xi, ki, vi) = image_rows(ob)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev)
                if direct and gather_now:
                    gathers[b].start_in_place()      # peer writes; the all-reduce below is their fence
                if fused_stats:
                    stats[b].reduce()
                elif do_stats:
                    stats[b].start(xi, sysd.views(ob).valid_out[-1])
                if gather_now and not direct:
                    if inplace:
                        gathers[b].start_in_place()
                    else:
                        gathers[b].start(xi, ki, vi)
                    gathers[b].wait()
                done = side_events[b]
                done.record(comm_stream)
                side_done[b] = done

    def final_gather(last_step):
        """the one-off image-plane all-gather of the last traced bundle (49 B/ray)"""
        (xi, ki, vi) = image_rows(bufs[(last_step % nbuf) % n_out_bufs])
        ev = torch.cuda.Event()
        ev.record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            ex["gathers"][0].start(xi, ki, vi)
            ex["gathers"][0].wait()
        comm_stream.synchronize()

    def finish():
        """every step's work (trace + per-step exchange) has completed"""
        comm_stream.synchronize()
        torch.cuda.synchronize()

    issue_s = [0.0]
    reduce_dev = dev if rccl else "cpu"

    def timed_region(n_steps, **kw):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(i, **kw)
        issue_s[0] = time.perf_counter() - t0          # the host is done issuing; the device may still be busy
        finish()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    watchdog.stage = "warm-up"
    for _ in range(PREWARM_LAUNCHES):
        sysd.trace_into(x0, k0, bufs[0], e0d, uniform=uni)
    torch.cuda.synchronize()
    probe = None
    if exchange == "probe":
        # Which form of the per-step gather is faster on THIS node is measured, not assumed: three steps of each behind
        # one warm-up step, max over ranks.  The direct form (IPC-mapped peer buffers) may be unavailable -- then every
        # rank says so before anybody uses it, and all ranks take the collective.
        watchdog.stage = "exchange probe"
        probe = {}
        built = {}
        for kind in ("gather", "gather-direct"):
            ok = True
            try:
                built[kind] = build_exchange(kind)
            except Exception as exc:          # (IPC not available, out of memory ...)
                print("bench.py: exchange '%s' unavailable on rank %d: %s" % (kind, rank, str(exc)[:300]), file=sys.stderr)
                ok = False
            if not all_ranks(ok):
                built.pop(kind, None)
                probe[kind] = None
                continue
            use_exchange(built[kind])
            for b in range(nbuf):
                side_done[b] = None
            step(0)
            finish()
            probe[kind] = timed_region(3) / 3 * 1e3
        exchange = min((k for k in probe if probe[k] is not None), key=lambda k: probe[k])
        for kind in list(built):
            if kind != exchange:
                del built[kind]
        for b in range(nbuf):
            side_done[b] = None
        use_exchange(built[exchange])
        torch.cuda.empty_cache()
    else:
        use_exchange(build_exchange(exchange))
    for i in range(args.warmup):
        step(i)
    finish()
    if do_final_gather and args.warmup > 0:
        final_gather(args.warmup - 1)          # warms the all-gather path too
    watchdog.stage = "timed region"
    elapsed = timed_region(args.steps)
    host_issue_ms = issue_s[0] / args.steps * 1e3
    # the same steps without the image-plane all-gather, measured right after (reported beside)
    watchdog.stage = "timed region without gather"
    elapsed_without_gather = timed_region(args.steps, with_gather=False) if do_step_gather else None
    # the final image-plane gather of --exchange final-gather is not one of the K steps: timed on its own
    final_gather_ms = None
    if do_final_gather:
        dist.barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        final_gather(args.steps - 1)
        dist.barrier()
        final_gather_ms = (time.perf_counter() - tg) * 1e3
    spot = None
    if do_stats:
        (cnt, cen, rms) = stats[(args.steps - 1) % nbuf].result()
        spot = {"rays": float(cnt), "centroid_mm": [float(c) for c in cen], "rms_spot_mm": rms}
    watchdog.stage = "kernel timing"
    kernel_ms = sysd.trace_timed(x0, k0, bufs[0], max(args.steps, 5), e0d, uniform=uni)
    torch.cuda.synchronize()
    # every rank checks what that launch wrote for its shard (all rays on their surfaces, |k| = n); rank 0 reports
    # its own figures and whether ALL ranks passed
    watchdog.stage = "verification"
    verified = verify_outputs(dict(wl, records=wl["record_sets"][0]), sysd, bufs[0],
                              with_oracle=(rank == 0 and not args.no_cpu_baseline))
    oks = [None] * world
    dist.all_gather_object(oks, bool(verified["ok"]))
    verified["all_ranks_ok"] = all(oks)
    verified["ok_per_rank"] = oks
    ops_total = n_total * S * args.steps
    if rank != 0:
        return None, None
    alg = algorithmic_bytes(wl, sysd, args.mode, record_bytes)
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "k_trace_iso", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg,
                "bytes_per_ray_surface_op": alg / (n_local * S),
                "note": "per rank (rank 0's march); every rank runs the same launch on its shard"}
    kinds_out = bufs[0]["placement"].get("kinds")
    # what DESIGN.md section 6 expects for this run (so that a measured curve can be read against it): the trace of
    # one shard takes what the 1-GPU march takes at this size (kernel_ms); the per-step all-gather moves
    # 49 B x n_pad to each of the N-1 peers over one xGMI link each (point-to-point, ~153 GB/s per link and
    # direction, 0.65-0.8 of it reached) and overlaps the next trace -> a step costs max(trace, gather)
    n_pad = pdist.shard_stride(n_total, n_gpus, align)
    gather_ms = [49.0 * n_pad / (f * 153e9) * 1e3 for f in (0.8, 0.65)] if n_gpus > 1 else [0.0, 0.0]
    expected = {"trace_ms_per_step": kernel_ms,
                "gather_ms_per_step_at_0.8_and_0.65_of_the_link_rate": gather_ms,
                "ms_per_step_with_gather": [max(kernel_ms, g) for g in gather_ms],
                "value_with_gather": [n_total * S / (max(kernel_ms, g) * 1e-3) for g in gather_ms],
                "value_without_gather": n_total * S / (kernel_ms * 1e-3),
                "basis": "DESIGN.md section 6: step = max(this rank's march, image-plane all-gather of 49 B x %d rays "
                         "per peer over one xGMI link each at 0.65-0.8 x 153 GB/s)" % n_pad}
    ms_total = elapsed / args.steps * 1e3
    ms_no_gather = elapsed_without_gather / args.steps * 1e3 if elapsed_without_gather else None
    backend = "rccl" if rccl else "gloo dry run (host staged)"
    per_step = {"gather": "spot moments from the trace kernel + one 7-double all-reduce, then image-plane all-gather "
                          "49 B/ray (7 row collectives straight into the [row][global ray] layout%s), side stream, "
                          "overlaps the next trace" % (", IN PLACE: the march wrote the shard's rows into its slot of the "
                                                       "receive buffer" if ex.get("inplace") else ""),
                "gather-direct": "spot moments from the trace kernel; the march writes the shard's image plane into its "
                                 "slot of the rank's receive buffer, then one device-to-device copy per row and peer "
                                 "into the peers' IPC-mapped receive buffers, closed by the 7-double all-reduce; side "
                                 "stream, overlaps the next trace",
                "stats": "spot moments from the trace kernel + one 7-double all-reduce (side stream)",
                "final-gather": "spot moments + one 7-double all-reduce (side stream)",
                "none": "none"}[exchange] + ("" if fused_stats or not do_stats else " [two-pass statistics]")
    base = {"metric": "ray_surface_ops_per_s", "value": ops_total / elapsed, "unit": "ray-surface-ops/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_total,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic"}
    build = prt_build.build_info(_lib.LIB_PATH)
    compact_cfg = {"workload": SHORT_WORKLOAD["doublegauss"].replace("configs[1]", "configs[4]") + (
                       ", ONE bundle split over the GPUs" if strong else ", weak scaling"),
                   "rays_per_gpu": n_local, "rays_total": n_total, "surfaces": S, "mode": args.mode,
                   "first_segment": "uniform" if uni is not None else "arrays", "record_bytes": record_bytes,
                   "sharding": "rays", "wavelengths": len(sysds), "exchange": exchange, "exchange_probe_ms": probe,
                   "backend": backend, "rccl_world": world if rccl else None,
                   "row_batching": pdist.gather_batch_mode(), "in_place": bool(ex.get("inplace")),
                   "ms_trace": kernel_ms, "ms_total": ms_total,
                   "ms_gather": (max(ms_total - ms_no_gather, 0.0) if ms_no_gather is not None else None),
                   "ms_per_step_without_gather": ms_no_gather,
                   "value_without_gather": (ops_total / elapsed_without_gather if elapsed_without_gather else None),
                   "final_gather_ms": final_gather_ms, "host_issue_ms_per_step": host_issue_ms,
                   "placement": {"policy": bufs[0]["placement"]["policy"], "kinds": kinds_out, "input_kind": input_kind,
                                 "note": placement_note},
                   "expected": {"ms_per_step_with_gather": expected["ms_per_step_with_gather"],
                                "value_with_gather": expected["value_with_gather"],
                                "value_without_gather": expected["value_without_gather"]},
                   "image_plane_spot": spot, "build": build.get("libprt_sha256_16"),
                   "detail": "bench_detail.json"}
    short_verified = {k: verified.get(k) for k in ("ok", "all_ranks_ok", "ok_per_rank", "tolerance", "max_resid",
                                                   "max_rel_x", "max_abs_k", "n_checked")}
    short_verified["oracle_sample"] = verified.get("oracle_sample")
    line = dict(base, config=compact_cfg,
                roofline={k: roofline[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                                   "kernel_ms", "algorithmic_bytes_per_launch")},
                cpu_baseline=None, verified=short_verified,
                # the same shape as the N = 1 line's scaling_point (this bundle on ONE GPU)
                scaling_point={"rays": n_total, "ms": ms_total, "frac": roofline["frac"], "ok": bool(verified["ok"] and
                                                                                                     verified["all_ranks_ok"]),
                               "value": ops_total / elapsed})
    detail = dict(base, config=dict(compact_cfg, workload=wl["workload"], build=build, expected=expected,
                                    image_plane_exchange={"per_step": per_step, "final": (
                                        "image-plane all-gather 49 B/ray, once after the K timed steps"
                                        if do_final_gather else "none")},
                                    trace_stream=args.trace_stream, masks=("valid | valid_out << 1 in one byte" if packed
                                                                           else "two byte arrays"),
                                    arena=arena_obj.stats() if arena_obj is not None else None),
                  roofline=roofline, cpu_baseline=None, verified=verified)
    return line, detail
