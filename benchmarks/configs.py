"""The single-GPU measurements of bench.py: one record per configuration (K timed steps + the dominant kernel's
HIP-event time + roofline + verification + CPU baseline), the plugin-granular sweep, image mode with fused moments,
the end-to-end drop-in call, and the run that strings them together (``run_single_gpu``).
"""
import os
import shutil
import sys
import time

import numpy as np
import torch

from . import pmc
from .cpu_baseline import cpu_baseline
from .verify import verify_outputs
from .workloads import (FP64_VALU_PEAK_TFLOPS, HBM_PEAK_GBS, PREWARM_LAUNCHES, PREWARM_MS, ROOT, SECONDARY_MARCH_CONFIGS,
                        SHORT_WORKLOAD, SINGLE_GPU_CONFIGS, STRONG_SCALING_RAYS, VERIFY_TOL, algorithmic_bytes,
                        input_bytes_per_ray, kernel_label, make_workload)


def _lookup(fname, key):
    """a per-launch figure measured by rocprofv3 PMC passes of an EARLIER run of the same workload -- looked up,
    not measured in this run"""
    import json
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def alloc_outputs_or_torch(sysd, n_local, mode, packed, placement, pitch, count=1):
    """the path arrays of a run: from the arena (the product path's allocation for arrays of this size) or,
    where the device / driver offers no placement control, from the torch allocator -- and says so"""
    note = None
    try:
        bufs = [sysd.alloc_outputs(n_local, mode, packed_flags=packed, placement=placement, pitch=pitch)
                for _ in range(count)]
    except RuntimeError as exc:
        if placement != "arena":
            raise
        note = "arena unavailable (%s): path arrays from the torch allocator" % exc
        print("bench.py: " + note, file=sys.stderr)
        placement = "torch"
        bufs = [sysd.alloc_outputs(n_local, mode, packed_flags=packed, placement=placement, pitch=pitch)
                for _ in range(count)]
    return bufs, placement, note


def measure_single(config, args, dev, rays, with_cpu, verify_oracle=None):
    from pyrate_amd import engine, placed, _lib
    wl = make_workload(config, rays, dev, first_segment=args.first_segment)
    sysd = engine.DeviceSystem(wl["records"], dev.index)
    iso = sysd.all_isotropic
    mode = _lib.MODE_PATH if args.mode == "path" else _lib.MODE_IMAGE
    packed = iso and not args.two_mask_arrays
    record_bytes = 49 if packed else 50
    placement = args.placement if mode == _lib.MODE_PATH else "torch"
    (x0, k0, e0, uni, n_local) = (wl["x0"], wl["k0"], wl["e0"], wl["uniform"], wl["n_local"])
    if args.inputs == "torch" and iso:
        # A/B: the inputs in torch-allocated arrays instead of arena memory of a third kind
        moved = []
        for t in (x0, k0, e0):
            if t is None:
                moved.append(None)
                continue
            buf = torch.empty((3, t.stride(0)), dtype=torch.float64, device=dev)[:, :n_local]
            buf.copy_(t)
            moved.append(buf)
        (x0, k0, e0) = moved
        torch.cuda.synchronize()
    pitch = engine.recommended_pitch(n_local) if iso else None
    (bufs, placement, placement_note) = alloc_outputs_or_torch(sysd, n_local, mode, packed, placement, pitch)
    ob = bufs[0]
    arena_obj = placed.PlacedArena.for_device(dev.index) if placement == "arena" else None
    input_kind = arena_obj.kind_of(x0) if arena_obj is not None else None

    launch = sysd.launcher(x0, k0, ob, e0, uniform=uni)     # the argument struct is built once

    # device wake-up (not one of the W warm-up steps): after idle the first ~25 ms of launches run at ramping
    # clocks; 30 plain launches of the same kernel -- and, for kernels as short as the crystal march (0.12 ms), as many
    # more as it takes to fill 50 ms -- bring the chip to its steady state before anything is counted
    for _ in range(PREWARM_LAUNCHES):
        launch()
    torch.cuda.synchronize()
    est_ms = sysd.trace_timed(x0, k0, ob, 10, e0, uniform=uni)
    prewarm = PREWARM_LAUNCHES + 10
    while prewarm * est_ms < PREWARM_MS:
        launch()
        prewarm += 1
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        launch()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # dominant kernel: average launch duration from HIP events on the launch stream
    kernel_ms = sysd.trace_timed(x0, k0, ob, max(args.steps, 5), e0, uniform=uni)
    torch.cuda.synchronize()

    S = wl["S"]
    alg = algorithmic_bytes(wl, sysd, args.mode, record_bytes)
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": None, "traffic_source": None, "kernel": kernel_label(config), "kernel_ms": kernel_ms,
           "algorithmic_bytes_per_launch": alg, "bytes_per_ray_surface_op": alg / (n_local * S)}
    if config == "doublegauss" and args.mode == "path":
        # SURVEY 8d's convention charges every surface a re-read of the state (98 B per op); the fused march does not do
        # that, so this is a RATIO against that convention's roof (> 1 by construction), not a fraction of anything
        hbm["ops_vs_98B_convention"] = (n_local * S * 98 / (kernel_ms * 1e-3) / 1e9) / HBM_PEAK_GBS
    kinds_out = ob["placement"].get("kinds")
    if placement_note is None and kinds_out and len(set(kinds_out[:2])) < 2:
        placement_note = ("x_hit and k_out share a kind of HBM (the arena found no second kind within its "
                          "hunt): expect the 5.6 TB/s regime of same-kind write streams")
    elif placement_note is None and arena_obj is not None and iso and input_kind is not None \
            and kinds_out and input_kind in kinds_out[:2]:
        placement_note = "the inputs share a kind of HBM with a path array (no third kind found): about 5 % slower"
    (n_in, n_out) = sysd.ray_counts(n_local)
    rec = {"name": config, "workload": wl["workload"], "value": n_local * S * args.steps / elapsed,
           "unit": "ray-surface-ops/s", "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "dtype": "f64", "prewarm_launches": prewarm,
           "rays": n_local, "surfaces": S, "mode": args.mode,
           "rays_per_surface": None if iso else {"entering": n_in, "leaving": n_out},
           "first_segment": ("uniform k0 / E0 (collimated bundle: one vector each, 24 B/ray of loads)"
                             if uni is not None else "arrays x0, k0, E0 (72 B/ray of loads)"),
           "record_bytes": record_bytes if iso else 25,
           "masks": ("valid | valid_out << 1 in one byte" if packed else "two byte arrays"),
           "output_placement": {"policy": ob["placement"]["policy"], "note": placement_note,
                                "memory_kinds_of_x_hit_and_k_out": kinds_out,
                                "memory_kind_of_inputs": input_kind,
                                "inputs": "arena" if input_kind is not None else "torch allocator"},
           "roofline": hbm, "cpu_baseline": None, "_iso": iso, "_alg": alg, "_n_local": n_local}
    # what the timed launches wrote, checked (outside every timed region; the oracle leg runs with the CPU baseline)
    rec["verified"] = verify_outputs(wl, sysd, ob, with_oracle=with_cpu if verify_oracle is None else verify_oracle)
    if config == "benchmark":
        # context, not a published number (vs_baseline stays null): what the reference itself reaches on this workload
        rec["reference_rate"] = {"value": 7.8e4, "unit": "ray-surface-ops/s",
                                 "where": "BASELINE.md section 2: the reference's demo_benchmark.py system verbatim in the "
                                          "survey container (8 vCPU, 99 693 rays, 8.94 s)",
                                 "ratio": rec["value"] / 7.8e4}
    if with_cpu:
        rec["cpu_baseline"] = cpu_baseline(wl, budget_s=args.cpu_budget, with_numpy=(config == (args.config or "doublegauss")))
    del bufs, ob, x0, k0, e0
    return rec


def _event_timed(fn, steps, warmup):
    """average milliseconds of fn() over `steps` calls, HIP events on the current stream (the stream fn launches on)"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    (a, b) = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps, (time.perf_counter() - t0) / steps * 1e3


def plugin_sweep(dev, rays, placement="auto", tap=None, fused=False):
    """(sweep, ctx): ``sweep()`` drives the double Gauss surface by surface through DeviceSystem.propagate /
    DeviceSystem.interact (prt_propagate_rows + prt_interact_rows), the loop of optical_element.py:336-375 -- or, ``fused``,
    through DeviceSystem.surface_step (prt_surface_step_rows: both calls of a surface in one launch); ``tap(s, x_hit,
    valid_hit, k_out, valid_out)`` sees every surface's arrays.  ``ctx['last']`` = the last surface's arrays."""
    from pyrate_amd import engine
    wl = make_workload("doublegauss", rays, dev, first_segment="arrays")
    sysd = engine.DeviceSystem(wl["records"], dev.index)
    (x0, k0, e0, S) = (wl["x0"], wl["k0"], wl["e0"], wl["S"])
    ctx = {"wl": wl, "sysd": sysd, "last": {}, "tap": tap}

    def sweep():
        (x, k, valid) = (x0, k0, None)
        see = ctx["tap"]
        for s in range(S):
            if fused:
                (xh, k, v, valid) = sysd.surface_step(s, x, k, e_re=e0 if s == 0 else None, default_e=False,
                                                      valid_in=valid, placement=placement)
            else:
                if s == 0:
                    (xh, v) = sysd.propagate(0, x, k, e_re=e0, valid_in=None, placement=placement)
                else:
                    (xh, v) = sysd.propagate(s, x, k, default_e=False, valid_in=valid, placement=placement)
                (k, _, valid, _, _) = sysd.interact(s, xh, k, valid_in=v, placement=placement)
            x = xh
            if see is not None:
                see(s, xh, v, k, valid)
        ctx["last"] = dict(x=x, k=k, valid=valid, hit=v)
    return sweep, ctx


def measure_plugin(args, dev, rays, with_oracle=True, m=10_000, fused=False):
    """The double Gauss through the PLUGIN-GRANULAR calls: per surface one propagate (Material.propagate ->
    Surface.intersect) and one interact (Material.refract), the loop of optical_element.py:336-375 -- what a
    caller gets who drives the trace surface by surface.  Roof: SURVEY 8d's 98 B per ray-surface-op (one read and one
    write of the 49-B state per surface); two calls per surface cannot move less than 148 B (each of them reads the state
    it works on).  Verified: every surface's record of a 1e4-ray sample against the CPU oracle, and the last surface's
    record of ALL rays against the fused march (masks bit for bit).
    ``fused`` (config ``surface_step``): one launch per surface, DeviceSystem.surface_step = both calls of the surface
    (prt_surface_step_rows) -- 98 B per op, SURVEY 8d's figure itself."""
    from pyrate_amd import engine, _lib
    placement = "auto" if args.placement == "arena" else "torch"
    (sweep, ctx) = plugin_sweep(dev, rays, placement=placement, fused=fused)
    (wl, sysd) = (ctx["wl"], ctx["sysd"])
    (x0, k0, e0, n, S) = (wl["x0"], wl["k0"], wl["e0"], wl["n_local"], wl["S"])
    steps = max(5, min(args.steps, 20))
    (ms, wall_ms) = _event_timed(sweep, steps, 3)
    ops = n * S
    achieved = 98.0 * ops / (ms * 1e-3) / 1e9
    # one more sweep (untimed) whose per-surface arrays are sampled for the oracle
    sample = None
    if with_oracle:
        idx = np.unique(np.linspace(0, n - 1, min(m, n)).astype(np.int64))
        it = torch.from_numpy(idx).to(dev)
        taken = []
        ctx["tap"] = lambda s, xh, v, k, w: taken.append((xh[:, it].cpu().numpy(), v[it].cpu().numpy().astype(bool),
                                                          k[:, it].cpu().numpy(), w[it].cpu().numpy().astype(bool)))
        sweep()
        torch.cuda.synchronize()
        ctx["tap"] = None
        from oracle import seqtrace_c, seqtrace_np
        (o, kk, ee) = [np.ascontiguousarray(t[:, it].cpu().numpy()) for t in (x0, k0, e0)]
        use_c = seqtrace_c.supports(wl["records"])
        with np.errstate(all="ignore"):
            ref = seqtrace_c.trace(wl["records"], o, kk, ee) if use_c else seqtrace_np.trace(wl["records"], o, kk, ee)
        (rel_x, abs_k, mask_diff) = (0.0, 0.0, 0)
        for s in range(S):
            (gx, gv, gk, gw) = taken[s]
            rv = np.asarray(ref[s]["valid"], dtype=bool)
            rw = np.asarray(ref[s]["valid_out"], dtype=bool)
            mask_diff += int(np.count_nonzero(gv != rv)) + int(np.count_nonzero(gw != rw))
            if rv.any():
                dx = np.linalg.norm(gx[:, rv] - ref[s]["x_hit"][:, rv], axis=0)
                sc = np.maximum(np.linalg.norm(ref[s]["x_hit"][:, rv], axis=0), 1.0)
                rel_x = max(rel_x, float(np.nan_to_num(dx / sc, nan=np.inf).max()))
            if rw.any():
                abs_k = max(abs_k, float(np.nan_to_num(np.abs(gk[:, rw] - np.real(ref[s]["k_out"][:, rw])), nan=np.inf).max()))
        sample = {"rays": int(idx.size), "oracle": "oracle/seqtrace_c.c" if use_c else "oracle/seqtrace_np.py",
                  "max_rel_x": rel_x, "max_abs_k": abs_k, "mask_mismatches": mask_diff}
    # the fused march on the same bundle (image mode: the last surface's record)
    last = ctx["last"]
    ob = sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=False, placement="torch")
    sysd.trace_into(x0, k0, ob, e0)
    torch.cuda.synchronize()
    res = sysd.views(ob)
    mk = res.valid_out[0].bool()
    # (masks bit for bit; values to rounding: the per-surface calls and the fused march scale their intermediates
    #  differently)
    masks_equal = bool(torch.equal(last["valid"], res.valid_out[0]) and torch.equal(last["hit"], res.valid[0]))
    scale = res.x_hit[0][:, mk].norm(dim=0).clamp_min(1.0)
    rel_m = float(((last["x"][:, mk] - res.x_hit[0][:, mk]).abs().max(dim=0).values / scale).max()) if bool(mk.any()) else 0.0
    abs_m = float((last["k"][:, mk] - res.k_out[0][:, mk]).abs().max()) if bool(mk.any()) else 0.0
    ok = masks_equal and rel_m <= VERIFY_TOL and abs_m <= VERIFY_TOL
    if sample is not None:
        ok = ok and sample["max_rel_x"] <= VERIFY_TOL and sample["max_abs_k"] <= VERIFY_TOL and sample["mask_mismatches"] == 0
    from pyrate_amd import placed
    kinds = None
    if placement == "auto" and placed.DISABLED is None:
        arena = placed.PlacedArena.for_device(dev.index)
        kinds = [arena.kind_of(last[q]) for q in ("x", "k")]
    floor = 98.0 if fused else 148.0
    rec = {"name": "surface_step" if fused else "plugin",
           "workload": ("the double Gauss of configs[1] (%d rays x %d surfaces) surface by surface: " % (n, S)) + (
               "DeviceSystem.surface_step per surface (prt_surface_step_rows: Material.propagate + Material.refract of a surface "
               "in one launch, the loop body of optical_element.py:336-375)" if fused else
               "the plugin-granular calls, DeviceSystem.propagate + DeviceSystem.interact per surface (Material.propagate / "
               "Surface.intersect / Material.refract, optical_element.py:336-375)"),
           "value": ops / (wall_ms * 1e-3), "unit": "ray-surface-ops/s", "steps": steps, "ms_per_step": wall_ms,
           "rays": n, "surfaces": S, "mode": "per-surface calls", "dtype": "f64",
           "output_placement": {"policy": placement, "kinds_of_last_x_and_k": kinds},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "k_surface_step_rows" if fused else "k_propagate_rows + k_interact_iso_rows",
                        "kernel_ms": ms, "algorithmic_bytes_per_launch": 98.0 * ops, "bytes_per_ray_surface_op": 98.0,
                        "note": ("kernel_ms = device time of one sweep over the 12 surfaces (12 launches), HIP events" if fused
                                 else "kernel_ms = device time of one sweep over the 12 surfaces (24 launches), HIP events; "
                                      "floor of two calls per surface: 148 B per op (each call reads the 49-B state it works on)"),
                        "floor_bytes_per_ray_surface_op": floor,
                        "frac_at_floor_traffic": floor * ops / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "verified": {"ok": bool(ok), "what": "every surface's record of a sample against the CPU oracle; the last surface's "
                                                "record of all rays against the fused march's (masks bit for bit)",
                        "oracle_sample": sample, "max_rel_x": sample["max_rel_x"] if sample else None,
                        "max_abs_k": max(abs_m, sample["max_abs_k"]) if sample else abs_m,
                        "vs_fused_march": {"masks_equal": masks_equal, "max_rel_x": rel_m, "max_abs_k": abs_m},
                        "tolerance": VERIFY_TOL, "n_checked": n},
           "cpu_baseline": None, "_custom": True, "_alg": 98.0 * ops}
    del ob, res, last
    ctx.clear()
    return rec


def measure_image_moments(args, dev, rays):
    """The optimiser's call (optimize/optimize.py:73-91: trace, then a merit function of the image plane): ONE
    image-mode launch of the double Gauss that reduces the spot moments itself (prt_trace_moments) -- no path arrays,
    7 doubles out.  Bound by FP64 arithmetic, not HBM: the roofline is the FP64 vector peak (flops per launch: the
    path-mode march's, measured by this run's PMC pass -- the two modes do the same arithmetic)."""
    from pyrate_amd import engine, _lib
    wl = make_workload("doublegauss", rays, dev, first_segment=args.first_segment)
    sysd = engine.DeviceSystem(wl["records"], dev.index)
    (x0, k0, e0, uni, n, S) = (wl["x0"], wl["k0"], wl["e0"], wl["uniform"], wl["n_local"], wl["S"])
    ob = sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=True, placement="torch",
                            pitch=engine.recommended_pitch(n))
    ws = engine.MomentsWorkspace(dev, n_rays=n)

    def call():
        sysd.trace_moments_into(x0, k0, ob, ws, 0, e0, uniform=uni)
    for _ in range(PREWARM_LAUNCHES):
        call()
    (ms, wall_ms) = _event_timed(call, args.steps, args.warmup)
    mom = ws.out[0].cpu().numpy()
    (cnt, cen, rms) = engine.spot_from_moments(mom, sysd.moments_reference())
    # the same statistics from the image-plane arrays the launch wrote (torch, float64)
    res = sysd.views(ob)
    m = res.valid_out[0].bool()
    xs = res.x_hit[0][:, m]
    cen_ref = xs.mean(dim=1)
    rms_ref = float(torch.sqrt(((xs - cen_ref[:, None]) ** 2).sum() / (int(m.sum()) - 1)))
    dev_c = float((torch.tensor(cen, dtype=torch.float64, device=dev) - cen_ref).abs().max())
    ok = bool(int(cnt) == int(m.sum()) and dev_c <= 1e-10 and abs(rms - rms_ref) <= 1e-10 * max(1.0, rms_ref))
    rec = {"name": "image_moments", "workload": "the double Gauss of configs[1] (%d rays x %d surfaces), IMAGE mode with "
                                                "the spot moments reduced by the same launch (prt_trace_moments): trace "
                                                "+ merit function of an optimiser step, no path arrays" % (n, S),
           "value": n * S / (wall_ms * 1e-3), "unit": "ray-surface-ops/s", "steps": args.steps, "ms_per_step": wall_ms,
           "rays": n, "surfaces": S, "mode": "image + moments", "dtype": "f64",
           "roofline": {"bound": "fp64_valu", "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": None, "traffic": None, "kernel": "k_trace_iso<image, moments> + k_moments_stage/final",
                        "kernel_ms": ms, "hbm_frac": (24.0 + 49.0) * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "verified": {"ok": ok, "what": "count, centroid and RMS spot radius from the launch's 7 moments against the "
                                          "same statistics of the image-plane arrays it wrote", "count": int(cnt),
                        "max_abs_centroid_difference": dev_c, "rms_difference": abs(rms - rms_ref), "n_checked": n},
           "cpu_baseline": None, "_custom": True}
    del ob, res, ws
    return rec


def finish_roofline(rec, traffic, flops, lookup=True):
    """fill roofline.traffic (+ the FP64 roof of the crystal march) from the live PMC passes or the files"""
    if rec.get("_custom"):
        live = (traffic or {}).get(rec["name"])
        if live and live.get("bytes_per_launch"):
            r = rec["roofline"]
            r["traffic"] = live["bytes_per_launch"]
            r["traffic_ratio_to_algorithmic"] = live["bytes_per_launch"] / rec["_alg"] if rec.get("_alg") else None
            r["traffic_source"] = live["source"]
            r["frac_at_measured_traffic"] = live["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if rec["name"] == "image_moments":
            fl = (flops or {}).get("doublegauss")
            if fl and fl.get("flops_per_launch"):
                r = rec["roofline"]
                r["achieved"] = fl["flops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12
                r["frac"] = r["achieved"] / FP64_VALU_PEAK_TFLOPS
                r["flops_per_launch"] = fl["flops_per_launch"]
                r["flops_source"] = "the path-mode double Gauss march of this run (same arithmetic): " + fl["source"]
                valu = fl.get("valu_wave_instructions")
                r["valu_issue_frac"] = (valu * 4.0 / (1024 * 2.4e9) / (r["kernel_ms"] * 1e-3)) if valu else None
        for k in [k for k in rec if k.startswith("_")]:
            del rec[k]
        return rec
    hbm = rec["roofline"]
    live = (traffic or {}).get(rec["name"])
    if live and live.get("bytes_per_launch"):
        hbm["traffic"] = live["bytes_per_launch"]
        hbm["traffic_ratio_to_algorithmic"] = live["bytes_per_launch"] / rec["_alg"]
        hbm["traffic_source"] = live["source"]
    elif lookup:
        fkey = ("%s_%d_uniform" if rec["first_segment"].startswith("uniform") else "%s_%d") \
            % (rec["mode"], rec["_n_local"])
        if rec["name"] != "doublegauss":
            fkey = rec["name"] + "_" + fkey
        tent = _lookup("hbm_traffic.json", fkey)
        if tent:
            hbm["traffic"] = tent["bytes_per_launch"]
            hbm["traffic_source"] = ("profiles/hbm_traffic.json[%s]: rocprofv3 PMC passes of an earlier run of this "
                                     "workload, looked up by ray count -- NOT measured in this run%s"
                                     % (fkey, "" if not traffic else " (" + str(traffic.get("error")) + ")"))
    if not rec["_iso"]:
        # crystal march: FP64-VALU bound -- flops per launch from SQ instruction counters,
        # HBM as the secondary roof
        fl = (flops or {}).get(rec["name"])
        src = None
        if fl and fl.get("flops_per_launch"):
            (fpl, valu, src) = (fl["flops_per_launch"], fl.get("valu_wave_instructions"), fl["source"])
        else:
            fent = _lookup("fp64_flops.json", "%s_%s_%d" % (rec["name"], rec["mode"], rec["_n_local"]))
            (fpl, valu) = (None, None)
            if fent:
                fpl = fent["flops_per_launch"]
                valu = fent.get("counters", {}).get("SQ_INSTS_VALU")
                src = ("profiles/fp64_flops.json: SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 of an earlier PMC run of "
                       "this workload (2 flop per FMA, 64 lanes per wave instruction), looked up -- NOT measured "
                       "in this run")
        if fpl:
            ms = hbm["kernel_ms"]
            tf = fpl / (ms * 1e-3) / 1e12
            fp64 = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / FP64_VALU_PEAK_TFLOPS, "traffic": hbm["traffic"],
                    "flops_per_launch": fpl, "flops_source": src, "kernel": hbm["kernel"], "kernel_ms": ms,
                    # all VALU wave instructions (selects, compares, address arithmetic included) at one
                    # per 4 cycles and SIMD against the 1024 SIMDs at 2.4 GHz
                    "valu_issue_frac": (valu * 4.0 / (1024 * 2.4e9) / (ms * 1e-3)) if valu else None}
            # the roof the launch is closer to is its bound (round 4: without the eigenvectors the path-mode march is
            # a write-bound kernel like the isotropic one; image mode stays on the VALU side)
            if hbm["frac"] >= max(fp64["frac"], fp64["valu_issue_frac"] or 0.0):
                rec["roofline"] = dict(hbm, secondary=fp64)
            else:
                rec["roofline"] = dict(fp64, secondary=hbm)
        else:
            hbm["note"] = "FP64-VALU bound kernel; no flop count available, HBM fraction shown"
    else:
        # isotropic marches are HBM bound; the FP64 / VALU-issue side is carried as the secondary roof when the counters
        # were taken in this run (it is what separates the Newton marches from the conic ones: DESIGN.md section 5)
        fl = (flops or {}).get(rec["name"])
        if fl and fl.get("flops_per_launch"):
            ms = hbm["kernel_ms"]
            tf = fl["flops_per_launch"] / (ms * 1e-3) / 1e12
            valu = fl.get("valu_wave_instructions")
            hbm["secondary"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tf / FP64_VALU_PEAK_TFLOPS, "flops_per_launch": fl["flops_per_launch"],
                                "flops_source": fl["source"],
                                "valu_issue_frac": (valu * 4.0 / (1024 * 2.4e9) / (ms * 1e-3)) if valu else None}
    for k in [k for k in rec if k.startswith("_")]:
        del rec[k]
    return rec


# ------------------------------------------------------------------------------------------------
# end to end: what a caller of the drop-in OpticalSystem.seqtrace experiences ("reported separately")
# ------------------------------------------------------------------------------------------------
class _HostBundle(object):
    """the attributes of the reference's RayBundle that dropin.as_device_bundle reads (raytracer/ray.py:35-75):
    (1, 3, N) host arrays"""

    def __init__(self, o, k, e, wave):
        (self.x, self.k, self.Efield) = (o[None], k[None], e[None])
        self.rayID = np.arange(o.shape[1])
        self.wave = wave


def measure_e2e(dev, rays=10_000_000, small_rays=1000, calls=200):
    """The reference times the whole ``s.seqtrace(...)`` call (demos/demo_benchmark.py:76-78).  Here: the double Gauss of
    configs[1] through ``dropin.seqtrace`` with HOST arrays in and ``list[RayPath]`` out (lazy: results stay on the
    device until looked at), all times wall clock and synchronised:
      first_call_ms             the first call on this bundle size in this (warm) process: table upload, arena buffers of a
                                new size, launch
      h2d_ms                    upload of the bundle (x0 alone: a collimated bundle is recognised as uniform)
      seqtrace_call_ms          host arrays in -> list[RayPath] out (upload + table + launch), steady state (median of 5)
      seqtrace_device_bundle_ms the same call on a bundle that is already on the device
      last_bundle_to_host_ms    first look at the image plane: compaction + D2H of the last bundle's x and k
      full_path_to_host_ms      x and k of every bundle of the path on the host (5.9 GB through page-locked staging)
    and the call latency of an optimiser loop on a small bundle (microseconds per call, back to back): table unchanged /
    one curvature changed (tables cycling through the device cache) / a table never seen before."""
    from pyrate_amd import dropin, systems
    from pyrate_amd.builders import build_rotationally_symmetric_optical_system
    from pyrate_amd.raytracer import _dispatch
    from pyrate_amd.raytracer.ray import RayBundle
    (s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    (o, k, e0) = systems.double_gauss_bundle(rays)
    hb = _HostBundle(o, k, e0, systems.DLINE)
    n = int(o.shape[1])

    def sync():
        torch.cuda.synchronize(dev)

    def timed(fn):
        sync()
        t0 = time.perf_counter()
        out = fn()
        sync()
        return out, (time.perf_counter() - t0) * 1e3
    (_, first_ms) = timed(lambda: dropin.seqtrace(s, hb, seq))               # first call: table upload, arena buffers
    (ib, h2d_ms) = timed(lambda: (lambda b: (b._ensure(), b)[1])(dropin.as_device_bundle(hb, dev)))
    call_ms = sorted(timed(lambda: dropin.seqtrace(s, hb, seq))[1] for _ in range(5))[2]
    dev_ms = sorted(timed(lambda: dropin.seqtrace(s, ib, seq))[1] for _ in range(5))[2]
    rp = dropin.seqtrace(s, ib, seq)
    sync()
    t0 = time.perf_counter()
    last = rp[0].raybundles[-1]
    (x_img, k_img) = (last.x, last.k)
    last_ms = (time.perf_counter() - t0) * 1e3
    n_img = int(x_img.shape[-1])
    del x_img, k_img, last
    rp = dropin.seqtrace(s, ib, seq)
    sync()
    t0 = time.perf_counter()
    nbytes = 0
    for b in rp[0].raybundles:
        nbytes += b.x.nbytes + b.k.nbytes
    full_ms = (time.perf_counter() - t0) * 1e3
    del rp, b
    # ---- small bundle: the optimiser's loop (optimize/optimize.py:73-91)
    (os_, ks_, es_) = systems.double_gauss_bundle(small_rays)
    sb = RayBundle(os_, ks_, es_, wave=systems.DLINE)
    curv = s.elements["stdelem"].surfaces["lens1front"].shape.curvature
    c0 = curv()
    step = [0]

    def per_call_us(pre):
        for _ in range(10):
            if pre:
                pre()
            s.seqtrace(sb, seq)
        sync()
        t0 = time.perf_counter()
        for _ in range(calls):
            if pre:
                pre()
            s.seqtrace(sb, seq)
        sync()
        return (time.perf_counter() - t0) / calls * 1e6

    def change_cached():
        step[0] += 1
        curv.set_value(c0 * (1.0 + 1e-9 * (step[0] % 7)))

    def change_new():
        step[0] += 1
        curv.set_value(c0 * (1.0 + 1e-12 * step[0]))
    small = {"rays": int(os_.shape[1]), "unchanged_us": per_call_us(None), "one_variable_changed_us": per_call_us(change_cached),
             "new_table_us": per_call_us(change_new)}
    curv.set_value(c0)
    _dispatch.clear()
    return {"what": "dropin.seqtrace on the double Gauss of configs[1]: host arrays in -> list[RayPath] out, wall clock, "
                    "synchronised (demos/demo_benchmark.py:76-78 times the whole call)",
            "rays": n, "surfaces": 12, "first_call_ms": first_ms, "h2d_ms": h2d_ms, "seqtrace_call_ms": call_ms,
            "seqtrace_device_bundle_ms": dev_ms, "ops_per_s_whole_call": n * 12 / (call_ms * 1e-3),
            "last_bundle_to_host_ms": last_ms, "rays_at_the_image": n_img,
            "full_path_to_host_ms": full_ms, "full_path_host_bytes": int(nbytes), "small_bundle_call": small}


# ------------------------------------------------------------------------------------------------
# the N = 1 run: headline + the other configurations + PMC passes + scaling point + end to end
# ------------------------------------------------------------------------------------------------
def run_single_gpu(args, dev, watchdog, headline):
    """-> (head record, all records, scaling_point, e2e, arena statistics, wall seconds of the pieces)"""
    from pyrate_amd import placed

    def default_rays(config):
        return args.rays if args.rays is not None else {"aniso": 1_000_000, "aniso_biaxial": 1_000_000,
                                                        "aniso_chain": 20_000}.get(config, 10_000_000)
    configs = [headline]
    full = args.config is None and not args.headline_only and args.mode == "path"
    if full:
        configs += [c for c in SINGLE_GPU_CONFIGS if c != headline]
    if args.configs:
        configs = [c.strip() for c in args.configs.split(",") if c.strip()]
        (args.no_secondary, args.no_scaling_point, args.no_e2e) = (True, True, True)
        full = False
    rays_of = {c: default_rays(c) for c in configs}
    # the other shipped paths ride along with the default run (the biaxial crystal instantiation, the per-surface
    # crystal march, the plugin-granular calls, image mode with fused moments)
    secondary = full and args.rays is None and not args.no_secondary
    recs = []
    timing = {}

    def stage(name):
        watchdog.stage = name
        print("bench.py: %s" % name, file=sys.stderr, flush=True)
        timing[name] = time.perf_counter()
    per_trace = {}
    for c in list(configs):
        stage("measure " + c)
        if c in ("plugin", "surface_step"):    # (--configs plugin: the custom measurements on their own, for experiments)
            recs.append(measure_plugin(args, dev, rays_of[c], with_oracle=not args.no_cpu_baseline, fused=(c == "surface_step")))
            per_trace[c] = (1 if c == "surface_step" else 2) * recs[-1]["surfaces"]
        elif c == "image_moments":
            recs.append(measure_image_moments(args, dev, rays_of[c]))
            configs.remove(c)            # (no PMC pass of its own)
        else:
            recs.append(measure_single(c, args, dev, rays_of[c], with_cpu=not args.no_cpu_baseline))
            if c == "aniso_chain":
                per_trace[c] = 2 * recs[-1]["surfaces"]
    if secondary:
        for c in SECONDARY_MARCH_CONFIGS:
            stage("measure " + c)
            rays_of[c] = 20_000 if c == "aniso_chain" else 1_000_000
            recs.append(measure_single(c, args, dev, rays_of[c], with_cpu=False, verify_oracle=(c == "aniso_chain")))
            configs.append(c)
        per_trace["aniso_chain"] = 2 * recs[-1]["surfaces"]
        stage("measure plugin")
        rays_of["plugin"] = 10_000_000
        recs.append(measure_plugin(args, dev, rays_of["plugin"], with_oracle=not args.no_cpu_baseline))
        per_trace["plugin"] = 2 * recs[-1]["surfaces"]
        configs.append("plugin")
        torch.cuda.empty_cache()
        # The fused step of one surface (DeviceSystem.surface_step / prt_surface_step_rows: 98 B per op).  Written after
        # round 6's last GPU lease: verified on the host build of the sources and through engine.py in the test suite's
        # host mode, compiled for gfx950, never timed.  Its first contact with a device must not cost the
        # line anything: a failure is reported on stderr and the record is simply absent; no PMC pass of its own.
        stage("measure surface_step")
        try:
            step_rec = measure_plugin(args, dev, 10_000_000, with_oracle=not args.no_cpu_baseline, fused=True)
            if step_rec["verified"]["ok"]:
                recs.append(step_rec)
            else:
                print("bench.py: surface_step measured but NOT verified, record dropped: %r" % (step_rec["verified"],),
                      file=sys.stderr)
        except Exception as exc:          # noqa: BLE001
            print("bench.py: surface_step not measured: %r" % (exc,), file=sys.stderr)
        torch.cuda.empty_cache()
        stage("measure image_moments")
        recs.append(measure_image_moments(args, dev, 10_000_000))
        torch.cuda.empty_cache()
    traffic, flops = None, None
    # (a run that is itself being profiled -- rocprofv3 -- python bench.py -- does not start a profiler of its own)
    profiled = any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES",
                                                                        "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_PATH"))
    want_live = args.traffic == "live" or (args.traffic == "auto" and shutil.which("rocprofv3") is not None
                                           and not profiled)
    if want_live:
        stage("PMC passes")
        (traffic, flops) = pmc.measure_live(configs, args, rays_of, args.traffic_timeout, launches_per_trace=per_trace)
        if "error" in traffic:
            print("bench.py: live HBM traffic unavailable: %s" % traffic["error"], file=sys.stderr)
    arena_stats = None
    if args.placement == "arena" and placed.DISABLED is None:
        try:
            arena_stats = placed.PlacedArena.for_device(dev.index).stats()
        except Exception:
            arena_stats = None
    recs = [finish_roofline(r, traffic, flops, lookup=args.traffic != "none") for r in recs]
    scaling_point = None
    if full and not args.no_scaling_point and args.rays is None:
        # the bundle of the multi-GPU protocol (--scaling strong: 1e8 rays, 61 GB of path arrays) on this one GPU
        stage("scaling point (1e8 rays)")
        try:
            # (the arena starts over: the 2 x 29 slabs of this bundle are taken and classified like in a process
            # of their own, not pieced together from what the smaller configurations left cached)
            if args.placement == "arena" and placed.DISABLED is None:
                torch.cuda.synchronize()
                placed.PlacedArena.for_device(dev.index).trim()
            torch.cuda.empty_cache()
            sp = measure_single("doublegauss", args, dev, STRONG_SCALING_RAYS, with_cpu=False, verify_oracle=False)
            scaling_point = {"what": "BASELINE configs[4]'s bundle (--scaling strong: %d rays) traced by ONE GPU, one "
                                     "wavelength: the N = 1 point of the strong-scaling curve" % sp["rays"],
                             "rays_total": sp["rays"], "value": sp["value"], "ms_per_step": sp["ms_per_step"],
                             "kernel_ms": sp["roofline"]["kernel_ms"], "hbm_frac": sp["roofline"]["frac"],
                             "verified": sp["verified"], "output_placement": sp["output_placement"]}
        except (RuntimeError, MemoryError) as exc:          # a device too small / too busy for 66 GB
            scaling_point = {"error": "not measured: %s" % str(exc)[:200]}
        torch.cuda.empty_cache()
    e2e = None
    if full and not args.no_e2e and args.rays is None:
        stage("end to end (drop-in call)")
        try:
            e2e = measure_e2e(dev)
        except (RuntimeError, MemoryError) as exc:
            e2e = {"error": "not measured: %s" % str(exc)[:300]}
    stage("done")
    names = list(timing)
    wall = {names[i]: round(timing[names[i + 1]] - timing[names[i]], 2) for i in range(len(names) - 1)}
    return recs, scaling_point, e2e, arena_stats, wall
