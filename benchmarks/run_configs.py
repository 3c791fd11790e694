#!/usr/bin/env python
"""
Secondary measurements for DESIGN.md (NOT the driver's bench line -- that is bench.py):
the other BASELINE.json configs and the API-level paths, one JSON line each.

    python benchmarks/run_configs.py [--quick]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from pyrate_amd import engine, systems, _lib


def emit(**kw):
    print(json.dumps(kw))
    sys.stdout.flush()


def timed_trace(sysd, x0, k0, e0, mode, iters, warm=5):
    # big isotropic path arrays come from the placement-aware arena (the default of alloc_outputs)
    bufs = sysd.alloc_outputs(x0.shape[1], mode)
    for _ in range(warm):
        sysd.trace_into(x0, k0, bufs, e0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        sysd.trace_into(x0, k0, bufs, e0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    return wall, bufs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    scale = 0.1 if args.quick else 1.0
    iters = 20

    # ---- config 2: double Gauss, path and image mode, on-axis and 5 deg field
    recs = systems.double_gauss_records()
    sysd = engine.DeviceSystem(recs, 0)
    for field in (0.0, 5.0):
        (o, k, e0) = systems.double_gauss_bundle(int(1e7 * scale), field_deg=field)
        n = o.shape[1]
        t0 = time.perf_counter()
        (x0, k0, e0d) = [engine.to_device_rays(a, dev) for a in (o, k, e0)]
        torch.cuda.synchronize()
        h2d = time.perf_counter() - t0
        for (mname, mode) in (("path", _lib.MODE_PATH), ("image", _lib.MODE_IMAGE)):
            (wall, bufs) = timed_trace(sysd, x0, k0, e0d, mode, iters)
            emit(config="double_gauss", field_deg=field, mode=mname, rays=n, surfaces=12,
                 ms=wall * 1e3, ops_per_s=n * 12 / wall, h2d_s=h2d)
        # D2H of the image plane + of the full path (PCIe-inclusive figures)
        (wall, bufs) = timed_trace(sysd, x0, k0, e0d, _lib.MODE_PATH, 3, warm=1)
        t0 = time.perf_counter()
        xh = bufs["x_hit"].cpu()
        kh = bufs["k_out"].cpu()
        vh = bufs["valid"].cpu()
        d2h = time.perf_counter() - t0
        # the same into page-locked host buffers a caller keeps across calls (second pass: buffers exist)
        pinned = [torch.empty(bufs[key].shape, dtype=bufs[key].dtype, pin_memory=True)
                  for key in ("x_hit", "k_out", "valid")]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for (dst, key) in zip(pinned, ("x_hit", "k_out", "valid")):
                dst.copy_(bufs[key], non_blocking=True)
            torch.cuda.synchronize()
            d2h_pinned = time.perf_counter() - t0
        emit(config="double_gauss", field_deg=field, mode="path+PCIe", rays=n, h2d_s=h2d, d2h_path_s=d2h,
             d2h_path_pinned_s=d2h_pinned, ops_per_s_pcie_inclusive=n * 12 / (wall + h2d + d2h),
             ops_per_s_pcie_inclusive_pinned=n * 12 / (wall + h2d + d2h_pinned))
        del bufs, xh, kh, vh, pinned

    # ---- config 3: even asphere (Newton intersect), 4 surfaces
    for (tag, coeffs, curv, cc) in (("mild", (0.0, 1e-7, -1e-10), -1. / 50., -1.),
                                    ("strong", (1e-3, -1e-6, 1e-8), -1. / 30., -1.5)):
        recs = systems.asphere_records(coefficients=coeffs, curv=curv, cc=cc)
        sysd = engine.DeviceSystem(recs, 0)
        for field in (0.0, 5.0):
            (o, k, e0) = systems.double_gauss_bundle(int(1e7 * scale), rpup=9.0, z0=-5.0, field_deg=field)
            n = o.shape[1]
            (x0, k0, e0d) = [engine.to_device_rays(a, dev) for a in (o, k, e0)]
            (wall, bufs) = timed_trace(sysd, x0, k0, e0d, _lib.MODE_PATH, iters)
            v = sysd.views(bufs)
            # residual of the hit points on the asphere
            p = v.x_hit[2] - torch.tensor(recs[2]["g_shape"], dtype=torch.float64, device=dev)[:, None]
            r2 = p[0] ** 2 + p[1] ** 2
            c = recs[2]["shape"]["curv"]
            kk = recs[2]["shape"]["cc"]
            F = c * r2 / (1 + torch.sqrt(1 - c * c * (1 + kk) * r2))
            for (q, a) in enumerate(recs[2]["shape"]["coeffs"]):
                F = F + a * r2 ** (q + 1)
            resid = float((p[2] - F).abs().max())
            emit(config="asphere_" + tag, field_deg=field, rays=n, surfaces=4, ms=wall * 1e3,
                 ops_per_s=n * 4 / wall, max_residual=resid,
                 valid_frac=float(v.valid_out[-1].float().mean()))
            del bufs, v

    # ---- config 4: anisotropic doublet, 1e6 rays in -> 4e6 at the image
    c = systems.CALCITE_TILTED
    eps_sets = {
        "isoeps": (1.5168 ** 2 * np.eye(3), 1.6727 ** 2 * np.eye(3)),
        "uniaxial": (systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
                     systems.uniaxial_eps(1.6727, 1.60, (math.sin(0.2), 0.0, math.cos(0.2)))),
        "biaxial": (np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2]), np.diag([1.62 ** 2, 1.66 ** 2, 1.70 ** 2])),
    }
    for (tag, (e1, e2)) in eps_sets.items():
        recs = systems.aniso_doublet_records(e1, e2)
        sysd = engine.DeviceSystem(recs, 0)
        (o, k) = systems.collimated_bundle(int(1e6 * scale), 11.43, -5.0)
        e0 = np.cross(k, np.array([1., 0., 0.]), axisa=0, axisb=0).T.copy()
        n = o.shape[1]
        (x0, k0, e0d) = [engine.to_device_rays(a, dev) for a in (o, k, e0)]
        (wall, bufs) = timed_trace(sysd, x0, k0, e0d, _lib.MODE_PATH, 10, warm=3)
        (n_in, n_out) = sysd.ray_counts(n)
        emit(config="aniso_doublet_" + tag, rays_in=n, rays_image=n_in[-1], surfaces=5, ms=wall * 1e3,
             ops_per_s=n * 5 / wall, ops_per_s_counting_split_rays=sum(n_in) / wall)
        del bufs

    # ---- API level: OpticalSystem.seqtrace drop-in (flatten + fused launch + lazy compaction
    #      of the image bundle + D2H of its x)
    from pyrate_amd.builders import build_rotationally_symmetric_optical_system
    from pyrate_amd.raytracer.ray import RayBundle
    (s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    for nr in (100, 10000, int(1e6 * scale), int(1e7 * scale)):
        (o, k, e0) = systems.double_gauss_bundle(nr)
        ib = RayBundle(o, k, e0, wave=systems.DLINE)
        for _ in range(3):                 # the first calls after a change of size pay one-off costs
            s.seqtrace(ib, seq)
        torch.cuda.synchronize()
        reps = 20 if nr <= 10000 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            rp = s.seqtrace(ib, seq)
        torch.cuda.synchronize()
        t_call = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        ximg = rp[0].raybundles[-1].x[-1, 0, :]
        t_touch = time.perf_counter() - t0
        emit(config="dropin_seqtrace", rays=o.shape[1], call_ms=t_call * 1e3,
             touch_image_bundle_ms=t_touch * 1e3, ops_per_s_call=o.shape[1] * 12 / t_call,
             n_image=int(ximg.shape[0]))


if __name__ == "__main__":
    main()
