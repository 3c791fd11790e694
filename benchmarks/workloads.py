"""Workloads of bench.py: the BASELINE.json configurations as surface records + a device-resident bundle, their
algorithmic byte counts, and the constants every part of the measurement shares.

Every bundle is the reference's RectGrid disk raster (sampling2d/raster.py:40-60), generated on the device
(bit-identical to the host raster: tests/golden/rasters.json).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz
PREWARM_LAUNCHES = 30
PREWARM_MS = 50.0            # ... and at least this much device time of them (short kernels)
SINGLE_GPU_CONFIGS = ("doublegauss", "asphere", "aniso", "xypoly", "benchmark")
# the other shipped paths, measured beside the BASELINE configurations by the default run:
#   aniso_biaxial   configs[3]'s geometry with two BIAXIAL crystals: the quartic solver of the fused crystal march
#                   (material/material.py:407-454 in the reference)
#   aniso_chain     nine crystal interfaces: more than the fused walk parks -> the per-surface march, two launches
#                   per surface (k_propagate + k_interact_aniso), rays doubling 1 -> 512
#   plugin          the double Gauss through the plugin-granular calls, prt_propagate + prt_interact per surface: the
#                   literal Material.propagate / Surface.intersect / Material.refract loop of
#                   optical_element.py:336-375 (SURVEY 8d's 98 B per ray-surface-op)
#   image_moments   the double Gauss in image mode with the fused spot moments: the optimiser's call
#                   (optimize/optimize.py:73-91: trace + merit), bound by FP64 arithmetic, not by HBM
SECONDARY_MARCH_CONFIGS = ("aniso_biaxial", "aniso_chain")
SECONDARY_CUSTOM_CONFIGS = ("plugin", "image_moments")
VERIFY_TOL = 1e-10            # BASELINE.json north_star: 1e-10 relative on intersection points and direction cosines
STRONG_SCALING_RAYS = 100_000_000   # "1/2/4/8-GPU scaling on a 1e8-ray bundle"
# short names of the workloads for the compact line (the prose is in bench_detail.json)
SHORT_WORKLOAD = {"doublegauss": "demo_doublegauss: 12 spherical Conic surfaces, RectGrid disk bundle (BASELINE configs[1])",
                  "asphere": "demo_asphere: even asphere, Newton intersection (BASELINE configs[2])",
                  "aniso": "demo_anisotropic_doublet: two uniaxial crystals, ray doubling (BASELINE configs[3])",
                  "xypoly": "demo_asphere geometry with an XYPolynomials surface",
                  "benchmark": "demo_benchmark: the reference's own 8-surface benchmark, divergent bundle",
                  "aniso_biaxial": "configs[3] geometry with two biaxial crystals",
                  "aniso_chain": "nine crystal interfaces, per-surface march"}


def make_workload(config, rays, dev, n_gpus=1, rank=0, multi=False, first_segment="uniform", align=1, total_rays=None):
    """records + the device-resident input bundle of one BASELINE configuration.  Every bundle is the
    RectGrid disk raster of the reference, collimated, generated on the device (bit-identical to the host
    raster); rank r owns a contiguous, equal-stride slice of it (pdist.shard_range)."""
    from pyrate_amd import engine, systems
    from pyrate_amd import distributed as pdist
    record_sets = None
    if config == "doublegauss":
        # N = 1: BASELINE configs[1] (d line).  N > 1: configs[4] -- the same lens at the five
        # wavelengths of the prescription (spd:5), per-wavelength indices from the Conrady fit
        # through the (d, F, C) indices; step i traces wavelength i % 5.
        records = systems.double_gauss_records()
        if multi:
            record_sets = [systems.double_gauss_records(w) for w in systems.DOUBLE_GAUSS_WAVES_MM]
        bundle = dict()
        workload = (("demo_doublegauss: 12 spherical Conic surfaces (Rudolph 1897 double Gauss, "
                     "ConstantIndexGlass d-line), RectGrid disk bundle, BASELINE configs[1]") if not multi else
                    ("demo_doublegauss: 12 spherical Conic surfaces (Rudolph 1897 double Gauss), 5 wavelengths "
                     "cycled (Conrady indices), RectGrid disk bundle ray-sharded over the GPUs, BASELINE configs[4]"))
    elif config == "asphere":
        # configs[2]: demo_asphere.py geometry (stop, plane front, even asphere back, image) with the
        # test-suite coefficient set (tests/test_surf_shape.py:115-127) scaled to stay in-domain, bundle
        # radius 9, 5 degree field: the Newton iteration count varies over the wavefront
        records = systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5)
        bundle = dict(rpup=9.0, z0=-5.0, field_deg=5.0)
        workload = ("demo_asphere: stop, plane, even asphere (curv -1/30, cc -1.5, A2..A6 = 1e-3, -1e-6, 1e-8; "
                    "Newton intersection), image; RectGrid disk bundle r = 9 mm at 5 deg, BASELINE configs[2]")
    elif config == "xypoly":
        # the XYPolynomials shape of BASELINE's north_star on the geometry of configs[2]
        records = systems.xypoly_records()
        bundle = dict(rpup=9.0, z0=-5.0, field_deg=5.0)
        workload = ("demo_asphere geometry with an XYPolynomials back surface (12 terms up to degree 4: paraboloid "
                    "-r^2/60 + small terms of every order; Newton intersection); RectGrid disk bundle r = 9 mm at "
                    "5 deg -- the XY-polynomial companion of BASELINE configs[2]")
    elif config == "aniso":
        c = systems.CALCITE_TILTED
        records = systems.aniso_doublet_records(
            systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
            systems.uniaxial_eps(1.6727, 1.60, (np.sin(0.2), 0.0, np.cos(0.2))))
        bundle = dict(rpup=11.43, z0=-5.0)
        workload = ("demo_anisotropic_doublet: cemented doublet of two uniaxial crystals (calcite-like, tilted "
                    "axes), k-vector solve + ray doubling at two interfaces (1 -> 2 -> 4 rays), RectGrid disk "
                    "bundle r = 11.43 mm, BASELINE configs[3]")
    elif config == "aniso_biaxial":
        def rot(ax, ay, az):
            (ca, sa, cb, sb, cg, sg) = (np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az))
            rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
            ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
            rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
            return rz.dot(ry).dot(rx)
        (r1, r2) = (rot(0.4, 0.25, -0.3), rot(-0.2, 0.35, 0.15))
        records = systems.aniso_doublet_records(r1.dot(np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2])).dot(r1.T),
                                                r2.dot(np.diag([1.62 ** 2, 1.66 ** 2, 1.71 ** 2])).dot(r2.T))
        bundle = dict(rpup=11.43, z0=-5.0)
        workload = ("configs[3]'s cemented doublet with two BIAXIAL crystals (principal indices 1.55 / 1.60 / 1.68 and "
                    "1.62 / 1.66 / 1.71, rotated): the quartic k-vector solve of the fused crystal march, 1 -> 2 -> 4 rays")
    elif config == "aniso_chain":
        c = systems.CALCITE_TILTED
        eps = systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"])
        build = [({"shape": "Conic"}, {"decz": 0.0}, None, "stop", {})]
        for q in range(9):
            build.append(({"shape": "Conic", "curv": 0.002 * (q - 4)}, {"decz": 2.0}, {"eps": eps * (1 + 0.01 * q)},
                          "c%d" % q, {}))
        build.append(({"shape": "Conic"}, {"decz": 5.0}, None, "image", {}))
        records = systems.simple_system_records(build)
        bundle = dict(rpup=2.0, z0=-10.0)
        workload = ("nine uniaxial crystal interfaces in a row (more than the fused walk parks): the per-surface march, "
                    "k_propagate + k_interact_aniso per surface, rays doubling 1 -> 512 (11 surfaces)")
    elif config == "benchmark":
        # the reference's own benchmark (demos/demo_benchmark.py:47-78): 8 surfaces, n = 1.7 / 1.5, a DIVERGENT bundle
        # from the origin, half angle 10 degrees, RectGrid raster of angles -- every ray has its own k0 and E0
        records = systems.benchmark_records()
        bundle = dict(radius=systems.BENCHMARK_HALF_ANGLE)
        workload = ("demo_benchmark: the reference's own benchmark system (8 Conic surfaces, n = 1.7 / 1.5 around a "
                    "stop), divergent RectGrid bundle from the origin, half angle 10 deg, per-ray k0 / E0 arrays "
                    "(demos/demo_benchmark.py:47-78; the reference runs it at 1e5 rays)")
    else:
        raise ValueError(config)
    want = total_rays if total_rays is not None else rays * n_gpus
    if config == "benchmark":
        from pyrate_amd.sampling2d import raster as praster
        tables = praster.RectGrid().device_tables(want)
        (x0, k0, e0, n_total) = engine.raster_bundle_device(tables, "divergent", dev, radius=bundle["radius"])
        return dict(config=config, records=records, record_sets=[records], x0=x0, k0=k0, e0=e0, uniform=None,
                    n_total=n_total, n_local=n_total, lo=0, hi=n_total, S=len(records), workload=workload,
                    bundle=bundle, first_segment="arrays")
    (_, n_total) = engine.rect_grid_count(want, dev)
    (lo, hi) = pdist.shard_range(n_total, rank, n_gpus, align)
    uniform = first_segment == "uniform"
    (x0, k0, e0, _) = systems.double_gauss_bundle_device(want, dev, lo=lo, hi=hi, uniform=uniform, **bundle)
    uni = None
    if uniform:
        (uni, k0, e0) = (k0, None, None)
    if config in ("aniso", "aniso_biaxial", "aniso_chain"):            # the crystal marches take tight arrays
        (x0, k0, e0) = [None if t is None else t.contiguous() for t in (x0, k0, e0)]
    return dict(config=config, records=records, record_sets=record_sets or [records], x0=x0, k0=k0, e0=e0,
                uniform=uni, n_total=n_total, n_local=hi - lo, lo=lo, hi=hi, S=len(records), workload=workload,
                bundle=bundle, first_segment=first_segment)


def host_bundle(wl, m):
    """the first m rays of a workload's bundle as host arrays (x0, k0, E0) for the CPU baselines"""
    m = min(m, wl["n_local"])

    def to_host(t):
        # through a page-locked staging array: a 96-MB copy straight into pageable memory makes the runtime pin
        # those pages in place, and the pinned range is torn down again when NumPy frees the array
        stage = torch.empty((3, m), dtype=torch.float64, pin_memory=True)
        stage.copy_(t[:, :m])
        return stage.numpy().copy()
    x = to_host(wl["x0"])
    if wl["uniform"] is not None:
        k = np.repeat(np.array(wl["uniform"].k)[:, None], m, axis=1)
        e = np.repeat(np.array(wl["uniform"].e_re)[:, None], m, axis=1)
    else:
        (k, e) = (to_host(wl["k0"]), to_host(wl["e0"]))
    return np.ascontiguousarray(x), np.ascontiguousarray(k), np.ascontiguousarray(e)


def input_bytes_per_ray(wl):
    """x0 24 B (+ k0 24 B + E0 24 B when the first segment travels as arrays)"""
    return 24 if wl["uniform"] is not None else 72


def algorithmic_bytes(wl, sysd, mode, record_bytes):
    """HBM bytes one launch of the fused march must move: the inputs once; per surface
    x_hit 24 B + k_out 24 B + one byte holding both masks (SURVEY 8d's 49-B ray-surface record; 50 B with the
    masks in two arrays).  Tables with crystals (concatenated layout, real k): per surface x_hit 24 B + mask
    1 B per entering ray and k_out 24 B + mask 1 B per leaving ray (crystal interfaces double the rays)."""
    n = wl["n_local"]
    read = input_bytes_per_ray(wl) * n
    if wl["config"] == "aniso_chain" and mode == "path":
        # the per-surface march: every surface reads the state of the rays that enter it (x, k, mask: 49 B) and writes
        # the record of those that leave (49 B) -- SURVEY 8d's 98 B per op, with the ray count doubling at crystals
        (n_in, n_out) = sysd.ray_counts(n)
        return 49 * (sum(n_in) + sum(n_out))
    if not sysd.all_isotropic:
        (n_in, n_out) = sysd.ray_counts(n)
        return read + (25 * (sum(n_in) + sum(n_out)) if mode == "path" else 25 * (n_in[-1] + n_out[-1]))
    return read + n * record_bytes * (wl["S"] if mode == "path" else 1)



def kernel_label(config):
    if config == "aniso_chain":
        return "k_propagate + k_interact_aniso per surface"
    return "k_trace_general" if config.startswith("aniso") else "k_trace_iso"
