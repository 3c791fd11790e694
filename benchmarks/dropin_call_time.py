"""drop-in OpticalSystem.seqtrace at 1e7 rays vs the bench kernel: the COLD START (first call of a fresh process:
library load, table upload, the arena's hunt for three kinds of HBM, first launch -- everything a user's first
seqtrace pays), then one call at a time (host work + kernel, synchronised) and back to back (the host work of
call i+1 overlaps the kernel of call i: what an optimiser loop or a wavelength sweep sees).

    python benchmarks/dropin_call_time.py > profiles/<tag>_dropin_call_time.json
"""
import json, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyrate_amd import engine, systems, placed
from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer.ray import RayBundle
t_import = time.perf_counter() - T0
(s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
(o, k, e0) = systems.double_gauss_bundle(10000000)
torch.cuda.init()
torch.zeros(1, device="cuda").item()                 # HIP context up: not the engine's cost
t0 = time.perf_counter()
ib = RayBundle(o, k, e0, wave=systems.DLINE)          # host arrays of a collimated bundle: recognised as uniform
torch.cuda.synchronize()
t_bundle = time.perf_counter() - t0
t0 = time.perf_counter()
rp = s.seqtrace(ib, seq)                              # FIRST call: table upload + arena hunt + launch
torch.cuda.synchronize()
t_first = time.perf_counter() - t0
arena_cold = placed.PlacedArena.for_device(0).stats()
t0 = time.perf_counter()
x_img = rp[0].raybundles[-1].x                           # first look at a result: compaction + D2H of the image plane
t_first_result = time.perf_counter() - t0
del rp, x_img
t0 = time.perf_counter()
rp = s.seqtrace(ib, seq)
torch.cuda.synchronize()
t_second = time.perf_counter() - t0
del rp
for _ in range(5):
    rp = s.seqtrace(ib, seq); del rp
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rp = s.seqtrace(ib, seq)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    del rp
for _ in range(6):                      # two result sets are alive in this loop: let the arena build the second
    rp = s.seqtrace(ib, seq)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    rp = s.seqtrace(ib, seq)
torch.cuda.synchronize()
back_to_back = (time.perf_counter() - t0) * 1e3 / 20
# the kernel alone into the same kind of arrays
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
x0 = ib._x[-1]
bufs = sysd.alloc_outputs(o.shape[1], packed_flags=True)
sysd.trace_timed(x0, None, bufs, 10, uniform=ib._uniform)
kernel_ms = sysd.trace_timed(x0, None, bufs, 30, uniform=ib._uniform)
# ---- BASELINE configs[3] through the drop-in layer: the crystal doublet at 1e6 rays (lazy E fields: the trace itself
# computes no eigenvectors; the first look at an E field traces once more with them)
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import systems_zoo as zoo
api = zoo.mirror_api()
c = systems.CALCITE_TILTED
(sc, seqc) = zoo.aniso_doublet(api, systems.uniaxial_eps(c["n_o"], c["n_e"], c["axis"]),
                               systems.uniaxial_eps(1.6727, 1.60, (np.sin(0.2), 0.0, np.cos(0.2))))
(oc, kc) = systems.collimated_bundle(1000000, 11.43, -5.0)
ec = np.ascontiguousarray(np.cross(kc, np.array([1., 0., 0.]), axisa=0, axisb=0).T)
ibc = api.RayBundle(x0=oc, k0=kc, Efield0=ec, wave=systems.DLINE)
for _ in range(5):
    rpc = sc.seqtrace(ibc, seqc)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    rpc = sc.seqtrace(ibc, seqc)
t_issue = (time.perf_counter() - t0) * 1e3 / 20
torch.cuda.synchronize()
crystal_ms = (time.perf_counter() - t0) * 1e3 / 20
t0 = time.perf_counter()
e_last = rpc[0].raybundles[-1].Efield
crystal_first_efield_ms = (time.perf_counter() - t0) * 1e3
crystal = {"rays_in": int(oc.shape[1]), "rays_at_the_image": int(e_last.shape[2]),
           "dropin_seqtrace_ms_back_to_back": crystal_ms, "host_issue_ms": t_issue,
           "first_look_at_an_E_field_ms": crystal_first_efield_ms}
print(json.dumps({"crystal_doublet": crystal, "cold_start": {"import_s": t_import, "bundle_upload_ms": t_bundle * 1e3,
                                 "first_seqtrace_ms": t_first * 1e3, "second_seqtrace_ms": t_second * 1e3,
                                 "first_look_at_the_image_plane_ms": t_first_result * 1e3,
                                 "arena_after_first_call": arena_cold,
                                 "uniform_bundle": ib._uniform is not None},
                  "dropin_seqtrace_call_ms_synchronised": [round(t, 4) for t in ts],
                  "dropin_seqtrace_ms_back_to_back": back_to_back, "kernel_ms": kernel_ms, "rays": o.shape[1],
                  "arena": placed.PlacedArena.for_device(0).stats()}))
