#!/usr/bin/env python
"""Call latency of the drop-in layer on SMALL bundles -- the optimiser loop (thousands of merit
evaluations on 1e2 .. 1e5 rays; reference callers: optimize/optimize.py:73-91, demos/demo_doublegauss.py:137-153).

For the 12-surface double Gauss and bundles of 1e2, 1e3, 1e4, 1e5 rays, per call and in microseconds:

    seqtrace          OpticalSystem.seqtrace (full path, lazy bundles), back to back / synchronised after every call
    image_moments     OpticalSystem.image_moments (image-mode launch that reduces the spot moments itself; the 7 doubles
                      come back to the host, so every call is synchronised by nature)
    prt_trace_moments the C-ABI call under it alone (DeviceSystem.trace_moments_into: table already on the device),
                      back to back -- the floor of this process's host cost per launch

each with the table UNCHANGED between calls and with ONE CURVATURE CHANGED before every call (the optimiser's pattern:
the object graph is re-flattened and the table re-uploaded -- raytracer/_dispatch.py).  `host_us` is the time the host
spends issuing a call (back to back, the device keeps up at these sizes or the figure says so: `device_bound`).

    python benchmarks/call_latency.py > profiles/<tag>_call_latency.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from pyrate_amd import engine, systems, _lib
from pyrate_amd.builders import build_rotationally_symmetric_optical_system
from pyrate_amd.raytracer import _dispatch
from pyrate_amd.raytracer.ray import RayBundle

CALLS = int(os.environ.get("PRT_LATENCY_CALLS", "300"))


def per_call_us(fn, calls, sync_each):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    return {"us": t_all / calls * 1e6, "issue_us": t_issue / calls * 1e6}


def main():
    dev = torch.device("cuda", 0)
    (s, seq) = build_rotationally_symmetric_optical_system(systems.double_gauss_tuples())
    curv = s.elements["stdelem"].surfaces["lens1front"].shape.curvature
    c0 = curv()
    out = {"system": "12-surface double Gauss (builders.build_rotationally_symmetric_optical_system)", "calls": CALLS,
           "sizes": {}}
    for n_req in (100, 1000, 10000, 100000):
        (o, k, e0) = systems.double_gauss_bundle(n_req)
        ib = RayBundle(o, k, e0, wave=systems.DLINE)
        n = o.shape[1]
        step = [0]

        def change():
            step[0] += 1
            curv.set_value(c0 * (1.0 + 1e-9 * (step[0] % 7)))      # seven tables cycle through the device cache ...

        def change_new():
            step[0] += 1
            curv.set_value(c0 * (1.0 + 1e-12 * step[0]))            # ... or every call brings a table never seen before

        def f_seq():
            return s.seqtrace(ib, seq)

        def f_mom():
            return s.image_moments(ib, seq)

        rec = {"rays": n}
        for (label, pre) in (("unchanged", None), ("one_curvature_changed_cached", change), ("one_curvature_changed_new", change_new)):
            def call_seq():
                if pre:
                    pre()
                f_seq()

            def call_mom():
                if pre:
                    pre()
                f_mom()
            rec[label] = {"seqtrace_back_to_back": per_call_us(call_seq, CALLS, False),
                          "seqtrace_synchronised": per_call_us(call_seq, CALLS, True),
                          "image_moments": per_call_us(call_mom, CALLS, False)}
            curv.set_value(c0)
        # the C-ABI call alone
        sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
        x0 = ib._x[-1]
        bufs = sysd.alloc_outputs(n, _lib.MODE_IMAGE, packed_flags=True)
        ws = engine.MomentsWorkspace(dev, n_results=1, n_rays=n)
        from pyrate_amd.raytracer.optical_system import _first_segment
        first = _first_segment(ib)            # uniform (k, E) of a collimated bundle, or the arrays
        k0 = first.pop("k0")
        rec["first_segment"] = "uniform" if first.get("uniform") is not None else "arrays"

        def f_abi():
            sysd.trace_moments_into(x0, k0, bufs, ws, slot=0, **first)
        rec["prt_trace_moments_alone_back_to_back"] = per_call_us(f_abi, CALLS, False)
        launch = sysd.launcher(x0, k0, sysd.alloc_outputs(n, _lib.MODE_PATH, packed_flags=True), **first)
        rec["prt_trace_ex_prebuilt_arguments_back_to_back"] = per_call_us(launch, CALLS, False)
        out["sizes"][str(n_req)] = rec
        if os.environ.get("PRT_LATENCY_PROFILE") and n_req == 1000:      # where the host's time goes (stderr)
            import cProfile, pstats
            def f_new():
                change_new()
                f_seq()
            for (tag, fn) in (("seqtrace", f_seq), ("image_moments", f_mom), ("seqtrace with a table never seen before", f_new)):
                prof = cProfile.Profile()
                prof.enable()
                for _ in range(500):
                    fn()
                prof.disable()
                torch.cuda.synchronize()
                print("==== %s, 500 calls" % tag, file=sys.stderr)
                pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(22)
        _dispatch.clear()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
