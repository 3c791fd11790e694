"""cost of the fused image-plane moments: plain trace vs trace+moments (kernel-level, HIP events)"""
import sys, time, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
recs = systems.double_gauss_records()
sysd = engine.DeviceSystem(recs, 0)
(x0, k0, e0, _) = systems.double_gauss_bundle_device(10_000_000, dev)
n = x0.shape[1]
bufs = sysd.alloc_outputs(n, _lib.MODE_PATH)
ws = engine.MomentsWorkspace(dev, n_results=2, n_rays=n)
def timed(fn, iters=50):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for rep in range(2):
    print("plain   %.4f ms" % timed(lambda: sysd.trace_into(x0, k0, bufs, e0)))
    print("moments %.4f ms" % timed(lambda: sysd.trace_moments_into(x0, k0, bufs, ws, 0, e0)))
    v = sysd.views(bufs)
    print("two-pass stats alone %.4f ms" % timed(lambda: (engine.bundle_moments_async(v.x_hit[-1], v.valid_out[-1], ws, 0),
                                                         engine.bundle_moments_async(v.x_hit[-1], v.valid_out[-1], ws, 1, ref_dev=ws.out[0], ref_kind=2))))

# ---- the bench's per-step choreography, piece by piece ---------------------------------------
import os
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
bufs2 = [bufs, sysd.alloc_outputs(n, _lib.MODE_PATH)]
wss = [ws, engine.MomentsWorkspace(dev, n_results=2, n_rays=n)]
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
done = [None, None]
def step_factory(use_events, use_allreduce):
    cnt = [0]
    def step():
        b = cnt[0] % 2
        cnt[0] += 1
        if use_events and done[b] is not None:
            main.wait_event(done[b])
        sysd.trace_moments_into(x0, k0, bufs2[b], wss[b], 0, e0)
        if use_events:
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if use_allreduce:
                    dist.all_reduce(wss[b].out[0])
                d = torch.cuda.Event(); d.record(side)
                done[b] = d
    return step
def wall(fn, iters=50):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3
print("double-buffered moments, no side stream      %.4f ms" % wall(step_factory(False, False)))
print("  + events / side stream (no collective)     %.4f ms" % wall(step_factory(True, False)))
print("  + all_reduce(7 doubles) on the side stream %.4f ms" % wall(step_factory(True, True)))
t = torch.zeros(7, dtype=torch.float64, device=dev)
print("all_reduce alone (7 doubles, world 1)        %.4f ms" % wall(lambda: dist.all_reduce(t)))
dist.destroy_process_group()
