"""``-m gpu`` test of the multi-GPU path with RCCL: one process per GPU (world = min(device count, 8)),
ray-sharded trace, spot-statistics all-reduce and the image-plane all-gather straight into the final
layout -- checked bit for bit against the unsharded trace.  Skipped on boxes with fewer than 2 GPUs (the
1-GPU development boxes); the same code path runs there through ``bench.py --force-multi`` (RCCL with a
single rank) and, with two ranks on CPU tensors, in ``tests/test_distributed_gloo.py``."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank(rank, world, port, nrays, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from pyrate_amd import distributed as pdist, engine, systems, _lib
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        sysd = engine.DeviceSystem(systems.double_gauss_records(), rank)
        (_, n_total) = engine.rect_grid_count(nrays, dev)
        (lo, hi) = pdist.shard_range(n_total, rank, world)
        (x0, k0, e0d, _) = systems.double_gauss_bundle_device(nrays, dev, field_deg=2.0, lo=lo, hi=hi)
        pitch = engine.recommended_pitch(pdist.shard_stride(n_total, world))
        bufs = sysd.alloc_outputs(hi - lo, packed_flags=True, pitch=pitch)
        stats = pdist.SpotStatistics(dev, n_rays=hi - lo)
        stats.trace_and_start(sysd, x0, k0, bufs, e0d)
        stats.reduce()
        v = sysd.views(bufs)
        g = pdist.ImagePlaneGather(n_total, dev)
        g.start(v.x_hit[-1], v.k_out[-1], v.flags[-1])
        (gx, gk, gf) = g.finish()
        (cnt, cen, rms) = stats.result()
        # the unsharded trace, on this rank's GPU
        (xa, ka, ea, _) = systems.double_gauss_bundle_device(nrays, dev, field_deg=2.0)
        whole = sysd.trace(xa, ka, ea, packed_flags=True)

        def same(a, b):
            return torch.equal(a.contiguous().view(torch.int64), b.contiguous().view(torch.int64))
        ok = same(gx, whole.x_hit[-1]) and same(gk, whole.k_out[-1]) and torch.equal(gf, whole.flags[-1])
        m = whole.valid_out[-1].bool()
        xs = whole.x_hit[-1][:, m]
        cen_ref = xs.mean(dim=1).cpu().numpy()
        rms_ref = float(torch.sqrt(((xs - xs.mean(dim=1, keepdim=True)) ** 2).sum() / (int(m.sum()) - 1)))
        ok = ok and cnt == int(m.sum()) and bool(np.allclose(cen, cen_ref, rtol=0, atol=1e-10)) \
            and abs(rms - rms_ref) < 1e-10
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_trace_with_rccl_gather_equals_the_unsharded_trace():
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs at least 2 GPUs (RCCL, one process per GPU)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, 2000003, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]
