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
    """one rank; whatever happens, exactly one (rank, ok, message) goes into the queue"""
    try:
        q.put((rank, bool(_rank_body(rank, world, port, nrays)), ""))
    except BaseException as exc:      # the parent must hear about it instead of waiting for its timeout
        import traceback
        q.put((rank, False, "%r\n%s" % (exc, traceback.format_exc())))


def _rank_body(rank, world, port, nrays):
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
        return ok
    finally:
        dist.destroy_process_group()


def test_sharded_trace_with_rccl_gather_equals_the_unsharded_trace():
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs at least 2 GPUs (RCCL, one process per GPU)")
    _run_ranks(world, 2000003, 420)


def _run_ranks(world, nrays, timeout):
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    # daemon processes, terminated in any case: a rank stuck in a collective must not keep the test run alive
    procs = [ctx.Process(target=_rank, args=(r, world, port, nrays, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            try:
                results.append(q.get(timeout=timeout))
            except queue.Empty:
                break
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    assert len(results) == world, "ranks that did not answer within the timeout: %s" % (
        sorted(set(range(world)) - {r[0] for r in results}),)
    assert all(ok for (_, ok, _) in results), [r for r in results if not r[1]]


def test_the_rank_program_with_a_single_rccl_rank():
    """the very program every rank of the multi-GPU test runs, as a world of one (RCCL communicator, sharding,
    fused statistics + all-reduce, all-gather into the final layout, comparison with the unsharded trace): what
    a 1-GPU box can check of it"""
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    _run_ranks(1, 500003, 300)
