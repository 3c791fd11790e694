"""world_size-2 ``gloo`` tests (CPU) of the multi-GPU path: ray sharding and the
image-plane all-gather reassembly in global ray order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyrate_amd import distributed as pdist


def test_shard_ranges_partition_the_bundle():
    for n in (0, 1, 7, 8, 1000, 9994476):
        for world in (1, 2, 3, 8):
            spans = [pdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b) in zip(spans[:-1], spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for (lo, hi) in spans]
            # equal stride: rank r starts at r * n_pad (clamped), so a gathered row is in global order
            n_pad = pdist.shard_stride(n, world)
            assert world * n_pad - n < world
            assert all(lo == min(r * n_pad, n) for (r, (lo, _)) in enumerate(spans))
            assert sizes == [max(0, min(n_pad, n - r * n_pad)) for r in range(world)]
            assert sizes == pdist.shard_sizes(n, world)
            # an aligned common stride: still a partition, every slot starts on a multiple of the alignment
            spans_a = [pdist.shard_range(n, r, world, align=512) for r in range(world)]
            assert spans_a[0][0] == 0 and spans_a[-1][1] == n and all(a[1] == b[0] for (a, b) in zip(spans_a[:-1], spans_a[1:]))
            assert pdist.shard_stride(n, world, 512) % 512 == 0 and pdist.shard_stride(n, world, 512) >= n_pad
            assert all(lo == min(r * pdist.shard_stride(n, world, 512), n) for (r, (lo, _)) in enumerate(spans_a))
    # tiny bundles: several trailing ranks can be short or empty (zero-size shards are legal)
    assert pdist.shard_sizes(9, 8) == [2, 2, 2, 2, 1, 0, 0, 0]
    assert pdist.shard_sizes(1, 2) == [1, 0]


def test_gather_batch_mode_is_an_explicit_switch(monkeypatch):
    pdist.set_gather_batch_mode(None)
    monkeypatch.delenv("PRT_GATHER_BATCH", raising=False)
    monkeypatch.delenv("PRT_GATHER_COALESCE", raising=False)
    assert pdist.gather_batch_mode() == "coalesced"               # the public batched entry point is the default
    pdist.set_gather_batch_mode(None)
    monkeypatch.setenv("PRT_GATHER_COALESCE", "0")                # the legacy switch still means one by one
    assert pdist.gather_batch_mode() == "single"
    pdist.set_gather_batch_mode(None)
    monkeypatch.setenv("PRT_GATHER_BATCH", "manager")             # the private coalescing manager: opt-in only
    assert pdist.gather_batch_mode() == "manager"
    pdist.set_gather_batch_mode(None)
    monkeypatch.setenv("PRT_GATHER_BATCH", "sometimes")
    with pytest.raises(ValueError):
        pdist.gather_batch_mode()
    pdist.set_gather_batch_mode(None)


def test_gather_rows_must_be_contiguous():
    """a collective needs a contiguous input row: a view with a stride along the rays goes through staging"""
    t = torch.arange(24, dtype=torch.float64).reshape(3, 8)
    assert pdist._row_of(t, 1, 2, 4, 4).tolist() == [10., 11., 12., 13.]
    assert pdist._row_of(t[:, ::2], 1, 0, 4, 4) is None
    g = pdist.ImagePlaneGather(4, torch.device("cpu"), world=1, rank=0)
    g.deposit(0, t[:, ::2], t[:, 1::2], torch.ones(4, dtype=torch.uint8))
    (x, k, v) = g.finish()
    assert torch.equal(x, t[:, ::2]) and torch.equal(k, t[:, 1::2]) and int(v.sum()) == 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, q, batch_mode="coalesced"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    pdist.set_gather_batch_mode(batch_mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(123)
        x = rng.rand(3, n_total)
        k = rng.rand(3, n_total)
        v = (rng.rand(n_total) > 0.3).astype(np.uint8)
        (lo, hi) = pdist.shard_range(n_total, rank, world)
        g = pdist.ImagePlaneGather(n_total, torch.device("cpu"))
        g.start(torch.from_numpy(x[:, lo:hi].copy()), torch.from_numpy(k[:, lo:hi].copy()),
                torch.from_numpy(v[lo:hi].copy()))
        (gx, gk, gv) = g.finish()
        ok = bool(np.array_equal(gx.numpy(), x) and np.array_equal(gk.numpy(), k)
                  and np.array_equal(gv.numpy(), v))
        # second round re-uses the buffers
        g.start(torch.from_numpy(2 * x[:, lo:hi]), torch.from_numpy(k[:, lo:hi].copy()),
                torch.from_numpy(v[lo:hi].copy()))
        (gx, _, _) = g.finish()
        ok = ok and bool(np.array_equal(gx.numpy(), 2 * x))
        # in place: shards of an aligned common stride write their rows straight into their slot of the receive
        # buffer (what the march's image-plane redirect does on the GPU), the collectives run in place
        (lo_a, hi_a) = pdist.shard_range(n_total, rank, world, align=512)
        ga = pdist.ImagePlaneGather(n_total, torch.device("cpu"), align=512)
        (ox, ok_, ov) = ga.own_rows()
        assert ox.shape == (3, hi_a - lo_a) and ga.n_pad % 512 == 0
        ox.copy_(torch.from_numpy(x[:, lo_a:hi_a].copy()))
        ok_.copy_(torch.from_numpy(k[:, lo_a:hi_a].copy()))
        ov.copy_(torch.from_numpy(v[lo_a:hi_a].copy()))
        ga.start_in_place()
        (gx, gk, gv) = ga.finish()
        ok = ok and bool(np.array_equal(gx.numpy(), x) and np.array_equal(gk.numpy(), k)
                         and np.array_equal(gv.numpy(), v))
        # sharded spot statistics == statistics of the whole bundle (NumPy stand-in for the
        # device reduction: this test runs without a GPU; the all-reduce plumbing is what it checks)
        def np_moments(xs, mask=None, ref=None):
            a = xs.numpy()
            m = np.ones(a.shape[1], dtype=bool) if mask is None else mask.numpy().astype(bool)
            r = np.zeros(3) if ref is None else np.asarray(ref)
            dlt = a[:, m] - r[:, None]
            return float(m.sum()), dlt.sum(axis=1), (dlt ** 2).sum(axis=1)
        (cnt, cen, rms) = pdist.global_spot_statistics(torch.from_numpy(x[:, lo:hi].copy()),
                                                       torch.from_numpy(v[lo:hi].copy()), moments_fn=np_moments)
        m = v.astype(bool)
        if m.sum() < 2:                  # (the one-ray bundle of the empty-shard case: no spread to compare)
            q.put((rank, ok and cnt == m.sum()))
            return
        cen_ref = x[:, m].sum(axis=1) / m.sum()
        rms_ref = np.sqrt(((x[:, m] - cen_ref[:, None]) ** 2).sum() / (m.sum() - 1))
        ok = ok and cnt == m.sum() and bool(np.allclose(cen, cen_ref, rtol=1e-13)) and abs(rms - rms_ref) < 1e-13
        # one-pass form of the fused trace kernel: shard moments about a common reference point,
        # ONE all-reduce, then spot_from_moments
        from pyrate_amd.engine import spot_from_moments
        ref = np.array([0.4, 0.5, 0.6])
        (c1, s1, s2) = np_moments(torch.from_numpy(x[:, lo:hi].copy()), torch.from_numpy(v[lo:hi].copy()), ref)
        t = torch.from_numpy(np.concatenate(([c1], s1, s2)))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        (cnt1, cen1, rms1) = spot_from_moments(t.numpy(), ref)
        ok = ok and cnt1 == m.sum() and bool(np.allclose(cen1, cen_ref, rtol=1e-13)) and abs(rms1 - rms_ref) < 1e-12
        # crystals: rays double inside the shard ([branch][local ray]); the gather returns the
        # single-GPU order [branch][global ray] with rayID / branch join keys
        for (m, fields) in ((2, False), (4, True)):
            gx = rng.rand(3, m, n_total)
            gk = rng.rand(3, m, n_total)
            ge = rng.rand(6, m, n_total)
            gv = (rng.rand(m, n_total) > 0.5).astype(np.uint8)
            def loc(a):
                return torch.from_numpy(np.ascontiguousarray(a[..., lo:hi]).reshape(a.shape[:-2] + (-1,)))
            g2 = pdist.ImagePlaneGather(n_total, torch.device("cpu"), branches=m, with_fields=fields)
            if fields:
                g2.start(loc(gx), loc(gk), loc(gv), loc(ge[0:3]), loc(ge[3:6]))
                (ax, ak, av, aer, aei) = g2.finish_with_fields()
                ok = ok and bool(np.array_equal(aer.numpy(), ge[0:3].reshape(3, -1))
                                 and np.array_equal(aei.numpy(), ge[3:6].reshape(3, -1)))
            else:
                g2.start(loc(gx), loc(gk), loc(gv))
                (ax, ak, av) = g2.finish()
            ok = ok and bool(np.array_equal(ax.numpy(), gx.reshape(3, -1))
                             and np.array_equal(ak.numpy(), gk.reshape(3, -1))
                             and np.array_equal(av.numpy(), gv.reshape(-1)))
            rid = g2.ray_id().numpy()
            br = g2.branch().numpy()
            ok = ok and bool(np.array_equal(ax.numpy()[0], gx[0][br, rid]))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch_mode", ["coalesced", "single"])
@pytest.mark.parametrize("n_total", [1000, 1001, 1])
def test_image_plane_gather_world2_gloo(n_total, batch_mode):
    """n_total = 1: rank 1 owns an empty shard.  batch_mode: the two public forms the rows of a bundle can go out in
    (pyrate_amd.distributed.gather_batch_mode) -- "coalesced" is the default of every backend, so these CPU ranks run
    the very lines an RCCL rank runs (ProcessGroup.allgather_into_tensor_coalesced)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q, batch_mode)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _strong_worker(rank, world, port, q):
    """one rank of a strong-scaling split: trace the shard of ONE bundle (the CPU oracle stands in for the device
    march -- per-ray scalar code, so a ray's result does not depend on the shard it is in), all-gather the image
    plane, hand back a digest of the gathered arrays"""
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import seqtrace_c
        from pyrate_amd import systems
        recs = systems.double_gauss_records()
        (o, k, e0) = systems.double_gauss_bundle(3000, rpup=20.0, field_deg=12.0)       # (rays miss and vignette: masks matter)
        n_total = o.shape[1]
        (lo, hi) = pdist.shard_range(n_total, rank, world, align=512)                 # bench.py's alignment
        out = seqtrace_c.trace(recs, np.ascontiguousarray(o[:, lo:hi]), np.ascontiguousarray(k[:, lo:hi]),
                               np.ascontiguousarray(e0[:, lo:hi]))[-1]
        g = pdist.ImagePlaneGather(n_total, torch.device("cpu"), align=512)
        g.start(torch.from_numpy(np.ascontiguousarray(out["x_hit"])), torch.from_numpy(np.ascontiguousarray(out["k_out"])),
                torch.from_numpy(np.ascontiguousarray(out["valid_out"]).astype(np.uint8)))
        (ax, ak, av) = g.finish()
        h = hashlib.sha256()
        for a in (ax, ak, av):
            h.update(np.ascontiguousarray(a.numpy()).tobytes())
        q.put((rank, n_total, h.hexdigest()))
    finally:
        dist.destroy_process_group()


def test_strong_scaling_split_of_one_bundle_gathers_to_the_same_plane_at_every_world_size():
    """bench.py --scaling strong: ONE bundle split N ways.  The gathered image plane (hit points, wave vectors, masks;
    global ray order) is bit-identical at world sizes 1, 2 and 3 -- 2944 rays do not divide evenly into 512-aligned shards by either -- and on every
    rank, and equals the unsharded trace."""
    import hashlib
    from oracle import seqtrace_c
    from pyrate_amd import systems
    recs = systems.double_gauss_records()
    (o, k, e0) = systems.double_gauss_bundle(3000, rpup=20.0, field_deg=12.0)
    whole = seqtrace_c.trace(recs, o, k, e0)[-1]
    assert 0 < int(np.count_nonzero(whole["valid_out"])) < o.shape[1]
    h = hashlib.sha256()
    for a in (whole["x_hit"], whole["k_out"], np.asarray(whole["valid_out"]).astype(np.uint8)):
        h.update(np.ascontiguousarray(a).tobytes())
    ctx = mp.get_context("spawn")
    for world in (1, 2, 3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_strong_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=180) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert sorted(r[0] for r in results) == list(range(world))
        assert all(r[1] == o.shape[1] and r[2] == h.hexdigest() for r in results), (world, results)


class _FakeDirectGather(pdist.DirectImagePlaneGather):
    """DirectImagePlaneGather without CUDA IPC: the handles are plain tuples, a 'peer buffer' is a local tensor; a
    chosen rank fails at a chosen stage -- what the bring-up protocol is there for"""
    fail = (None, None)          # (stage, rank)

    def _export_handles(self):
        if self.fail == ("export", self.rank):
            raise RuntimeError("no IPC handle on this rank")
        return ("handle of rank", self.rank)

    def _open_peer(self, handle):
        if self.fail == ("open", self.rank):
            raise RuntimeError("peer %d cannot be mapped here" % handle[1])
        return torch.zeros_like(self.recv_f), torch.zeros_like(self.recv_v)

    def _make_streams(self, device):
        self._streams = {r: None for r in self.peer_f}


def _bringup_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        outcomes = []
        for fail in ((None, None), ("export", 1), ("open", 0), ("open", world - 1)):
            _FakeDirectGather.fail = fail
            try:
                g = _FakeDirectGather(4096, torch.device("cpu"))
                outcomes.append(("up", sorted(g.peer_f)))
            except RuntimeError as exc:
                outcomes.append(("refused", "could not be brought up on every rank" in str(exc)))
        # ... and the ranks are still in step: a collective after all of that completes
        t = torch.ones(1)
        dist.all_reduce(t)
        q.put((rank, outcomes, float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_direct_gather_bring_up_fails_on_all_ranks_or_on_none(world):
    """The peer-write form of the image-plane exchange needs every rank to export its receive buffers and to map every
    peer's.  If ONE rank cannot (no IPC between two devices, no memory), ALL ranks must learn it and raise together --
    a rank that raised alone would leave its peers waiting inside a collective for good, and bench.py's start-up probe
    (--exchange auto) relies on every rank dropping the direct form before anybody uses it.  Fake handles, real
    collectives (gloo): a failure at either stage on any rank -> every rank refuses, nobody hangs, and the next
    collective still completes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bringup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for (rank, outcomes, total) in results:
        assert outcomes[0] == ("up", [r for r in range(world) if r != rank])
        assert outcomes[1:] == [("refused", True)] * 3
        assert total == world
