"""
Multi-GPU: rays are independent, so a bundle is sharded into
contiguous slices of the initial (3, N) arrays, one process per GPU, and traced
with no data-path collective.  The only exchange step is the final image-plane
gather: every rank contributes (x_img, k_img) (6, n) float64 + valid (n) uint8
= 49 B/ray and receives the whole image plane (RCCL all-gather over xGMI;
``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU
tests).  rayIDs are implicit: rank r owns ``shard_range(N, r, world)``.
"""
import os

import torch
import torch.distributed as dist


def _collectives_needed(group=None):
    """more than one rank -- or PRT_FORCE_COLLECTIVES=1 with an initialised process group, which
    sends the single-rank case through the collectives too (RCCL smoke test on a 1-GPU box)"""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("PRT_FORCE_COLLECTIVES", "0") == "1"


def shard_range(n_total, rank, world):
    """contiguous, ordered, near-equal slices: [lo, hi) of rank ``rank``"""
    base = n_total // world
    rem = n_total % world
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


class ImagePlaneGather(object):
    """All-gather of the image-plane arrays of a ray-sharded trace.

    Buffers are allocated once for ``n_local_max`` rays per rank (shards differ by at
    most one ray; the tail is padding).  ``start()`` packs on the caller's current
    stream and launches the two collectives asynchronously; ``finish()`` returns
    (x (3,N), k (3,N), valid (N,)) views in global ray order.

    Crystals (SURVEY.md 8e "Anisotropic"): every anisotropic interface doubles the rays inside
    the shard, so a rank holds ``branches`` * n_local image points laid out [branch][local ray]
    (the engine's dense doubling order, = the reference's hstack of the two solutions applied
    once per crystal).  Concatenating shards would give [rank][branch][ray]; ``finish()``
    returns [branch][global ray] instead -- exactly the arrays a single-GPU trace of the whole
    bundle produces -- and ``ray_id()`` / ``branch()`` give the join keys (rayID = global index
    of the initial ray).  ``with_fields`` adds the E vectors (re, im) to the exchange.
    """

    def __init__(self, n_total, device, group=None, stage_on_host=False, branches=1,
                 with_fields=False):
        """stage_on_host: exchange through pinned host buffers (for the ``gloo`` backend, which
        cannot all-gather device tensors; used by the single-GPU dry run of the multi-rank
        bench path -- the production path is RCCL on device buffers)."""
        self.group = group
        self.stage_on_host = stage_on_host
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.branches = int(branches)
        self.rows = 12 if with_fields else 6
        self.sizes = shard_sizes(n_total, self.world)
        self.n_max = max(self.sizes) if self.sizes else 0
        self.device = device
        bdev = torch.device("cpu") if stage_on_host else device
        (m, r) = (self.branches, self.rows)
        self.send_f = torch.zeros((r, m, self.n_max), dtype=torch.float64, device=bdev)
        self.send_v = torch.zeros((m, self.n_max), dtype=torch.uint8, device=bdev)
        self.recv_f = torch.empty((self.world, r, m, self.n_max), dtype=torch.float64, device=bdev)
        self.recv_v = torch.empty((self.world, m, self.n_max), dtype=torch.uint8, device=bdev)
        self._work = []

    def start(self, x_img, k_img, valid, e_re=None, e_im=None):
        m = self.branches
        n = x_img.shape[1] // m
        assert n == self.sizes[self.rank] and x_img.shape[1] == m * n, "shard size mismatch"
        self.send_f[0:3, :, :n].copy_(x_img.view(3, m, n), non_blocking=True)
        self.send_f[3:6, :, :n].copy_(k_img.view(3, m, n), non_blocking=True)
        if self.rows == 12:
            self.send_f[6:9, :, :n].copy_(e_re.view(3, m, n), non_blocking=True)
            self.send_f[9:12, :, :n].copy_(e_im.view(3, m, n), non_blocking=True)
        self.send_v[:, :n].copy_(valid.view(m, n), non_blocking=True)
        if self.stage_on_host and x_img.is_cuda:
            torch.cuda.current_stream(x_img.device).synchronize()     # D2H staging complete
        if not _collectives_needed(self.group):
            self.recv_f[0].copy_(self.send_f, non_blocking=True)
            self.recv_v[0].copy_(self.send_v, non_blocking=True)
            self._work = []
            return
        w1 = dist.all_gather_into_tensor(self.recv_f.view(-1), self.send_f.view(-1),
                                          group=self.group, async_op=True)
        w2 = dist.all_gather_into_tensor(self.recv_v.view(-1), self.send_v.view(-1),
                                          group=self.group, async_op=True)
        self._work = [w1, w2]

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []

    def _gathered(self):
        self.wait()
        (m, r) = (self.branches, self.rows)
        if all(s == self.n_max for s in self.sizes):
            # [rank][row][branch][ray] -> [row][branch][rank][ray] = [row][branch][global ray]
            f = self.recv_f.permute(1, 2, 0, 3).reshape(r, m * self.world * self.n_max)
            v = self.recv_v.permute(1, 0, 2).reshape(-1)
        else:
            f = torch.cat([self.recv_f[q, :, :, :s] for (q, s) in enumerate(self.sizes)],
                          dim=2).reshape(r, m * self.n_total)
            v = torch.cat([self.recv_v[q, :, :s] for (q, s) in enumerate(self.sizes)],
                          dim=1).reshape(-1)
        return f, v

    def finish(self):
        (f, v) = self._gathered()
        return f[0:3], f[3:6], v

    def finish_with_fields(self):
        """(x, k, valid, E_re, E_im) -- needs with_fields=True"""
        assert self.rows == 12
        (f, v) = self._gathered()
        return f[0:3], f[3:6], v, f[6:9], f[9:12]

    def ray_id(self):
        """rayID of every gathered column (global index of its initial ray)"""
        return torch.arange(self.n_total, dtype=torch.int64).repeat(self.branches)

    def branch(self):
        """doubling branch of every gathered column: bit a = solution picked at crystal a"""
        return torch.arange(self.branches, dtype=torch.int64).repeat_interleave(self.n_total)


def global_spot_statistics(x_img, valid=None, group=None, moments_fn=None):
    """Centroid and RMS spot radius (about the centroid) of a ray-sharded image plane without
    gathering it: every rank reduces its shard on the device (prt_bundle_moments), then two
    7-double all-reduces combine the shards ("all-reduce of (sum x, sum x^2,
    count) replaces the all-gather").  Same formulas as RayBundleAnalysis
    (analysis/ray_analysis.py:44-86).  Returns (count, centroid (3,), rms)."""
    import numpy as np
    if moments_fn is None:
        from . import engine
        moments_fn = engine.bundle_moments
    dev = x_img.device

    def allreduce(vec):
        if not _collectives_needed(group):
            return vec
        t = torch.tensor(vec, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    (cnt, s1, _) = moments_fn(x_img, valid)
    tot = allreduce(np.concatenate(([cnt], s1)))
    count = tot[0]
    centroid = tot[1:4] / (count + 1e-17)
    (_, _, s2) = moments_fn(x_img, valid, centroid)
    tot2 = allreduce(np.concatenate(([0.0], s2)))
    rms = float(np.sqrt(np.sum(tot2[1:4]) / (count - 1 + 1e-17)))
    return count, centroid, rms


class SpotStatistics(object):
    """Asynchronous, device-resident version of ``global_spot_statistics`` for stream pipelines.
    Two forms: ``start`` enqueues (current stream) pass 1 -> all-reduce -> pass 2 about the global
    centroid -> all-reduce over existing image-plane arrays; ``trace_and_start`` + ``reduce`` lets
    the trace kernel produce the moments (no extra pass over the arrays) and needs one all-reduce.
    No host synchronisation in either; ``result`` syncs and returns (count, centroid (3,), rms).
    One instance per in-flight bundle."""

    def __init__(self, device, group=None, n_rays=0):
        from . import engine
        self.engine = engine
        self.group = group
        self.ws = engine.MomentsWorkspace(device, n_results=2, n_rays=n_rays)
        self.multi = _collectives_needed(group)
        self._fused_ref = None

    def start(self, x_img, valid):
        eng = self.engine
        self._fused_ref = None
        m1 = eng.bundle_moments_async(x_img, valid, self.ws, slot=0)
        if self.multi:
            dist.all_reduce(m1, op=dist.ReduceOp.SUM, group=self.group)     # stream-ordered (NCCL)
        m2 = eng.bundle_moments_async(x_img, valid, self.ws, slot=1, ref_dev=m1, ref_kind=2)
        if self.multi:
            dist.all_reduce(m2, op=dist.ReduceOp.SUM, group=self.group)

    def trace_and_start(self, sysd, x0, k0, bufs, e0_re=None, e0_im=None):
        """Fused form: the trace kernel itself reduces the shard's moments about the vertex of the
        last surface (prt_trace_moments); ONE 7-double all-reduce combines the shards.  Call on the
        stream the trace should run on; ``reduce()`` may be issued on another stream afterwards."""
        self._fused_ref = sysd.moments_reference()
        return sysd.trace_moments_into(x0, k0, bufs, self.ws, slot=0, e0_re=e0_re, e0_im=e0_im)

    def reduce(self):
        """the all-reduce of the fused form (current stream)"""
        if self.multi:
            dist.all_reduce(self.ws.out[0], op=dist.ReduceOp.SUM, group=self.group)

    def result(self):
        import numpy as np
        if self._fused_ref is not None:
            return self.engine.spot_from_moments(self.ws.out[0].cpu().numpy(), self._fused_ref)
        m1 = self.ws.out[0].cpu().numpy()
        m2 = self.ws.out[1].cpu().numpy()
        count = m1[0]
        centroid = m1[1:4] / (count + 1e-17)
        rms = float(np.sqrt(np.sum(m2[4:7]) / (count - 1 + 1e-17)))
        return count, centroid, rms
