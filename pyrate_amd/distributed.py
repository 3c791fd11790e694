"""
Multi-GPU: rays are independent, so a bundle is sharded into
contiguous slices of the initial (3, N) arrays, one process per GPU, and traced
with no data-path collective.  The only exchange step is the final image-plane
gather: every rank contributes (x_img, k_img) (6, n) float64 + valid (n) uint8
= 49 B/ray and receives the whole image plane (RCCL all-gather over xGMI, row by row
straight into the final [row][global ray] layout; ``torch.distributed`` backend "nccl"
is RCCL on ROCm; "gloo" is used by the CPU tests).  rayIDs are implicit: rank r owns
``shard_range(N, r, world)``.
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


def shard_stride(n_total, world, align=1):
    """rays per rank: every rank but the trailing one(s) owns exactly this many.  ``align``: round the stride up to
    a multiple (512 = one row pitch unit: rank r's slot of a gathered row then starts on a 4-KiB boundary, which
    lets the trace write its image plane straight into that slot, ImagePlaneGather.own_rows)"""
    if world <= 0:
        return 0
    stride = -(-n_total // world)
    return -(-stride // align) * align if align > 1 else stride


def shard_range(n_total, rank, world, align=1):
    """Contiguous, ordered slices of EQUAL STRIDE: rank r owns [r*n_pad, min((r+1)*n_pad, N)) with
    n_pad = ceil(N / world).  All ranks together are short by world*n_pad - N < world rays; for N >> world
    that is the last rank alone, for tiny bundles several trailing ranks can be short or EMPTY (N = 9,
    world = 8: ranks 0..3 own 2 rays, rank 4 one, ranks 5..7 none) -- callers must cope with zero-size
    shards (the trace, the gather and the statistics do).  With one common stride, ray i of rank r is
    global ray r*n_pad + i, so an all-gather of one row lands in global ray order as it is
    (ImagePlaneGather needs no repacking)."""
    n_pad = shard_stride(n_total, world, align)
    lo = min(rank * n_pad, n_total)
    return lo, min(lo + n_pad, n_total)


def shard_sizes(n_total, world, align=1):
    return [shard_range(n_total, r, world, align)[1] - shard_range(n_total, r, world, align)[0] for r in range(world)]


def _row_of(t2d, row, start, n, n_pad):
    """1-d tensor of n_pad elements that starts at t2d[row, start]: the n rays of a shard's row plus
    whatever follows them in memory up to the common stride (padding of a pitched row, the next
    branch, the next row -- never looked at by anybody).  None if the storage ends before that."""
    if n_pad == n:
        return t2d[row, start:start + n] if (n == 0 or t2d.stride(1) == 1) else None   # collectives need contiguous rows
    off = t2d.storage_offset() + row * t2d.stride(0) + start * t2d.stride(1)
    room = t2d.untyped_storage().nbytes() // t2d.element_size() - off
    if t2d.stride(1) != 1 or room < n_pad:
        return None
    return t2d.as_strided((n_pad,), (1,), off)


# How the rows of a bundle's all-gather are issued -- an EXPLICIT switch (round 5; until round 4 the default went
# through torch's private ``dist._coalescing_manager``, a code path only RCCL took, never exercised with more than one
# rank and invisible to the gloo tests):
#   "coalesced" (default)  ``ProcessGroup.allgather_into_tensor_coalesced`` -- the PUBLIC batched entry point of the
#                          process group: one ncclGroupStart / End around the rows with RCCL, one call with gloo.  The
#                          CPU tests (gloo, world 2 / 3) and the RCCL runs take the same lines.
#   "single"               one ``all_gather_into_tensor`` per row (seven launches per isotropic bundle)
#   "manager"              the round-3 / 4 form through ``dist._coalescing_manager`` (opt-in: a private API)
# PRT_GATHER_BATCH selects; the legacy PRT_GATHER_COALESCE=0 still means "single".  A mode is decided once per
# process, before anything is enqueued: a rank never re-issues its collectives in another form after a failure (it
# would run a different sequence than its peers and hang or mismatch them) -- errors propagate.
GATHER_BATCH_MODES = ("coalesced", "single", "manager")
_BATCH_MODE = [None]


def gather_batch_mode():
    if _BATCH_MODE[0] is None:
        mode = os.environ.get("PRT_GATHER_BATCH", "").strip().lower()
        if not mode:
            mode = "single" if os.environ.get("PRT_GATHER_COALESCE", "1") == "0" else "coalesced"
        if mode not in GATHER_BATCH_MODES:
            raise ValueError("PRT_GATHER_BATCH must be one of %s, not %r" % (", ".join(GATHER_BATCH_MODES), mode))
        _BATCH_MODE[0] = mode
    return _BATCH_MODE[0]


def set_gather_batch_mode(mode):
    """choose the mode from code (tests: both forms in one process, results compared bit for bit); None: back to
    the environment's choice"""
    if mode is not None and mode not in GATHER_BATCH_MODES:
        raise ValueError(mode)
    _BATCH_MODE[0] = mode


def _all_gather_rows(pairs, group, device):
    """all_gather_into_tensor of every (destination row, source) pair, asynchronous; returns the work handles
    (each has ``wait()``).  See ``gather_batch_mode`` for the three forms; element types are grouped (the byte row of
    the masks travels as a row of doubles where it can, so an isotropic bundle is ONE group)."""
    mode = gather_batch_mode()
    if mode == "single" or not pairs:
        return [dist.all_gather_into_tensor(dst, src, group=group, async_op=True) for (dst, src) in pairs]
    work = []
    for dtype in sorted(set(dst.dtype for (dst, _) in pairs), key=str):      # one group per element type
        sel = [(dst, src) for (dst, src) in pairs if dst.dtype == dtype]
        if mode == "coalesced":
            pg = group if group is not None else dist.group.WORLD
            work.append(pg.allgather_into_tensor_coalesced([dst for (dst, _) in sel], [src for (_, src) in sel]))
        else:
            if dist.get_backend(group) != "nccl":
                raise ValueError("PRT_GATHER_BATCH=manager is the RCCL-only form (torch's private coalescing manager); "
                                 "use 'coalesced' or 'single' with the %s backend" % dist.get_backend(group))
            with dist._coalescing_manager(group=group, async_ops=True) as cm:
                for (dst, src) in sel:
                    dist.all_gather_into_tensor(dst, src, group=group)
            work.append(cm)
    return work


class ImagePlaneGather(object):
    """All-gather of the image-plane arrays of a ray-sharded trace, straight into the final layout.

    The receive buffers ARE the result: ``recv_f`` (rows, branches, world * n_pad) float64 and
    ``recv_v`` (branches, world * n_pad) uint8, with n_pad = ``shard_stride``.  Every row of a rank's
    image-plane arrays is the input of one ``all_gather_into_tensor`` whose output is the matching
    row of the receive buffer -- rank r's rays land at [r*n_pad, r*n_pad + n_r), which is global ray
    order, so there is no send-side packing and no reassembly afterwards (7 collectives of
    8 B x n_pad per rank for an isotropic bundle, issued back to back on one stream).  The inputs
    are views of the trace's own output arrays; a row whose storage ends before the common stride
    (the last rank's last row in a tight layout) goes through a small padded staging row instead.
    ``finish()`` returns (x (3,N), k (3,N), valid (N,)) views in global ray order.

    Crystals (SURVEY.md 8e "Anisotropic"): every anisotropic interface doubles the rays inside
    the shard, so a rank holds ``branches`` * n_local image points laid out [branch][local ray]
    (the engine's dense doubling order, = the reference's hstack of the two solutions applied
    once per crystal).  ``finish()`` returns [branch][global ray] -- exactly the arrays a
    single-GPU trace of the whole bundle produces (a view when N is a multiple of the world
    size, one compacting copy otherwise) -- and ``ray_id()`` / ``branch()`` give the join keys
    (rayID = global index of the initial ray).  ``with_fields`` adds the E vectors (re, im).
    """

    def __init__(self, n_total, device, group=None, stage_on_host=False, branches=1,
                 with_fields=False, world=None, rank=None, align=1):
        """stage_on_host: exchange through host buffers (for the ``gloo`` backend, which cannot
        all-gather device tensors; used by the single-GPU dry run of the multi-rank bench path --
        the production path is RCCL on device buffers).  ``world`` / ``rank``: override the process
        group's (single-process emulation of a sharded run, see ``deposit``)."""
        self.group = group
        self.stage_on_host = stage_on_host
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.n_total = n_total
        self.branches = int(branches)
        self.rows = 12 if with_fields else 6
        self.align = align
        self.sizes = shard_sizes(n_total, self.world, align)
        self.n_pad = shard_stride(n_total, self.world, align)
        self.device = device
        self.bdev = torch.device("cpu") if stage_on_host else device
        (m, r) = (self.branches, self.rows)
        self.recv_f = torch.zeros((r, m, self.world * self.n_pad), dtype=torch.float64, device=self.bdev)
        self.recv_v = torch.zeros((m, self.world * self.n_pad), dtype=torch.uint8, device=self.bdev)
        self._stage = {}          # (row, branch) -> padded staging row, only where a view cannot serve
        self._work = []

    def _sources(self, x_img, k_img, valid, e_re, e_im, n):
        """per (row, branch): the n_pad-element input of its collective"""
        m = self.branches
        assert x_img.shape[1] == m * n, "shard size mismatch"
        arrays = [x_img, k_img] + ([e_re, e_im] if self.rows == 12 else [])
        v2d = valid.reshape(1, -1)
        out = []
        for b in range(m):
            for (row, (t, c)) in enumerate([(t, c) for t in arrays for c in range(3)] + [(v2d, 0)]):
                src = None if self.stage_on_host else _row_of(t, c, b * n, n, self.n_pad)
                if src is None:
                    key = (row, b)
                    st = self._stage.get(key)
                    if st is None:
                        st = self._stage[key] = torch.zeros(self.n_pad, dtype=t.dtype, device=self.bdev)
                    st[:n].copy_(t[c, b * n:(b + 1) * n], non_blocking=True)
                    src = st
                out.append((row, b, src))
        return out

    def _dest(self, row, b):
        return self.recv_v[b] if row == self.rows else self.recv_f[row, b]

    def own_rows(self):
        """(x (3, n), k (3, n), mask (n,)): this rank's slot of the receive buffers as row-pitched views (row pitch
        world * n_pad).  A trace that writes its image plane THERE (``bufs["image_rows"]``, prt_trace_ex's redirect)
        needs no copy of its own rows: ``start_in_place()`` then runs the collectives in place (NCCL / RCCL: send
        buffer = the rank's slot of the receive buffer).  Isotropic bundles (one branch) on device buffers; the
        stride must make the slot 16-B aligned and even (``align`` = 512 does)."""
        if self.branches != 1 or self.stage_on_host:
            raise ValueError("own_rows: one branch, device buffers")
        if (self.rank * self.n_pad) % 2 or (self.world * self.n_pad) % 2:
            raise ValueError("own_rows needs an even stride (construct the gather and the shards with align=512)")
        (lo, n) = (self.rank * self.n_pad, self.sizes[self.rank])
        return self.recv_f[0:3, 0, lo:lo + n], self.recv_f[3:6, 0, lo:lo + n], self.recv_v[0, lo:lo + n]

    def start_in_place(self):
        """the all-gather of rows that already sit in this rank's slot (own_rows): in place, nothing is staged or
        copied locally; with a single rank there is nothing to do at all"""
        self._work = []
        if not _collectives_needed(self.group):
            return
        (lo, n_pad) = (self.rank * self.n_pad, self.n_pad)
        pairs = [(self.recv_f[row, 0], self.recv_f[row, 0, lo:lo + n_pad]) for row in range(self.rows)]
        mask = self.recv_v[0]
        if n_pad % 8 == 0 and mask.data_ptr() % 8 == 0:
            # the mask row travels as n_pad / 8 doubles: all rows are of ONE element type and go out as one group
            m8 = mask.view(torch.float64)
            pairs.append((m8, m8[lo // 8:(lo + n_pad) // 8]))
        else:
            pairs.append((mask, mask[lo:lo + n_pad]))
        self._work = _all_gather_rows(pairs, self.group, self.device)

    def deposit(self, rank, x_img, k_img, valid, e_re=None, e_im=None):
        """what the collectives do with rank ``rank``'s contribution, as local copies: the
        single-rank path, and the way one process emulates a sharded run shard by shard"""
        n = self.sizes[rank]
        for (row, b, src) in self._sources(x_img, k_img, valid, e_re, e_im, n):
            self._dest(row, b)[rank * self.n_pad:rank * self.n_pad + n].copy_(src[:n], non_blocking=True)

    def start(self, x_img, k_img, valid, e_re=None, e_im=None):
        n = self.sizes[self.rank]
        if not _collectives_needed(self.group):
            self.deposit(self.rank, x_img, k_img, valid, e_re, e_im)
            self._work = []
            return
        srcs = self._sources(x_img, k_img, valid, e_re, e_im, n)
        if x_img.is_cuda:
            # the collectives read the trace's own output arrays on THIS stream: if they live in the placement arena,
            # their release must wait for it (placed.record_stream; no-op for torch-allocated arrays)
            from . import placed
            for t in (x_img, k_img, valid):
                placed.record_stream(t)
        if self.stage_on_host and x_img.is_cuda:
            torch.cuda.current_stream(x_img.device).synchronize()     # D2H staging complete
        self._work = _all_gather_rows([(self._dest(row, b), src) for (row, b, src) in srcs], self.group,
                                      self.bdev)

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []

    def _gathered(self):
        self.wait()
        (m, r, n) = (self.branches, self.rows, self.n_total)
        f = self.recv_f[:, :, :n]
        v = self.recv_v[:, :n]
        if m == 1:
            return f[:, 0], v[0]                 # (rows, N) view, row pitch world * n_pad
        return f.reshape(r, m * n), v.reshape(m * n)

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


class DirectImagePlaneGather(ImagePlaneGather):
    """The image-plane exchange as direct peer writes (one node, one process per GPU): every rank maps the receive
    buffers of all its peers (IPC handles, exchanged once through the process group) and ``start_in_place()`` copies
    the rank's own slot -- which the march filled through ``own_rows()`` -- into the same slot of every peer's buffer:
    one contiguous device-to-device copy per row and peer, the peers served in parallel on one stream each, rank r
    starting with peer r + 1.  xGMI is point to point (one link per pair of GPUs), so the seven copies of a row use
    seven different links at once; there is no ring, no intermediate hop and no collective kernel on the compute
    units -- the copies are the runtime's (copy engines).  SURVEY.md 8(e) "prefer the direct / one-shot algorithm".

    Completion: the copies of THIS rank are done when the streams have run (``start_in_place`` joins them into the
    current stream); that all PEERS have finished writing into this rank's buffer is known after any collective that
    every rank issues behind its copies on the same stream -- the fused statistics' all-reduce
    (``SpotStatistics.reduce()`` after ``start_in_place()``), or ``fence()``.  A buffer must not be filled again
    before the local consumer is done with it AND a fence has passed in between (two gathers used alternately, each
    step fenced, satisfy this by construction).

    Isotropic bundles, device buffers, ``align`` = 512 (as for ``own_rows``)."""

    def __init__(self, n_total, device, group=None, align=512):
        super().__init__(n_total, device, group=group, align=align)
        if not dist.is_initialized():
            raise ValueError("DirectImagePlaneGather needs an initialised process group")
        # Bringing the peer mappings up can fail on ONE rank (no IPC between two devices, no memory): every rank walks
        # through the same collectives whatever happened locally, learns what happened elsewhere, and all of them
        # raise together -- a rank that left early would leave its peers waiting in a collective for good.
        mine = None
        try:
            mine = self._export_handles()
        except Exception as exc:          # noqa: BLE001 -- reported to the peers below, then raised by all
            self._local_error = "export: %s" % str(exc)[:200]
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
        missing = [r for (r, h) in enumerate(handles) if h is None]
        self.peer_f, self.peer_v, self._streams = {}, {}, {}
        ok = not missing
        if ok:
            try:
                for (r, h) in enumerate(handles):
                    if r != self.rank:
                        # tensors in THIS process whose storage is rank r's buffer (on rank r's GPU; written over xGMI)
                        (self.peer_f[r], self.peer_v[r]) = self._open_peer(h)
                self._make_streams(device)
            except Exception as exc:      # noqa: BLE001
                self._local_error = "open: %s" % str(exc)[:200]
                ok = False
        opened = [None] * self.world
        dist.all_gather_object(opened, ok, group=group)
        if missing or not all(opened):
            self.peer_f, self.peer_v, self._streams = {}, {}, {}
            raise RuntimeError("DirectImagePlaneGather: the peer buffers could not be brought up on every rank "
                               "(no export on ranks %s, no mapping on ranks %s%s)"
                               % (missing, [r for (r, o) in enumerate(opened) if not o],
                                  "; here: " + self._local_error if getattr(self, "_local_error", None) else ""))
        dist.barrier(group=group)          # every rank has opened every buffer before anybody writes

    _local_error = None

    def _export_handles(self):
        """what a peer needs to map this rank's receive buffers (torch's CUDA IPC)"""
        from torch.multiprocessing.reductions import reduce_tensor
        return (reduce_tensor(self.recv_f), reduce_tensor(self.recv_v))

    @staticmethod
    def _open_peer(handle):
        (hf, hv) = handle
        return hf[0](*hf[1]), hv[0](*hv[1])

    def _make_streams(self, device):
        for r in self.peer_f:
            self._streams[r] = torch.cuda.Stream(device=device)
        self._ready = torch.cuda.Event()
        self._done = {r: torch.cuda.Event() for r in self.peer_f}
        self._token = torch.zeros(1, dtype=torch.float32, device=device)

    def start_in_place(self):
        """copies of this rank's slot into every peer's buffer; enqueued behind the work of the current stream,
        which afterwards waits for them"""
        self._work = []
        (lo, n_pad) = (self.rank * self.n_pad, self.n_pad)
        cur = torch.cuda.current_stream(self.device)
        self._ready.record(cur)
        for step in range(1, self.world):
            r = (self.rank + step) % self.world
            (st, pf, pv) = (self._streams[r], self.peer_f[r], self.peer_v[r])
            with torch.cuda.stream(st):
                st.wait_event(self._ready)
                for row in range(self.rows):
                    pf[row, 0, lo:lo + n_pad].copy_(self.recv_f[row, 0, lo:lo + n_pad], non_blocking=True)
                pv[0, lo:lo + n_pad].copy_(self.recv_v[0, lo:lo + n_pad], non_blocking=True)
                self._done[r].record(st)
        for r in self._done:
            cur.wait_event(self._done[r])

    def start(self, x_img, k_img, valid, e_re=None, e_im=None):
        """for arrays that are not in the receive buffer yet: one local copy into the own slot, then as above"""
        self.deposit(self.rank, x_img, k_img, valid, e_re, e_im)
        self.start_in_place()

    def fence(self):
        """after this (stream-ordered with RCCL; host-synchronising with gloo) every rank's copies issued before
        its own fence() have arrived"""
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(self._token, group=self.group)
        else:
            torch.cuda.current_stream(self.device).synchronize()
            dist.barrier(group=self.group)


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

    def trace_and_start(self, sysd, x0, k0, bufs, e0_re=None, e0_im=None, uniform=None):
        """Fused form: the trace kernel itself reduces the shard's moments about the vertex of the
        last surface (prt_trace_moments); ONE 7-double all-reduce combines the shards.  Call on the
        stream the trace should run on; ``reduce()`` may be issued on another stream afterwards.
        ``uniform``: the bundle's uniform first segment instead of k0 / e0 (engine.UniformFirst)."""
        self._fused_ref = sysd.moments_reference()
        return sysd.trace_moments_into(x0, k0, bufs, self.ws, slot=0, e0_re=e0_re, e0_im=e0_im, uniform=uniform)

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
