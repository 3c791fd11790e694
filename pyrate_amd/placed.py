"""Placement-aware device memory for the path arrays: host side of ``prt_arena_*`` (include/prt.h).

The HBM of an MI355X consists of three kinds of physical memory; the path-mode march writes at
7.0 TB/s when ``x_hit`` and ``k_out`` lie in two different kinds and at 5.6-5.7 TB/s when they share
one (DESIGN.md section 5 "Placement"; ``benchmarks/vmm_placement_probe.hip``).  ``hipMalloc`` / the torch
allocator give no control over that, so the engine takes the memory for large path arrays from a
``PlacedArena``, which builds every buffer from 1-GiB physical slabs of one known kind.

The buffers are handed out as ordinary torch tensors (zero copy) that own their arena buffer: when the
last view dies the buffer goes back to the arena, which keeps it mapped for the next trace of that size.
"""
import ctypes
import os
import threading
import weakref

import torch

from . import _lib

SLAB_BYTES = 1 << 30
MAX_BLOCK_SLABS = 512          # no buffer is longer than the device (record_stream's walk down to a block's base)
# below this many output bytes a trace keeps using the torch allocator: small path arrays live in the
# L2 / Infinity Cache for most of the march, and a 1-GiB slab per array would be mostly padding
PLACED_MIN_BYTES = int(os.environ.get("PRT_PLACED_MIN_BYTES", 512 << 20))
# input bundles (x0, k0, E0) from this many bytes per (3, N) array on are uploaded into arena memory of
# a kind the output arrays do not use (loads that share a kind with the write streams cost the march 5 %)
PLACED_INPUT_MIN_BYTES = int(os.environ.get("PRT_PLACED_INPUT_MIN_BYTES", 96 << 20))
OUTPUT_KINDS_MASK = 0b011       # a two-part output request takes kinds 0 and 1


# set (with the reason) when the arena turned out to be unusable on this machine -- e.g. a driver without
# the HIP virtual-memory API; automatic placement then stays with the torch allocator for the process.
# PRT_ARENA=off switches the arena off from the start (every array from the torch allocator).
DISABLED = "PRT_ARENA=%s" % os.environ["PRT_ARENA"] if os.environ.get("PRT_ARENA", "").lower() in ("off", "0", "no") \
    else None


def disable(reason):
    global DISABLED
    if DISABLED is None:
        DISABLED = str(reason)
        import warnings
        warnings.warn("pyrate_amd: placement-aware memory switched off, path arrays come from the torch "
                      "allocator (%s)" % reason, RuntimeWarning)


def record_stream(tensor, stream=None):
    """``PlacedArena.record_stream`` for whatever device the tensor lives on (no-op outside the arena)"""
    if DISABLED is not None or not tensor.is_cuda:
        return False
    arena = PlacedArena._instances.get(tensor.device.index)
    return arena.record_stream(tensor, stream) if arena is not None else False


def trim_all():
    """hand every arena's cached (unused) memory back to the driver -- what the engine does when a torch
    allocation runs out of memory (the caching allocator cannot reclaim what the arena holds)"""
    row_pool.clear()
    for arena in list(PlacedArena._instances.values()):
        try:
            arena.trim()
        except Exception:
            pass


class InputRows(object):
    """Row-pitched (3, pitch) blocks for big input bundles, nine rows (x0, k0, E0) per arena buffer."""

    def __init__(self):
        self._cur = {}            # device index -> [uint8 buffer, pitch, rows in the buffer, rows handed out]
        self._lock = threading.Lock()

    def take(self, device, pitch, nrows=3):
        with self._lock:
            ent = self._cur.get(device.index)
            if ent is None or ent[1] != pitch or ent[3] + nrows > ent[2]:
                (parts, _) = PlacedArena.for_device(device.index).alloc([9 * pitch * 8], n_distinct=1,
                                                                         avoid_mask=OUTPUT_KINDS_MASK)
                ent = self._cur[device.index] = [parts[0], pitch, 9, 0]
            lo = ent[3] * pitch * 8
            ent[3] += nrows
            return ent[0][lo:lo + nrows * pitch * 8].view(torch.float64).view(nrows, pitch)


input_rows = InputRows()


class RowPool(object):
    """Arrays of the PER-SURFACE calls (DeviceSystem.propagate / interact on big bundles), carved front to back out of
    1-GiB arena buffers of a CHOSEN kind: a (3, pitch) array of 1e7 rays is a quarter of a slab, so four of them share
    one.  ``take`` avoids the kinds its caller names -- the kinds of the arrays the same kernel READS -- so that a
    kernel's two input arrays and its output array lie in three different kinds of HBM whenever the device offers them
    (loads that share a kind with the write stream cost the stream ~5 %).  A buffer goes back to the
    arena when the last array carved from it dies (and the pool has moved on to another one); within a buffer no byte
    is handed out twice."""

    def __init__(self):
        self._cur = {}            # (device index, kind) -> [uint8 buffer, bytes handed out]
        self._lock = threading.Lock()

    def take(self, device, nbytes, avoid_kinds=()):
        """(uint8 tensor of ``nbytes`` bytes at a 4-KiB aligned address, its kind)"""
        nbytes = -(-int(nbytes) // 4096) * 4096
        avoid = set(int(q) for q in avoid_kinds if q is not None and q >= 0)
        with self._lock:
            for ((d, kind), ent) in self._cur.items():
                if d == device.index and kind not in avoid and ent[1] + nbytes <= ent[0].numel():
                    lo = ent[1]
                    ent[1] += nbytes
                    return ent[0][lo:lo + nbytes], kind
            mask = 0
            for q in avoid:
                mask |= 1 << q
            size = max(SLAB_BYTES, -(-nbytes // SLAB_BYTES) * SLAB_BYTES)
            (parts, kinds) = PlacedArena.for_device(device.index).alloc([size], n_distinct=1, avoid_mask=mask)
            self._cur[(device.index, kinds[0])] = [parts[0], nbytes]
            return parts[0][:nbytes], kinds[0]

    def clear(self):
        with self._lock:
            self._cur.clear()


row_pool = RowPool()


class _Block(object):
    """One arena buffer; exposes ``__cuda_array_interface__`` so that torch can wrap it without a copy
    and keeps a reference to this object for as long as any tensor uses the memory."""

    def __init__(self, arena, ptr, nbytes, kind):
        self.arena = arena
        self.ptr = ptr
        self.nbytes = nbytes
        self.kind = kind
        self.streams = set()          # streams other than the allocating one that used the memory (record_stream)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2,
                "strides": None}

    def __del__(self):
        try:
            self.arena._release(self.ptr, self.streams)
        except Exception:           # interpreter shutdown: the process takes the memory with it
            pass


# ---- DLPack route (used if torch cannot adopt the pointer through __cuda_array_interface__) --------
class _DLDevice(ctypes.Structure):
    _fields_ = [("device_type", ctypes.c_int), ("device_id", ctypes.c_int)]


class _DLDataType(ctypes.Structure):
    _fields_ = [("code", ctypes.c_uint8), ("bits", ctypes.c_uint8), ("lanes", ctypes.c_uint16)]


class _DLTensor(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("device", _DLDevice), ("ndim", ctypes.c_int),
                ("dtype", _DLDataType), ("shape", ctypes.POINTER(ctypes.c_int64)),
                ("strides", ctypes.POINTER(ctypes.c_int64)), ("byte_offset", ctypes.c_uint64)]


class _DLManagedTensor(ctypes.Structure):
    pass


_DLDeleter = ctypes.CFUNCTYPE(None, ctypes.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", ctypes.c_void_p),
                             ("deleter", _DLDeleter)]
_KDLROCM = 10
_dl_alive = {}            # address of the managed tensor -> (managed tensor, shape array, block)
_dl_lock = threading.RLock()    # re-entrant: a garbage collection inside _dlpack_tensor may run _dl_delete


@_DLDeleter
def _dl_delete(mt_ptr):
    with _dl_lock:
        _dl_alive.pop(ctypes.addressof(mt_ptr.contents), None)


def _dlpack_tensor(block, device_index):
    shape = (ctypes.c_int64 * 1)(block.nbytes)
    mt = _DLManagedTensor()
    mt.dl_tensor.data = block.ptr
    mt.dl_tensor.device = _DLDevice(_KDLROCM, device_index)
    mt.dl_tensor.ndim = 1
    mt.dl_tensor.dtype = _DLDataType(1, 8, 1)           # kDLUInt, 8 bits
    mt.dl_tensor.shape = shape
    mt.dl_tensor.strides = None
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _dl_delete
    with _dl_lock:
        _dl_alive[ctypes.addressof(mt)] = (mt, shape, block)
    new_capsule = ctypes.pythonapi.PyCapsule_New
    new_capsule.restype = ctypes.py_object
    new_capsule.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
    capsule = new_capsule(ctypes.addressof(mt), b"dltensor", None)
    return torch.utils.dlpack.from_dlpack(capsule)


class PlacedArena(object):
    """``prt_arena_t`` of one device (one per process and device: ``PlacedArena.for_device``)."""

    _instances = {}
    _instances_lock = threading.Lock()

    def __init__(self, device_index):
        self.lib = _lib.load()
        self.device_index = int(device_index)
        handle = ctypes.c_void_p()
        _lib.check(self.lib.prt_arena_create(self.device_index, ctypes.byref(handle)))
        self._h = handle
        self._wrap = os.environ.get("PRT_PLACED_WRAP", "")       # "", "cai" or "dlpack"
        self._blocks = {}                                        # base pointer -> weak reference to its _Block
        # _release runs from _Block.__del__ -- any thread, any garbage-collection point, also in the middle of alloc
        # or record_stream on this thread: every access to _blocks goes through this (re-entrant) lock
        self._blocks_lock = threading.RLock()

    @classmethod
    def for_device(cls, device_index):
        with cls._instances_lock:
            arena = cls._instances.get(device_index)
            if arena is None:
                arena = cls._instances[device_index] = cls(device_index)
            return arena

    def _release(self, ptr, streams=()):
        # prt_arena_free records an event on ONE stream instead of waiting; the next user of the memory is ordered
        # behind it.  That stream is the current one at the time the last tensor dies -- made to wait first for
        # every other stream the memory was used on (record_stream: a gather on a side stream, ...), so that
        # the release is behind all of its users whatever stream the garbage collector happens to run under.
        with self._blocks_lock:
            self._blocks.pop(ptr, None)
        if self._h:
            cur = torch.cuda.current_stream(self.device_index)
            for s in streams:
                if s != cur:
                    cur.wait_stream(s)
            self.lib.prt_arena_free(self._h, ctypes.c_void_p(ptr), ctypes.c_void_p(cur.cuda_stream))

    def _tensor(self, block):
        dev = torch.device("cuda", self.device_index)
        if self._wrap in ("", "cai"):
            try:
                t = torch.as_tensor(block, device=dev)
                if t.data_ptr() == block.ptr and t.device == dev:
                    self._wrap = "cai"
                    return t
            except Exception:
                if self._wrap == "cai":
                    raise
        self._wrap = "dlpack"
        return _dlpack_tensor(block, self.device_index)

    def alloc(self, sizes, n_distinct=2, avoid_mask=0, max_hunt_slabs=-1):
        """Buffers of ``sizes`` bytes as uint8 tensors, plus their kind indices.  The first ``n_distinct``
        land in pairwise different kinds of memory whenever the device has that many to offer; kinds in
        ``avoid_mask`` (bit q = kind q) are left to other requests if possible."""
        n = len(sizes)
        c_sizes = (ctypes.c_int64 * n)(*[int(s) for s in sizes])
        ptrs = (ctypes.c_void_p * n)()
        kinds = (ctypes.c_int32 * n)()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)
        _lib.check(self.lib.prt_arena_alloc(self._h, n, c_sizes, ptrs, kinds, int(n_distinct), int(avoid_mask),
                                            int(max_hunt_slabs),
                                            stream))
        out = []
        for i in range(n):
            rounded = -(-int(sizes[i]) // SLAB_BYTES) * SLAB_BYTES
            block = _Block(self, ptrs[i], rounded, int(kinds[i]))
            with self._blocks_lock:
                self._blocks[ptrs[i]] = weakref.ref(block)
            out.append(self._tensor(block))
        return out, [int(k) for k in kinds]

    def record_stream(self, tensor, stream=None):
        """tell the arena that ``tensor`` (a view into one of its buffers) is used on ``stream`` (default: the
        current one) -- like torch.Tensor.record_stream for the caching allocator: the buffer is not handed to
        its next user before the work queued on that stream so far ... at release time has finished.  No-op for
        tensors that do not live in the arena."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device_index)
        p = tensor.data_ptr()
        # buffers are whole 1-GiB slabs at 1-GiB aligned addresses: the block that holds p starts at a multiple of
        # the slab size at or below it -- a few dictionary look-ups instead of a scan over every live block
        with self._blocks_lock:
            base = p - p % SLAB_BYTES
            for _ in range(MAX_BLOCK_SLABS):
                ref = self._blocks.get(base)
                blk = ref() if ref is not None else None
                if blk is not None:
                    if p < base + blk.nbytes:
                        blk.streams.add(stream)
                        return True
                    return False
                base -= SLAB_BYTES
                if base < 0:
                    break
        return False

    def kind_of(self, tensor):
        kind = ctypes.c_int32(-1)
        rc = self.lib.prt_arena_kind_of(self._h, ctypes.c_void_p(tensor.data_ptr()), ctypes.byref(kind))
        return int(kind.value) if rc == 0 else None

    def trim(self):
        _lib.check(self.lib.prt_arena_trim(self._h))

    def set_budget(self, max_live_slabs):
        """cap on the 1-GiB slabs the arena holds at any time (None / negative: no cap)"""
        _lib.check(self.lib.prt_arena_set_budget(self._h, -1 if max_live_slabs is None else int(max_live_slabs)))

    def stats(self):
        v = (ctypes.c_int64 * 12)()
        r = (ctypes.c_double * 8)()
        _lib.check(self.lib.prt_arena_stats(self._h, v, 12, r, 8))
        return {"kinds_seen": v[0], "probes": v[1], "slabs_created": v[2], "slabs_released": v[3],
                "slabs_free": v[4], "slabs_in_use": v[5], "slabs_cached": v[6], "slab_bytes": v[7],
                "slabs_per_kind": [v[8], v[9], v[10], v[11]], "probe_same_kind_GBps": r[0],
                "probe_cross_kind_GBps": r[1], "probe_ms_total": r[2], "address_space_reserved_GiB": r[3] / 2 ** 30,
                # the window of address space every mapping is carved from (csrc/prt_placed.h, second rule): far away
                # from anything the host allocator hands out
                "va_window_first": int(r[4]), "va_windows": int(r[5]), "va_window_hinted": bool(r[6]),
                "va_window_GiB": r[7] / 2 ** 30,
                "tensor_wrap": self._wrap, "partition_and_note": self.note(),
                "hunt": os.environ.get("PRT_ARENA_HUNT", "bounded (32 slabs / 50 ms per call)")}

    def note(self):
        """'compute partition/memory partition[; note]' (prt_arena_note)"""
        v = self.lib.prt_arena_note(self._h)
        return v.decode() if v else ""
