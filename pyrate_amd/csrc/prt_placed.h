// prt_placed.h -- placement-aware device memory for the path arrays (include/prt.h "arena").
// Included by prt.hip (one translation unit; uses its fail / HIP_TRY / device_guard helpers).
//
// What it is for (measured: benchmarks/vmm_placement_probe.hip, profiles/r02_vmm_placement_probe.txt,
// DESIGN.md section 5 "Placement"): the physical HBM of an MI355X falls into THREE kinds of memory, each
// one third of the capacity in 1-GiB-aligned runs.  72 concurrent write streams (the path-mode march:
// 12 surfaces x 6 rows) run at 5.6-5.7 TB/s when all of them land in one kind and at 7.0 TB/s when
// x_hit lies in one kind and k_out in another -- every one of 6 216 chunk pairs tried falls on one of
// the two values.  Which kind an allocation gets is decided by the driver's physical allocator and is
// not visible in virtual addresses, so hipMalloc'ed output arrays are a lottery (round 1: 62-81 % of
// the HBM peak for the same binary).
//
// The arena takes physical memory itself, in 1-GiB slabs (hipMemCreate), finds every slab's kind with a
// 2-ms write probe against one representative slab per kind, and builds each requested buffer out of
// slabs of ONE kind, mapped to contiguous virtual addresses (hipMemMap); the buffers of one request
// get different kinds.  Slabs and mapped buffers are cached for reuse.
//
// Two rules shape the code.  First: A VIRTUAL ADDRESS IS MAPPED ONCE AND NEVER AGAIN.  With this ROCm stack
// (7.0 / 7.2) a range that is unmapped (hipMemUnmap) and then mapped to other physical memory keeps
// translating to the OLD pages: data written through addresses of their own never showed up through
// the remapped range, and probe stores issued through a re-used range landed in a live path array
// (profiles/r02b_vmm_placement_probe.txt, experiment E4; found by the exact-similarity test).  So every
// mapping -- a slab under test, a representative, a buffer -- gets a freshly reserved range, and ranges
// are not handed back to the runtime (hipMemAddressFree would allow the next reservation to return the
// same addresses).  Address space is the only thing that leaks: 1 GiB of it per slab tested, out of
// the 128 TiB a process has.
//
// Second (round 6): THE ARENA'S ADDRESSES COME FROM A WINDOW OF ITS OWN, FAR AWAY FROM THE HOST ALLOCATOR.  Until round 5
// every mapping reserved its range with hipMemAddressReserve(addr = 0), i.e. wherever the kernel's mmap puts the next
// anonymous mapping -- which is where the host allocator has just given memory back: benchmarks/va_reuse_probe.hip
// (profiles/r06a_va_reuse_probe.txt): after a PAGEABLE device-to-host copy into a malloc'ed array and free(), 6 of 12
// reservations COVERED the freed range.  The runtime pins the pages of a big pageable destination in place for the DMA
// (a user-pointer registration with the GPU driver) and tears that registration down lazily, so a device mapping made
// at those addresses a moment later shares them with a dying host registration -- the one construction under which
// "memory that was mapped a microsecond ago" can lose its translation, and it is exactly what the intermittent
// first-process fault of round 4 had in front of it (a 96-MB `.cpu().numpy()` of an arena-backed array, freed, then the
// next configuration's arena allocation + launch: DESIGN.md section 5).  Now the arena reserves ONE window of address
// space per TiB it uses, at a hinted address (0x2000'0000'0000 + 8 TiB x device: 32 TiB up, tens of TiB away from the
// heap below and from the mmap area near 0x7f00'0000'0000 above; the runtime honours the hint, probe H) and carves
// every range out of it, front to back, never twice.  Whatever a caller copies to pageable memory and frees, a host
// address and an arena address cannot coincide.  PRT_ARENA_VA_WINDOW_GIB (default 1024) / PRT_ARENA_VA_BASE (hex) /
// PRT_ARENA_VA_WINDOW=off (the round-5 form, for experiments).
#pragma once
#include <mutex>
#include <vector>
#include <algorithm>

#define PRT_SLAB_BYTES ((size_t)1 << 30)
#define PRT_ARENA_PROBE_ROWS 72

struct prt_probe_rows {
    double *p[PRT_ARENA_PROBE_ROWS];
};

// the store structure of k_trace_iso's path mode without its arithmetic: every lane writes 16 B to
// each of the 72 rows
__global__ __launch_bounds__(128) void k_arena_probe(prt_probe_rows rows, int64_t n, double v) {
    const int64_t i = ((int64_t)blockIdx.x * 128 + threadIdx.x) * 2;
    if (i >= n) return;
    double2 val = make_double2(v + (double)i, v - (double)i);
    for (int r = 0; r < PRT_ARENA_PROBE_ROWS; ++r) {
        *(double2 *)(rows.p[r] + i) = val;
        val.x += 1.0;
    }
}

struct prt_slab {
    hipMemGenericAllocationHandle_t handle;
    int32_t kind;
};

struct prt_placed_buffer {
    void *va = nullptr;
    size_t bytes = 0;              // mapped size (whole slabs)
    int32_t kind = -1;
    bool in_use = false;
    hipEvent_t released = nullptr;  // recorded when the buffer was handed back: the last work that used it
    std::vector<prt_slab> slabs;
};

struct prt_arena {
    int32_t device = 0;
    std::mutex mu;
    int32_t n_kinds = 0;
    // one slab per kind stays mapped for good and is never handed out: the probe writes into it
    prt_slab rep[PRT_ARENA_MAX_KINDS];
    void *rep_va[PRT_ARENA_MAX_KINDS] = {nullptr, nullptr, nullptr, nullptr};
    int32_t last_kind = 0;         // kind of the previous slab: slabs come in long runs of one kind
    double self_rate = 0.0;        // yardstick of the current hunt: a slab against itself, GB/s
    int64_t va_reserved = 0;       // bytes of address space handed out so far (never returned, see above)
    // the window the ranges are carved from (second rule above)
    bool win_on = true;
    char *win_base = nullptr;      // current window, slab aligned
    size_t win_bytes = 0, win_used = 0;
    size_t win_cfg_bytes = (size_t)1 << 40;
    uintptr_t win_next_hint = 0;   // where the next window is asked for
    uintptr_t win_first = 0;       // base of the first window (statistics)
    int32_t n_windows = 0;
    bool win_all_hinted = true;    // every window landed where it was asked for
    int64_t slab_budget = -1;      // cap on the slabs held at any time (created - released); < 0: none
    int64_t cache_cap = 64;        // cap on the slabs of cached (unused, still mapped) buffers + free slabs
    // A hunt for kinds is BOUNDED per call: at most hunt_slab_cap slabs beyond what the request needs and hunt_ms_cap
    // milliseconds of probing; a request that cannot get its kinds within that settles for fewer (kinds[] says so),
    // and the NEXT request continues the hunt from the slabs this one left behind.  Defaults 32 slabs / 50 ms --
    // a first call costs tens of milliseconds, not the 100-170 ms of an unbounded walk through a long run of one
    // kind; PRT_ARENA_HUNT=full (benchmarks: the steady-state placement on the first allocation) lifts both to
    // 256 slabs / 2 s, PRT_ARENA_HUNT_MS / PRT_ARENA_HUNT_SLABS set them individually.
    double hunt_ms_cap = 50.0;
    int32_t hunt_slab_cap = 32;
    // ... and in the share of the memory free at the time that a hunt may hold while it walks (it holds what it walks
    // through: a slab given back would be handed out again).  Half by default; PRT_ARENA_HUNT=full: nine tenths -- on
    // a freshly booted box the driver hands out one kind after the other, 91-96 GiB each, so the third kind lies up
    // to 190 slabs deep and half of the free memory (143 slabs) never reaches it (seen: the first process of a box
    // settled for two kinds, inputs sharing one with x_hit, 0.80 instead of 0.835 of the HBM peak).
    double hunt_free_share = 0.5;
    // The three kinds were observed with the device in its default partition modes (compute SPX, memory NPS1).  In
    // another mode the address interleave is a different one and the probe's two-rate picture may not exist: the
    // arena then does not classify at all (every slab is kind 0, no probes) and says so (prt_arena_note).
    bool classify = true;
    bool trace = false;            // PRT_ARENA_TRACE: every probe's rate on stderr (diagnostics)
    char compute_partition[32] = "";
    char memory_partition[32] = "";
    char note[256] = "";
    std::vector<prt_slab> free_slabs;
    std::vector<prt_slab> straddlers;              // slabs that hold memory of two kinds (arena_is_straddler): parked
    std::vector<prt_placed_buffer *> buffers;      // in use and cached
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipMemAllocationProp prop;
    hipMemAccessDesc access;
    // statistics
    int64_t n_probes = 0, n_created = 0, n_released = 0;
    double probe_ms_total = 0.0;
    double bw_same = 0.0, bw_cross = 0.0;          // last probe rates seen, GB/s
};

// a range of addresses nothing was ever mapped to
// (slab-aligned: the runtime ignores the alignment argument beyond 2 MiB, so a slab more is reserved and
// the start rounded up -- a 1-GiB slab at a 1-GiB aligned address can be mapped by the largest page
// table fragments)
static hipError_t arena_new_window(prt_arena *a, size_t at_least) {
    const size_t size = std::max(a->win_cfg_bytes, (at_least + 2 * PRT_SLAB_BYTES - 1) / PRT_SLAB_BYTES * PRT_SLAB_BYTES);
    void *raw = nullptr;
    hipError_t e = hipErrorOutOfMemory;
    bool hinted = false;
    for (int attempt = 0; attempt < 16; ++attempt) {          // (a hint that is taken: the next window-sized step up)
        void *hint = (void *)a->win_next_hint;
        raw = nullptr;
        e = hipMemAddressReserve(&raw, size, PRT_SLAB_BYTES, hint, 0);
        a->win_next_hint += size;
        if (e == hipSuccess && raw == hint) {
            hinted = true;
            break;
        }
        if (e == hipSuccess) (void)hipMemAddressFree(raw, size);      // (nothing was ever mapped there)
        else (void)hipGetLastError();
        raw = nullptr;
    }
    if (!hinted) {          // no hint was honoured: wherever the runtime likes (the round-5 behaviour), and say so
        e = hipMemAddressReserve(&raw, size, PRT_SLAB_BYTES, nullptr, 0);
        if (e != hipSuccess) return e;
        a->win_all_hinted = false;
    }
    char *base = (char *)(((uintptr_t)raw + PRT_SLAB_BYTES - 1) / PRT_SLAB_BYTES * PRT_SLAB_BYTES);
    a->win_base = base;
    a->win_bytes = size - (size_t)(base - (char *)raw);
    a->win_bytes = a->win_bytes / PRT_SLAB_BYTES * PRT_SLAB_BYTES;
    a->win_used = 0;
    if (a->n_windows == 0) a->win_first = (uintptr_t)base;
    a->n_windows += 1;
    if (a->trace) fprintf(stderr, "prt_arena: address window %d at %p, %zu GiB%s\n", a->n_windows, (void *)base,
                          a->win_bytes >> 30, hinted ? "" : " (no hint honoured)");
    return hipSuccess;
}

static hipError_t arena_fresh_va(prt_arena *a, void **va, size_t bytes) {
    if (!a->win_on) {          // the round-5 form: one reservation per mapping, wherever mmap puts it
        void *raw = nullptr;
        hipError_t e = hipMemAddressReserve(&raw, bytes + PRT_SLAB_BYTES, PRT_SLAB_BYTES, nullptr, 0);
        if (e != hipSuccess) return e;
        a->va_reserved += (int64_t)(bytes + PRT_SLAB_BYTES);
        *va = (void *)(((uintptr_t)raw + PRT_SLAB_BYTES - 1) / PRT_SLAB_BYTES * PRT_SLAB_BYTES);
        return hipSuccess;
    }
    bytes = (bytes + PRT_SLAB_BYTES - 1) / PRT_SLAB_BYTES * PRT_SLAB_BYTES;
    if (!a->win_base || a->win_used + bytes > a->win_bytes) {
        hipError_t e = arena_new_window(a, bytes);          // (the rest of the old window is simply left behind)
        if (e != hipSuccess) return e;
    }
    *va = a->win_base + a->win_used;
    a->win_used += bytes;
    a->va_reserved += (int64_t)bytes;
    return hipSuccess;
}

static hipError_t arena_map(prt_arena *a, void *va, const prt_slab &s) {
    hipError_t e = hipMemMap(va, PRT_SLAB_BYTES, 0, s.handle, 0);
    if (e != hipSuccess) return e;
    return hipMemSetAccess(va, PRT_SLAB_BYTES, &a->access, 1);
}

// HARDENING (round 5).  The mapping calls (hipMemMap / hipMemSetAccess / hipMemUnmap / hipMemRelease) are the one part
// of this library that is off the beaten track of the runtime: hipMalloc / hipFree -- what every other program uses --
// never UNMAP while kernels run (hipFree waits for the device first) and never hand memory to a kernel in the same
// microsecond it was mapped through an imported handle.  Three default bench runs of round 4 (of about twenty first
// processes on freshly leased boxes) died of an illegal-address fault right after a buffer had been built from cached
// slabs while a kernel of the process was still running; never reproduced since, so the arena
// simply stays on the beaten track:
//   * arena_quiesce(): the device is idle whenever a mapping is created or destroyed;
//   * arena_touch(): a new mapping is READ AND WRITTEN once per 2-MiB page by a kernel, and waited for, before anybody
//     gets the pointer -- a mapping that is not live faults here, inside the call that made it, with the range named.
// PRT_ARENA_SYNC_MAPS=0 switches both off (the experiment that looks for the fault: tests/campaigns/).
static bool arena_sync_maps() {
    static const bool on = !(getenv("PRT_ARENA_SYNC_MAPS") && atoi(getenv("PRT_ARENA_SYNC_MAPS")) == 0);
    return on;
}

static hipError_t arena_quiesce() {
    return arena_sync_maps() ? hipDeviceSynchronize() : hipSuccess;
}

__global__ __launch_bounds__(256) void k_arena_touch(unsigned long long *base, size_t n_pages, size_t page_words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pages) return;
    unsigned long long *p = base + i * page_words;
    const unsigned long long v = __builtin_nontemporal_load(p);
    __builtin_nontemporal_store(v ^ 0ull, p);
    unsigned long long *q = p + page_words - 1;
    __builtin_nontemporal_store(__builtin_nontemporal_load(q), q);
}

static hipError_t arena_touch(void *va, size_t bytes, hipStream_t st) {
    if (!arena_sync_maps()) return hipSuccess;
    const size_t page = (size_t)2 << 20, n_pages = bytes / page;
    hipLaunchKernelGGL(k_arena_touch, dim3((unsigned)((n_pages + 255) / 256)), dim3(256), 0, st,
                       (unsigned long long *)va, n_pages, page / 8);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        fprintf(stderr, "prt_arena: the first touch of a new mapping [%p, +%zu MiB) failed: %s\n", va, bytes >> 20,
                hipGetErrorString(e));
    return e;
}

// GB/s of the 72-row writer with rows [0,36) in slab memory `lo` and rows [36,72) in `hi`
static hipError_t arena_probe_rate(prt_arena *a, double *lo, double *hi, int64_t row_len, hipStream_t st,
                                   double *gbs) {
    prt_probe_rows rows;
    for (int r = 0; r < 36; ++r) rows.p[r] = lo + (int64_t)r * row_len;
    for (int r = 0; r < 36; ++r) rows.p[36 + r] = hi + (int64_t)r * row_len;
    const dim3 grid((unsigned)((row_len / 2 + 127) / 128)), block(128);
    float best = 1e30f;
    hipLaunchKernelGGL(k_arena_probe, grid, block, 0, st, rows, row_len, 1.0);      // warm-up
    for (int it = 0; it < 3; ++it) {
        hipError_t e = hipEventRecord(a->ev_a, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_arena_probe, grid, block, 0, st, rows, row_len, 1.0);
        if ((e = hipEventRecord(a->ev_b, st)) != hipSuccess) return e;
        if ((e = hipEventSynchronize(a->ev_b)) != hipSuccess) return e;
        float ms = 0.f;
        if ((e = hipEventElapsedTime(&ms, a->ev_a, a->ev_b)) != hipSuccess) return e;
        best = std::min(best, ms);
        a->probe_ms_total += ms;
    }
    a->n_probes += 1;
    *gbs = 72.0 * row_len * 8 / 1e6 / best;
    if (a->trace) fprintf(stderr, "prt_arena probe %p | %p rows of %lld: %.1f GB/s\n", (void *)lo, (void *)hi, (long long)row_len, *gbs);
    return hipGetLastError();
}

// Kind of a slab that is mapped at `mem`: the representative it is SLOW with.  The yardstick is a slab
// against itself (both halves of the streams inside it = one kind by construction; measured once per
// hunt): a pair of different kinds runs 1.23x faster than that, a pair of the same kind at the same
// rate.  The previous slab's kind is tried first -- the driver hands out long runs of one kind.
#define PRT_ARENA_STRADDLER (-2)      // arena_classify: the slab holds memory of two kinds -- not to be used

// Does the slab at `mem` straddle a boundary between two kinds?  (Three of the 284 GiB of a device do.)  Such a slab is
// FAST against itself -- its halves are different kinds --, which makes it useless as the yardstick of a hunt (every
// same-kind pair would pass for slow: no kind would ever be told from another) and, fast against every
// representative as well, it would pass for a kind of its own.  Test, with three probes of the SAME size (so that
// the fixed cost of a launch and the clocks of the moment drop out): 36 rows in each half ("across") against all 72
// rows inside the first half and all 72 inside the second ("within"; the boundary lies in at most one of them).  A
// pure slab gives the same rate three times (measured: within 5 %); a straddler is 1.2 x faster where its two kinds meet.
static hipError_t arena_is_straddler(prt_arena *a, double *mem, hipStream_t st, bool *straddler) {
    *straddler = false;
    if (a->straddlers.size() >= 4) return hipSuccess;      // a device has three: more than that and the test is off
    const int64_t quarter_len = (int64_t)(PRT_SLAB_BYTES / 2 / 72 / 8) / 512 * 512;
    const int64_t half_doubles = (int64_t)(PRT_SLAB_BYTES / 2 / 8);
    double across = 0.0, within[2] = {0.0, 0.0};
    hipError_t e = arena_probe_rate(a, mem, mem + half_doubles, quarter_len, st, &across);
    if (e != hipSuccess) return e;
    for (int h = 0; h < 2; ++h) {
        double *base = mem + h * half_doubles;
        if ((e = arena_probe_rate(a, base, base + 36 * quarter_len, quarter_len, st, &within[h])) != hipSuccess) return e;
    }
    // (the boundary may lie anywhere: in the middle -- `across` is the fast one --, or inside a half -- that half is fast
    //  within itself: any one of the three probes 1.12 x faster than another gives the slab away)
    const double hi = std::max(across, std::max(within[0], within[1])), lo = std::min(across, std::min(within[0], within[1]));
    *straddler = hi > 1.12 * lo;
    if (a->trace) fprintf(stderr, "prt_arena straddler test %p: across the halves %.1f, within %.1f / %.1f GB/s -> %s\n",
                          (void *)mem, across, within[0], within[1], *straddler ? "STRADDLER" : "pure");
    return hipSuccess;
}

static hipError_t arena_classify(prt_arena *a, double *mem, hipStream_t st, int32_t *kind) {
    const int64_t half_len = (int64_t)(PRT_SLAB_BYTES / 72 / 8) / 512 * 512;
    const int64_t full_len = (int64_t)(PRT_SLAB_BYTES / 36 / 8) / 512 * 512;
    hipError_t e;
    if (a->self_rate <= 0.0) {
        // The first measurement of a hunt is often the first GPU work of the process: the clocks are still
        // ramping and the yardstick would come out low -- every later same-kind pair would then look "fast"
        // and be taken for a new kind (seen: 4 "kinds", both path arrays in one of them, 68 % instead of
        // 78 % on the asphere config).  So: wake the device up, then measure until two readings agree.
        double r = 0.0, prev = 0.0;
        for (int it = 0; it < 8; ++it) {
            if ((e = arena_probe_rate(a, mem, mem + 36 * half_len, half_len, st, &r)) != hipSuccess) return e;
            if (it > 0 && fabs(r - prev) <= 0.02 * r) break;
            prev = r;
        }
        bool straddler = false;
        if ((e = arena_is_straddler(a, mem, st, &straddler)) != hipSuccess) return e;
        if (straddler) {              // no yardstick from this one: the next slab of the hunt gives it
            *kind = PRT_ARENA_STRADDLER;
            return hipSuccess;
        }
        a->self_rate = r > prev ? r : prev;
        a->bw_same = a->self_rate;
    }
    double pair[PRT_ARENA_MAX_KINDS] = {0.0, 0.0, 0.0, 0.0};
    for (int pass = 0; pass < 2; ++pass) {
        for (int t = 0; t < a->n_kinds; ++t) {
            const int q = (t == 0) ? a->last_kind : (t <= a->last_kind ? t - 1 : t);
            if (pass == 0 && (e = arena_probe_rate(a, (double *)a->rep_va[q], mem, full_len, st, &pair[q])) != hipSuccess)
                return e;
            if (pair[q] < 1.10 * a->self_rate) {
                a->bw_same = pair[q];
                *kind = q;
                return hipSuccess;
            }
            a->bw_cross = pair[q];
        }
        if (pass == 1 || a->n_kinds == 0) break;
        // "fast with every known kind" would make this slab a NEW kind.  Before believing that, take the
        // yardstick again on this very slab (a slab against itself is one kind by construction): if it reads
        // higher than the one of the hunt, that one was taken too early, and the pairs are judged again.
        double self_now = 0.0;
        if ((e = arena_probe_rate(a, mem, mem + 36 * half_len, half_len, st, &self_now)) != hipSuccess) return e;
        if (self_now > 1.03 * a->self_rate) {
            // ... unless this slab reads high against itself because it straddles two kinds (that is also why it is
            // fast with every representative): it is neither a new kind nor a reason to move the yardstick
            bool straddler = false;
            if ((e = arena_is_straddler(a, mem, st, &straddler)) != hipSuccess) return e;
            if (straddler) {
                *kind = PRT_ARENA_STRADDLER;
                return hipSuccess;
            }
        }
        if (self_now <= 1.03 * a->self_rate) break;
        a->self_rate = self_now;
        a->bw_same = self_now;
    }
    *kind = a->n_kinds;        // fast with every known kind: a new one
    return hipSuccess;
}

// One more slab from the driver, classified.  It is mapped for the test at addresses of its own; a
// slab of a new kind stays there as that kind's representative (became_rep) and is not available for
// buffers, any other is unmapped again (and will be mapped elsewhere, never there).
static hipError_t arena_new_slab(prt_arena *a, hipStream_t st, prt_slab *out, bool *became_rep) {
    *became_rep = false;
    size_t free_b = 0, total_b = 0;
    hipError_t e = hipMemGetInfo(&free_b, &total_b);
    if (e != hipSuccess) return e;
    // Leave the last GiBs alone: a hunt for a far-away kind holds everything it has walked through, and a
    // device with (almost) nothing free is where the runtime's own allocations start to fail -- once, with
    // ~3 GiB left, a HIP call aborted the process instead of returning an error.
    if (free_b < 12 * PRT_SLAB_BYTES) return hipErrorOutOfMemory;
    if (a->slab_budget >= 0 && a->n_created - a->n_released >= a->slab_budget) return hipErrorOutOfMemory;
    prt_slab s;
    s.kind = -1;
    if ((e = hipMemCreate(&s.handle, PRT_SLAB_BYTES, &a->prop, 0)) != hipSuccess) return e;
    a->n_created += 1;
    void *va = nullptr;
    if ((e = arena_quiesce()) != hipSuccess || (e = arena_fresh_va(a, &va, PRT_SLAB_BYTES)) != hipSuccess ||
        (e = arena_map(a, va, s)) != hipSuccess) {
        (void)hipMemRelease(s.handle);
        a->n_released += 1;      // (keeps created - released = slabs held: what the budget counts)
        return e;
    }
    int32_t kind = -1;
    if (!a->classify) {          // unknown interleave (partition mode): one kind, no probes
        kind = 0;                // (the first slab becomes kind 0's representative below, like a classified one)
        e = hipSuccess;
    } else {
        e = arena_classify(a, (double *)va, st, &kind);
    }
    (void)arena_quiesce();         // (the probes are done -- they were waited for --; so is everything else now)
    if (e != hipSuccess) {
        (void)hipMemUnmap(va, PRT_SLAB_BYTES);
        (void)hipMemRelease(s.handle);
        a->n_released += 1;
        return e;
    }
    if (kind == PRT_ARENA_STRADDLER) {
        // two kinds inside one slab: kept out of every buffer -- parked (mapped nowhere, handle held, so that the
        // driver does not hand it out again) until the arena is destroyed; a device has three of these
        (void)hipMemUnmap(va, PRT_SLAB_BYTES);
        s.kind = PRT_ARENA_STRADDLER;
        a->straddlers.push_back(s);
        *out = s;
        return hipSuccess;
    }
    if (kind == a->n_kinds && a->n_kinds < PRT_ARENA_MAX_KINDS) {
        s.kind = kind;
        a->rep[kind] = s;
        a->rep_va[kind] = va;
        a->n_kinds += 1;
        *became_rep = true;
    } else {
        (void)hipMemUnmap(va, PRT_SLAB_BYTES);
        s.kind = std::min(kind, PRT_ARENA_MAX_KINDS - 1);
    }
    a->last_kind = s.kind;
    *out = s;
    return hipSuccess;
}

static void arena_release_slab(prt_arena *a, const prt_slab &s) {
    (void)hipMemRelease(s.handle);
    a->n_released += 1;
}

static void arena_unmap_buffer(prt_arena *a, prt_placed_buffer *b, bool keep_slabs) {
    if (b->released) {
        (void)hipEventSynchronize(b->released);
        (void)hipEventDestroy(b->released);
        b->released = nullptr;
    }
    if (b->va) {
        (void)arena_quiesce();
        (void)hipMemUnmap(b->va, b->bytes);      // the addresses are retired, not handed back
    }
    for (const prt_slab &s : b->slabs) {
        if (keep_slabs) a->free_slabs.push_back(s);
        else arena_release_slab(a, s);
    }
    b->slabs.clear();
    b->va = nullptr;
}

static int64_t arena_count_free(const prt_arena *a, int32_t kind) {
    int64_t n = 0;
    for (const prt_slab &s : a->free_slabs) n += (s.kind == kind);
    return n;
}

// cached (mapped, unused) buffer of exactly this size and kind
static prt_placed_buffer *arena_cached_buffer(prt_arena *a, size_t n_slabs, int32_t kind) {
    for (prt_placed_buffer *b : a->buffers)
        if (!b->in_use && b->kind == kind && b->slabs.size() == n_slabs) return b;
    return nullptr;
}

static int64_t arena_count_cached(const prt_arena *a, size_t n_slabs, int32_t kind) {
    int64_t n = 0;
    for (const prt_placed_buffer *b : a->buffers)
        n += (!b->in_use && b->kind == kind && b->slabs.size() == n_slabs);
    return n;
}

static hipError_t arena_build_buffer(prt_arena *a, size_t n_slabs, int32_t kind, hipStream_t st, prt_placed_buffer **out) {
    prt_placed_buffer *b = new (std::nothrow) prt_placed_buffer;
    if (!b) return hipErrorOutOfMemory;
    b->bytes = n_slabs * PRT_SLAB_BYTES;
    b->kind = kind;
    hipError_t e = arena_quiesce();
    if (e == hipSuccess) e = arena_fresh_va(a, &b->va, b->bytes);
    if (e != hipSuccess) {
        delete b;
        return e;
    }
    for (size_t i = 0; i < a->free_slabs.size() && b->slabs.size() < n_slabs;) {
        if (a->free_slabs[i].kind == kind) {
            b->slabs.push_back(a->free_slabs[i]);
            a->free_slabs.erase(a->free_slabs.begin() + i);
        } else {
            ++i;
        }
    }
    for (size_t i = 0; i < b->slabs.size() && e == hipSuccess; ++i)
        e = hipMemMap((char *)b->va + i * PRT_SLAB_BYTES, PRT_SLAB_BYTES, 0, b->slabs[i].handle, 0);
    if (e == hipSuccess) e = hipMemSetAccess(b->va, b->bytes, &a->access, 1);
    if (e == hipSuccess && b->slabs.size() == n_slabs) e = arena_touch(b->va, b->bytes, st);
    if (e != hipSuccess || b->slabs.size() != n_slabs) {
        arena_unmap_buffer(a, b, true);
        delete b;
        return e != hipSuccess ? e : hipErrorOutOfMemory;
    }
    a->buffers.push_back(b);
    *out = b;
    return hipSuccess;
}

extern "C" {

int32_t prt_arena_create(int32_t device, prt_arena_t **out) {
    if (!out) return fail(PRT_ERR_INVALID_ARG, "prt_arena_create: null out");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PRT_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= count) return fail(PRT_ERR_INVALID_ARG, "prt_arena_create: bad device");
    PRT_ON_DEVICE(device);
    prt_arena *a = new (std::nothrow) prt_arena;
    if (!a) return fail(PRT_ERR_NOMEM, "prt_arena_create: host memory");
    a->device = device;
    memset(&a->prop, 0, sizeof a->prop);
    a->prop.type = hipMemAllocationTypePinned;
    a->prop.location.type = hipMemLocationTypeDevice;
    a->prop.location.id = device;
    memset(&a->access, 0, sizeof a->access);
    a->access.location = a->prop.location;
    a->access.flags = hipMemAccessFlagsProtReadWrite;
    hipError_t e = hipEventCreate(&a->ev_a);
    if (e == hipSuccess) e = hipEventCreate(&a->ev_b);
    if (e != hipSuccess) {
        if (a->ev_a) (void)hipEventDestroy(a->ev_a);
        if (a->ev_b) (void)hipEventDestroy(a->ev_b);
        delete a;
        return fail(PRT_ERR_DEVICE, "prt_arena_create", e);
    }
    // Defaults that keep the arena a good neighbour of the torch allocator and of other processes on the device
    // (ADVICE round 2): never more than three quarters of the device's memory, at most 64 GiB of cached (unused)
    // memory, a hunt bounded in time.  PRT_ARENA_BUDGET_GIB / PRT_ARENA_CACHE_GIB / PRT_ARENA_HUNT_MS override.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0)
        a->slab_budget = (int64_t)(total_b / PRT_SLAB_BYTES) * 3 / 4;
    if (const char *v = getenv("PRT_ARENA_HUNT"))      // (a full hunt may have to hold two whole kinds: 190 of 268 slabs)
        if (strcmp(v, "full") == 0 && total_b > 0) a->slab_budget = (int64_t)(total_b / PRT_SLAB_BYTES) * 9 / 10;
    if (const char *v = getenv("PRT_ARENA_BUDGET_GIB")) a->slab_budget = atoll(v);
    if (const char *v = getenv("PRT_ARENA_CACHE_GIB")) a->cache_cap = atoll(v);
    if (const char *v = getenv("PRT_ARENA_HUNT")) {
        if (strcmp(v, "full") == 0) {
            a->hunt_ms_cap = 2000.0;
            a->hunt_slab_cap = 256;
            a->hunt_free_share = 0.9;
        }
    }
    a->trace = getenv("PRT_ARENA_TRACE") != nullptr;
    a->win_next_hint = (uintptr_t)0x200000000000ull + (uintptr_t)device * ((uintptr_t)8 << 40);
    if (const char *v = getenv("PRT_ARENA_VA_BASE")) a->win_next_hint = (uintptr_t)strtoull(v, nullptr, 16) / PRT_SLAB_BYTES * PRT_SLAB_BYTES;
    if (const char *v = getenv("PRT_ARENA_VA_WINDOW_GIB")) a->win_cfg_bytes = (size_t)std::max(4ll, atoll(v)) << 30;
    if (const char *v = getenv("PRT_ARENA_VA_WINDOW")) a->win_on = !(strcmp(v, "off") == 0 || strcmp(v, "0") == 0);
    if (const char *v = getenv("PRT_ARENA_HUNT_MS")) a->hunt_ms_cap = atof(v);
    if (const char *v = getenv("PRT_ARENA_HUNT_SLABS")) a->hunt_slab_cap = atoi(v);
    // partition modes of the device, from sysfs (amdgpu: current_compute_partition / current_memory_partition beside
    // the PCI device); unknown (no such files: an older driver, a container without sysfs) counts as the default
    {
        char bus[64] = "";
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) == hipSuccess) {
            for (char *c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
            auto read_mode = [&](const char *file, char *dst, size_t cap) {
                char path[256];
                snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/%s", bus, file);
                FILE *f = fopen(path, "r");
                if (!f) return;
                if (fgets(dst, (int)cap, f)) {
                    size_t n = strlen(dst);
                    while (n > 0 && (dst[n - 1] == '\n' || dst[n - 1] == ' ')) dst[--n] = 0;
                }
                fclose(f);
            };
            read_mode("current_compute_partition", a->compute_partition, sizeof a->compute_partition);
            read_mode("current_memory_partition", a->memory_partition, sizeof a->memory_partition);
        }
        const bool spx = a->compute_partition[0] == 0 || strcmp(a->compute_partition, "SPX") == 0;
        const bool nps1 = a->memory_partition[0] == 0 || strcmp(a->memory_partition, "NPS1") == 0;
        if (!(spx && nps1) && !getenv("PRT_ARENA_CLASSIFY")) {
            a->classify = false;
            snprintf(a->note, sizeof a->note,
                     "partition mode %s / %s: the three kinds of HBM were characterised in SPX / NPS1 only -- slabs are "
                     "not classified (one kind, no probes; PRT_ARENA_CLASSIFY=1 forces the probes)",
                     a->compute_partition[0] ? a->compute_partition : "?", a->memory_partition[0] ? a->memory_partition : "?");
        }
    }
    *out = a;
    return PRT_OK;
}

// cached (unused) buffers beyond the cache cap go back to the driver, oldest first (caller holds the lock)
static void arena_shrink_cache(prt_arena *a) {
    if (a->cache_cap < 0) return;
    int64_t cached = (int64_t)a->free_slabs.size();
    for (const prt_placed_buffer *b : a->buffers)
        if (!b->in_use) cached += (int64_t)b->slabs.size();
    for (size_t i = 0; i < a->buffers.size() && cached > a->cache_cap;) {
        prt_placed_buffer *b = a->buffers[i];
        if (!b->in_use) {
            cached -= (int64_t)b->slabs.size();
            arena_unmap_buffer(a, b, false);     // waits for the buffer's last user (its release event)
            delete b;
            a->buffers.erase(a->buffers.begin() + i);
        } else {
            ++i;
        }
    }
    while (cached > a->cache_cap && !a->free_slabs.empty()) {
        arena_release_slab(a, a->free_slabs.back());
        a->free_slabs.pop_back();
        --cached;
    }
}

int32_t prt_arena_trim(prt_arena_t *a) {
    if (!a) return fail(PRT_ERR_INVALID_ARG, "prt_arena_trim: null arena");
    PRT_ON_DEVICE(a->device);
    std::lock_guard<std::mutex> lock(a->mu);
    HIP_TRY(hipDeviceSynchronize());
    for (size_t i = 0; i < a->buffers.size();) {
        prt_placed_buffer *b = a->buffers[i];
        if (!b->in_use) {
            arena_unmap_buffer(a, b, false);
            delete b;
            a->buffers.erase(a->buffers.begin() + i);
        } else {
            ++i;
        }
    }
    for (const prt_slab &s : a->free_slabs) arena_release_slab(a, s);
    a->free_slabs.clear();
    for (const prt_slab &s : a->straddlers) arena_release_slab(a, s);      // (a later hunt recognises them again)
    a->straddlers.clear();
    return PRT_OK;
}

int32_t prt_arena_set_budget(prt_arena_t *a, int64_t max_live_slabs) {
    if (!a) return fail(PRT_ERR_INVALID_ARG, "prt_arena_set_budget: null arena");
    std::lock_guard<std::mutex> lock(a->mu);
    a->slab_budget = max_live_slabs;
    return PRT_OK;
}

int32_t prt_arena_destroy(prt_arena_t *a) {
    if (!a) return PRT_OK;
    {
        PRT_ON_DEVICE(a->device);
        (void)hipDeviceSynchronize();
        for (prt_placed_buffer *b : a->buffers) {
            arena_unmap_buffer(a, b, false);
            delete b;
        }
        a->buffers.clear();
        for (const prt_slab &s : a->free_slabs) arena_release_slab(a, s);
        for (const prt_slab &s : a->straddlers) arena_release_slab(a, s);
        for (int q = 0; q < a->n_kinds; ++q) {
            (void)hipMemUnmap(a->rep_va[q], PRT_SLAB_BYTES);
            arena_release_slab(a, a->rep[q]);
        }
        (void)hipEventDestroy(a->ev_a);
        (void)hipEventDestroy(a->ev_b);
    }
    delete a;
    return PRT_OK;
}

int32_t prt_arena_alloc(prt_arena_t *a, int32_t n_parts, const int64_t *bytes, void **ptrs, int32_t *kinds,
                        int32_t n_distinct, int32_t avoid_mask, int32_t max_hunt_slabs, void *stream) {
    if (!a || n_parts <= 0 || n_parts > 8 || !bytes || !ptrs || n_distinct > PRT_ARENA_MAX_KINDS || avoid_mask < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_arena_alloc: bad arguments");
    if (n_distinct < 0) n_distinct = 2;
    if (n_distinct > n_parts) n_distinct = n_parts;
    for (int i = 0; i < n_parts; ++i) {
        if (bytes[i] <= 0) return fail(PRT_ERR_INVALID_ARG, "prt_arena_alloc: part sizes must be positive");
        ptrs[i] = nullptr;
    }
    PRT_ON_DEVICE(a->device);
    std::lock_guard<std::mutex> lock(a->mu);
    if (max_hunt_slabs < 0) {
        // a kind is 96 GiB, so the third one can be 192 slabs of the other two away -- but a hunt holds what it walks
        // through: by default it may take at most half of what is free right now (hunt_free_share; never more than
        // hunt_slab_cap slabs)
        size_t free_b = 0, total_b = 0;
        max_hunt_slabs = a->hunt_slab_cap;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            max_hunt_slabs = (int32_t)std::min<size_t>((size_t)std::max(0, a->hunt_slab_cap),
                                                       (size_t)((double)(free_b / PRT_SLAB_BYTES) * a->hunt_free_share));
    }
    hipStream_t st = (hipStream_t)stream;
    size_t need[8];
    for (int i = 0; i < n_parts; ++i) need[i] = ((size_t)bytes[i] + PRT_SLAB_BYTES - 1) / PRT_SLAB_BYTES;

    // Assignment of kinds to parts: the first n_distinct parts (x_hit, k_out -- the two halves of the
    // write streams --, then the inputs) must differ pairwise; further parts take a kind nobody uses
    // yet if one is at hand, otherwise whatever has room.  A part is served by a cached buffer of its size and kind, or by free slabs.
    // Greedy over the parts in order; returns false if a part cannot be served.
    int32_t chosen[8];
    // Kinds in avoid_mask (bit q = kind q) are left to other users -- the input arrays stay out of the
    // kinds the write streams use -- unless `use_avoided` gives up on that.
    auto assign = [&](int strict_parts, bool use_avoided) {
        int64_t slabs_taken[PRT_ARENA_MAX_KINDS] = {0, 0, 0, 0};
        bool from_cache[8];
        bool used[PRT_ARENA_MAX_KINDS] = {false, false, false, false};
        for (int i = 0; i < n_parts; ++i) {
            int best = -1;
            bool best_cached = false;
            for (int pass = 0; pass < 2 && best < 0; ++pass) {     // pass 0: kinds no earlier part uses
                if (pass == 1 && i < strict_parts) break;          // these parts must differ pairwise
                for (int q = 0; q < a->n_kinds && best < 0; ++q) {
                    if (pass == 0 && used[q]) continue;
                    if (!use_avoided && ((avoid_mask >> q) & 1)) continue;
                    int64_t cached_taken = 0;
                    for (int j = 0; j < i; ++j)
                        cached_taken += (chosen[j] == q && need[j] == need[i] && from_cache[j]);
                    if (arena_count_cached(a, need[i], q) > cached_taken) {
                        best = q;
                        best_cached = true;
                    } else if (arena_count_free(a, q) - slabs_taken[q] >= (int64_t)need[i]) {
                        best = q;
                    }
                }
            }
            if (best < 0) return false;
            chosen[i] = best;
            from_cache[i] = best_cached;
            used[best] = true;
            if (!best_cached) slabs_taken[best] += (int64_t)need[i];
        }
        return true;
    };

    // hunt: take slabs from the driver until the strict assignment works (or the hunt budget is spent)
    a->self_rate = 0.0;            // re-measured by the first slab this call takes from the driver
    int32_t hunted = 0;
    hipError_t hunt_err = hipSuccess;
    const double probe_ms_start = a->probe_ms_total;
    while (!assign(n_distinct, false)) {
        size_t total_need = 0;
        for (int i = 0; i < n_parts; ++i) total_need += need[i];
        if (hunted >= (int32_t)total_need + max_hunt_slabs) break;
        // bounded in time as well: once the request could be served with fewer kinds, a hunt that has probed for
        // hunt_ms_cap stops and settles for them (kinds[] tells the caller)
        if (a->hunt_ms_cap >= 0 && a->probe_ms_total - probe_ms_start > a->hunt_ms_cap && assign(0, true)) break;
        prt_slab s;
        bool became_rep = false;
        hunt_err = arena_new_slab(a, st, &s, &became_rep);
        if (hunt_err == hipErrorOutOfMemory) {
            // The driver has nothing left.  Buffers of other sizes that sit in the cache hold slabs this
            // request can use: take them apart (their addresses are retired, the slabs classified already).
            bool recycled = false;
            for (size_t i = 0; i < a->buffers.size();) {
                prt_placed_buffer *b = a->buffers[i];
                if (!b->in_use) {
                    arena_unmap_buffer(a, b, true);
                    delete b;
                    a->buffers.erase(a->buffers.begin() + i);
                    recycled = true;
                } else {
                    ++i;
                }
            }
            if (recycled) {
                hunt_err = hipSuccess;
                continue;
            }
        }
        if (hunt_err != hipSuccess) break;
        if (s.kind == PRT_ARENA_STRADDLER) continue;      // parked by arena_new_slab; (bounded: a device has three)
        if (!became_rep) a->free_slabs.push_back(s);
        ++hunted;
    }
    bool ok = false;                          // not enough memory of that many kinds: fewer it is
    for (int strict = n_distinct; strict >= 0 && !ok; --strict) ok = assign(strict, false);
    for (int strict = n_distinct; strict >= 0 && !ok; --strict) ok = assign(strict, true);
    if (!ok) {
        if (hunt_err != hipSuccess && hunt_err != hipErrorOutOfMemory)
            return fail(PRT_ERR_DEVICE, "prt_arena_alloc: taking memory from the driver", hunt_err);
        return fail(PRT_ERR_NOMEM, "prt_arena_alloc: not enough free device memory");
    }
    prt_placed_buffer *got[8];
    for (int i = 0; i < n_parts; ++i) {
        prt_placed_buffer *b = arena_cached_buffer(a, need[i], chosen[i]);
        if (!b) {
            hipError_t e = arena_build_buffer(a, need[i], chosen[i], st, &b);
            if (e != hipSuccess) {
                for (int j = 0; j < i; ++j) got[j]->in_use = false;
                return fail(e == hipErrorOutOfMemory ? PRT_ERR_NOMEM : PRT_ERR_DEVICE,
                            "prt_arena_alloc: mapping a buffer", e);
            }
        }
        if (b->released) HIP_TRY(hipStreamWaitEvent(st, b->released, 0));   // behind its previous user
        b->in_use = true;
        got[i] = b;
    }
    for (int i = 0; i < n_parts; ++i) {
        ptrs[i] = got[i]->va;
        if (kinds) kinds[i] = got[i]->kind;
    }
    // slabs the hunt took beyond what was needed go back to the driver right away (they are all of
    // kinds that are plentiful); slabs of buffers that were used once stay cached until prt_arena_trim
    if (hunted > 0) {
        // ... except a few per kind (8 GiB): the next request, of another size, then starts with both
        // kinds at hand instead of walking through the same run of one kind again
        int64_t kept[PRT_ARENA_MAX_KINDS] = {0, 0, 0, 0};
        std::vector<prt_slab> keep;
        for (const prt_slab &s : a->free_slabs) {
            if (kept[s.kind] < 8) {
                kept[s.kind] += 1;
                keep.push_back(s);
            } else {
                arena_release_slab(a, s);
            }
        }
        a->free_slabs.swap(keep);
    }
    return PRT_OK;
}

int32_t prt_arena_free(prt_arena_t *a, void *ptr, void *stream) {
    if (!a || !ptr) return fail(PRT_ERR_INVALID_ARG, "prt_arena_free: null argument");
    PRT_ON_DEVICE(a->device);
    std::lock_guard<std::mutex> lock(a->mu);
    for (prt_placed_buffer *b : a->buffers)
        if (b->va == ptr && b->in_use) {
            // No host wait: an event on the caller's stream marks the last work that may use the
            // buffer; whoever gets the buffer next is ordered behind it (prt_arena_alloc), and it is
            // waited for before the memory is unmapped.
            if (!b->released) HIP_TRY(hipEventCreateWithFlags(&b->released, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(b->released, (hipStream_t)stream));
            b->in_use = false;        // stays mapped: the next request of this size and kind takes it as it is
            arena_shrink_cache(a);     // ... unless the cache is over its cap
            return PRT_OK;
        }
    return fail(PRT_ERR_INVALID_ARG, "prt_arena_free: not a buffer of this arena");
}

int32_t prt_arena_kind_of(prt_arena_t *a, const void *ptr, int32_t *kind) {
    if (!a || !ptr || !kind) return fail(PRT_ERR_INVALID_ARG, "prt_arena_kind_of: null argument");
    std::lock_guard<std::mutex> lock(a->mu);
    for (prt_placed_buffer *b : a->buffers)
        if ((const char *)ptr >= (const char *)b->va && (const char *)ptr < (const char *)b->va + b->bytes) {
            *kind = b->kind;
            return PRT_OK;
        }
    return fail(PRT_ERR_INVALID_ARG, "prt_arena_kind_of: not inside a buffer of this arena");
}

int32_t prt_arena_stats(prt_arena_t *a, int64_t *out, int32_t n_out, double *rates, int32_t n_rates) {
    if (!a || (!out && n_out > 0)) return fail(PRT_ERR_INVALID_ARG, "prt_arena_stats: null argument");
    std::lock_guard<std::mutex> lock(a->mu);
    int64_t v[12] = {0};
    v[0] = a->n_kinds;
    v[1] = a->n_probes;
    v[2] = a->n_created;
    v[3] = a->n_released;
    v[4] = (int64_t)a->free_slabs.size();
    for (const prt_placed_buffer *b : a->buffers) {
        (b->in_use ? v[5] : v[6]) += (int64_t)b->slabs.size();
        if (b->kind >= 0 && b->kind < PRT_ARENA_MAX_KINDS) v[8 + b->kind] += (int64_t)b->slabs.size();
    }
    v[7] = (int64_t)(PRT_SLAB_BYTES);
    for (int i = 0; i < n_out && i < 12; ++i) out[i] = v[i];
    // (addresses fit a double exactly: 47 bits)
    double r[8] = {a->bw_same, a->bw_cross, a->probe_ms_total, (double)a->va_reserved, (double)a->win_first,
                   (double)a->n_windows, (a->win_on && a->win_all_hinted) ? 1.0 : 0.0, (double)a->win_cfg_bytes};
    for (int i = 0; i < n_rates && i < 8; ++i) rates[i] = r[i];
    return PRT_OK;
}

// "compute partition / memory partition[; note]" -- what the arena read from sysfs when it was created, and, if it
// does not classify slabs in this mode, why.  The string lives in a per-thread buffer: valid until the calling thread's
// next prt_arena_note call (copy it to keep it).
const char *prt_arena_note(prt_arena_t *a) {
    if (!a) return "";
    static thread_local char buf[400];
    snprintf(buf, sizeof buf, "%s/%s%s%s", a->compute_partition[0] ? a->compute_partition : "unknown",
             a->memory_partition[0] ? a->memory_partition : "unknown", a->note[0] ? "; " : "", a->note);
    return buf;
}

}  // extern "C"
