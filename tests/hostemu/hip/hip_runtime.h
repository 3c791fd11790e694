// TEST INFRASTRUCTURE -- not part of the product, never shipped, never loaded by pyrate_amd.
//
// A stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED sources of libprt (pyrate_amd/csrc/prt.hip and its
// headers) be compiled by a host C++ compiler (clang++ for x86-64, with AddressSanitizer / UBSan) into
// tests/hostemu/_build/libprt_hostemu.so.  The build container has no GPU and GPU sanitizers are not available on
// the pool: this is the "sanitizers on the CPU build" of the kernels.  What it gives the `-m "not gpu"` suite:
//   * every index computation of every kernel (row pitches, two-rays-per-thread tails, the concatenated crystal
//     layout, the walk program, partial last blocks) runs under ASan on exact-size NumPy arrays;
//   * the arithmetic of the kernels -- the same C++ expressions, the same launch-site logic of prt.hip (which
//     instantiation, which grid, which layout) -- is compared with the oracle and the golden vectors before a GPU
//     has seen a change.
// What it is NOT: a model of the hardware.  Threads of a block run as cooperative fibres on one OS thread (a block at a
// time, blocks in order; several host threads may launch at once, each with its own fibres), `__syncthreads` is a
// rendezvous of the block's live threads, wave votes / shuffles are rendezvous of the lanes that are active at one call site,
// v_rcp_f64 / v_rsq_f64 are exact divisions, memory is malloc'ed.  Results agree with the GPU's to rounding, not to
// the bit.  "Device pointers" are host pointers.  The product has no CPU path: pyrate_amd/_lib.py loads
// pyrate_amd/csrc/libprt.so (the gfx950 build) or raises -- and refuses THIS build by its ABI version (hostemu_prt.cpp).
// Sanitizers: the fibres are announced to AddressSanitizer and ThreadSanitizer (start / finish_switch_fiber, __tsan_*_fiber).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <chrono>
#include <functional>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HOSTEMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#if __has_feature(thread_sanitizer)
#define HOSTEMU_TSAN 1
#include <sanitizer/tsan_interface.h>
#endif
#endif

// ---- language ----------------------------------------------------------------------------------------------------
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// block-static and dynamic LDS arrays: one block runs at a time on one OS thread, so one instance per array is the
// block's instance (`thread_local` is valid both on a block-scope definition and after `extern`)
#define __shared__ thread_local

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
// (thread_local: several host threads may launch at once -- each runs its own blocks on its own fibres)
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
#define warpSize 64

// ---- the fibres of a block ------------------------------------------------------------------------------------------
// A block's threads are fibres on one OS thread, resumed round robin.  __syncthreads is a rendezvous of the block's live
// threads.  Wave votes and shuffles are rendezvous of lanes of ONE wave at ONE textual call site: the hardware evaluates
// them over the lanes that are active there (the exec mask), e.g. the `__all(done)` of the Newton loop inside
// `if (i < N) { ... }` over the lanes that have a ray while the others wait at the reduction behind the branch.  The
// emulation: a lane that reaches such a call waits; when every live lane of its wave is waiting somewhere (at call
// sites of this kind or at the block barrier), ONE group -- the lanes waiting at one site -- is let through as the
// active set; votes before shuffles (the kernels vote inside divergent regions and shuffle only where a wave has
// reconverged), lower site number first.  A block in which nobody can be let through is reported, not left spinning.
namespace hostemu {
constexpr int WAVE = 64;
constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_BYTES = 512 * 1024;
enum { KIND_VOTE = 0, KIND_SHUFFLE = 1 };

struct fibre {
    ucontext_t ctx;
    char *stack = nullptr;
    bool done = false;
    dim3 tid;
    // waiting at a wave call site
    bool waiting = false, released = false;
    int site = 0, kind = 0;
    uint64_t group = 0;        // the lanes let through together with this one
    bool at_block_barrier = false;
#ifdef HOSTEMU_ASAN
    void *fake_stack = nullptr;
#endif
#ifdef HOSTEMU_TSAN
    void *tsan_fiber = nullptr;
#endif
};

struct block_state {
    std::vector<fibre> fibres;
    ucontext_t sched;
    int current = -1;
    int live = 0;
    int arrived = 0;           // block barrier
    uint64_t generation = 0;
    int wave_live[MAX_THREADS / WAVE];
    double wave_buf[MAX_THREADS / WAVE][WAVE];
    uint64_t wave_bits[MAX_THREADS / WAVE];
    uint64_t progress = 0;     // releases + finished threads: the deadlock check of run_block
    const std::function<void()> *body = nullptr;
#ifdef HOSTEMU_ASAN
    void *sched_fake = nullptr;
    const void *sched_bottom = nullptr;
    size_t sched_size = 0;
#endif
#ifdef HOSTEMU_TSAN
    void *sched_tsan = nullptr;
#endif
};
inline thread_local block_state *g_block = nullptr;
inline thread_local std::vector<char *> g_stacks;   // reused between blocks

inline void switch_to_sched(bool dying) {
    block_state &b = *g_block;
    fibre &f = b.fibres[b.current];
#ifdef HOSTEMU_ASAN
    __sanitizer_start_switch_fiber(dying ? nullptr : &f.fake_stack, b.sched_bottom, b.sched_size);
#endif
#ifdef HOSTEMU_TSAN
    __tsan_switch_to_fiber(b.sched_tsan, 0);
#endif
    swapcontext(&f.ctx, &b.sched);
#ifdef HOSTEMU_ASAN
    __sanitizer_finish_switch_fiber(f.fake_stack, &b.sched_bottom, &b.sched_size);
#endif
    (void)dying;
}

inline void yield() { switch_to_sched(false); }

// every live lane of wave w waiting?  then let the lanes of one call site through
inline void wave_decide(int w) {
    block_state &b = *g_block;
    const int n = (int)b.fibres.size();
    int blocked = 0;
    bool have = false;
    int best_kind = 0, best_site = 0;
    for (int lane = 0; lane < WAVE; ++lane) {
        const int t = w * WAVE + lane;
        if (t >= n) break;
        const fibre &f = b.fibres[t];
        if (f.done) continue;
        if (f.at_block_barrier) { blocked += 1; continue; }
        if (f.waiting && !f.released) {
            blocked += 1;
            if (!have || f.kind < best_kind || (f.kind == best_kind && f.site < best_site)) {
                have = true; best_kind = f.kind; best_site = f.site;
            }
        }
    }
    if (!have || blocked < b.wave_live[w]) return;
    uint64_t group = 0;
    for (int lane = 0; lane < WAVE && w * WAVE + lane < n; ++lane) {
        const fibre &f = b.fibres[w * WAVE + lane];
        if (!f.done && f.waiting && !f.released && f.kind == best_kind && f.site == best_site) group |= 1ull << lane;
    }
    for (int lane = 0; lane < WAVE && w * WAVE + lane < n; ++lane)
        if ((group >> lane) & 1) {
            fibre &f = b.fibres[w * WAVE + lane];
            f.released = true; f.waiting = false; f.group = group;
        }
    b.progress += 1;
}

// returns the set of lanes that passed this call site together (the active lanes of the vote / shuffle)
inline uint64_t wave_rendezvous(int site, int kind) {
    block_state &b = *g_block;
    const int t = b.current, w = t / WAVE;
    fibre &f = b.fibres[t];
    f.waiting = true; f.released = false; f.site = site; f.kind = kind; f.group = 0;
    for (;;) {
        wave_decide(w);
        if (f.released) break;
        yield();
    }
    f.released = false;
    return f.group;
}

inline void trampoline() {
    block_state &b = *g_block;
#ifdef HOSTEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &b.sched_bottom, &b.sched_size);
#endif
    (*b.body)();
    fibre &f = b.fibres[b.current];
    f.done = true;
    b.live -= 1;
    b.progress += 1;
    const int w = b.current / WAVE;
    b.wave_live[w] -= 1;
    // a thread that has returned no longer takes part in barriers: release who waits for it
    if (b.live > 0 && b.arrived == b.live) { b.arrived = 0; b.generation += 1; }
    if (b.wave_live[w] > 0) wave_decide(w);
    switch_to_sched(true);
}

inline void block_barrier() {
    block_state &b = *g_block;
    fibre &f = b.fibres[b.current];
    const uint64_t mine = b.generation;
    b.arrived += 1;
    if (b.arrived == b.live) { b.arrived = 0; b.generation += 1; b.progress += 1; return; }
    f.at_block_barrier = true;
    wave_decide(b.current / WAVE);      // (lanes of this wave may wait at a wave call site for "everybody else is blocked")
    while (b.generation == mine) yield();
    f.at_block_barrier = false;
}

inline void run_block(dim3 block, const std::function<void()> &body) {
    const int n = (int)(block.x * block.y * block.z);
    if (n > MAX_THREADS) { fprintf(stderr, "hostemu: block of %d threads\n", n); abort(); }
    block_state b;
    b.fibres.resize(n);
    b.body = &body;
    b.live = n;
    for (int w = 0; w < MAX_THREADS / WAVE; ++w) { b.wave_live[w] = 0; b.wave_bits[w] = 0; }
    while ((int)g_stacks.size() < n) g_stacks.push_back((char *)malloc(STACK_BYTES));
    g_block = &b;
    for (int t = 0; t < n; ++t) {
        fibre &f = b.fibres[t];
        f.stack = g_stacks[t];
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        b.wave_live[t / WAVE] += 1;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
#ifdef HOSTEMU_TSAN
        f.tsan_fiber = __tsan_create_fiber(0);
#endif
    }
#ifdef HOSTEMU_TSAN
    b.sched_tsan = __tsan_get_current_fiber();
#endif
    int idle_passes = 0;
    while (b.live > 0) {
        const uint64_t before = b.progress;
        for (int t = 0; t < n; ++t) {
            fibre &f = b.fibres[t];
            if (f.done) continue;
            b.current = t;
            threadIdx = f.tid;
#ifdef HOSTEMU_ASAN
            __sanitizer_start_switch_fiber(&b.sched_fake, f.stack, STACK_BYTES);
#endif
#ifdef HOSTEMU_TSAN
            __tsan_switch_to_fiber(f.tsan_fiber, 0);
#endif
            swapcontext(&b.sched, &f.ctx);
#ifdef HOSTEMU_ASAN
            __sanitizer_finish_switch_fiber(b.sched_fake, nullptr, nullptr);
#endif
        }
        idle_passes = (b.progress == before) ? idle_passes + 1 : 0;
        if (idle_passes > 2) {
            int at_sites = 0;
            for (int t = 0; t < n; ++t) at_sites += (!b.fibres[t].done && b.fibres[t].waiting) ? 1 : 0;
            fprintf(stderr, "hostemu: block (%u,%u,%u) is stuck: %d live threads, %d at the block barrier, %d at wave votes / "
                            "shuffles -- a __syncthreads that not every live thread reaches\n",
                    blockIdx.x, blockIdx.y, blockIdx.z, b.live, b.arrived, at_sites);
            abort();
        }
    }
#ifdef HOSTEMU_TSAN
    for (int t = 0; t < n; ++t) __tsan_destroy_fiber(b.fibres[t].tsan_fiber);
#endif
    g_block = nullptr;
}

inline void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    gridDim = grid;
    blockDim = block;
    for (uint32_t bz = 0; bz < grid.z; ++bz)
        for (uint32_t by = 0; by < grid.y; ++by)
            for (uint32_t bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                run_block(block, body);
            }
}

template <typename T>
inline T shfl_down(int site, T v, unsigned delta, int width = 64) {
    block_state &b = *g_block;
    const int w = b.current / WAVE, lane = b.current % WAVE;
    static_assert(sizeof(T) <= sizeof(double), "hostemu shuffle: 8-byte values");
    double slot = 0.0;
    memcpy(&slot, &v, sizeof(T));
    b.wave_buf[w][lane] = slot;
    const uint64_t active = wave_rendezvous(2 * site, KIND_SHUFFLE);
    T r = v;       // (a source lane that is not active: the hardware's result is undefined; here the lane's own value)
    const int src = lane + (int)delta;
    if (src < WAVE && (lane / width) == (src / width) && ((active >> src) & 1)) memcpy(&r, &b.wave_buf[w][src], sizeof(T));
    wave_rendezvous(2 * site + 1, KIND_SHUFFLE);      // everybody has read before anybody writes again
    return r;
}

inline unsigned long long vote(int site, int pred, unsigned long long *active) {
    block_state &b = *g_block;
    const int w = b.current / WAVE, lane = b.current % WAVE;
    if (pred) b.wave_bits[w] |= (1ull << lane);
    else b.wave_bits[w] &= ~(1ull << lane);
    *active = wave_rendezvous(2 * site, KIND_VOTE);
    const unsigned long long r = b.wave_bits[w] & *active;
    wave_rendezvous(2 * site + 1, KIND_VOTE);
    return r;
}
}  // namespace hostemu

// `kernel` may be a parenthesised template-id; calling it by name keeps its default arguments
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ((void)(shmem), (void)(stream), hostemu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); }))

inline void __syncthreads() { hostemu::block_barrier(); }
// (one site number per textual call)
#define __shfl_down(...) hostemu::shfl_down(__COUNTER__ + 1, __VA_ARGS__)
inline unsigned long long hostemu_ballot(int site, int pred) { unsigned long long a; return hostemu::vote(site, pred, &a); }
inline int hostemu_all(int site, int pred) { unsigned long long a; const unsigned long long r = hostemu::vote(site, pred, &a); return r == a; }
inline int hostemu_any(int site, int pred) { unsigned long long a; return hostemu::vote(site, pred, &a) != 0; }
#define __ballot(p) hostemu_ballot(__COUNTER__ + 1, (p))
#define __all(p) hostemu_all(__COUNTER__ + 1, (p))
#define __any(p) hostemu_any(__COUNTER__ + 1, (p))
#define __popcll(x) __builtin_popcountll(x)

struct double2 { double x, y; };
inline double2 make_double2(double x, double y) { return double2{x, y}; }

// ---- gfx950 builtins the kernels use ---------------------------------------------------------------------------
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / __builtin_sqrt(x))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)

// ---- runtime ---------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotSupported = 801,
    hipErrorUnknown = 999
};
typedef struct hostemu_stream *hipStream_t;
struct hostemu_event { std::chrono::steady_clock::time_point t; };
typedef hostemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer; void *hostPointer; };
typedef void *hipMemGenericAllocationHandle_t;
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; int requestedHandleType; hipMemLocation location; void *win32HandleMetaData; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };

inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hostemu: error"; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceGetPCIBusId(char *s, int n, int) { snprintf(s, n, "0000:00:00.0"); return hipSuccess; }
// exact sizes, so that AddressSanitizer sees the first byte past an array
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
template <typename T> inline hipError_t hipMallocAsync(T **p, size_t n, hipStream_t s) { return hipMallocAsync((void **)p, n, s); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipFreeAsync(void *p, hipStream_t) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hostemu_event(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    a->type = hipMemoryTypeDevice; a->device = 0; a->devicePointer = (void *)p; a->hostPointer = nullptr;
    return hipSuccess;
}
// no virtual-memory management: the placement arena (prt_placed.h) reports that it cannot be created
inline hipError_t hipMemAddressReserve(void **, size_t, size_t, void *, unsigned long long) { return hipErrorNotSupported; }
inline hipError_t hipMemAddressFree(void *, size_t) { return hipErrorNotSupported; }
inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *, size_t, const hipMemAllocationProp *, unsigned long long) { return hipErrorNotSupported; }
inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipErrorNotSupported; }
inline hipError_t hipMemMap(void *, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return hipErrorNotSupported; }
inline hipError_t hipMemUnmap(void *, size_t) { return hipErrorNotSupported; }
inline hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipErrorNotSupported; }
