// prt.hip -- C ABI (include/prt.h) of the gfx950 sequential raytrace engine: argument
// checking, table upload, kernel selection and launches.  Kernels: prt_kernels.h; per-ray
// device functions: prt_device.h (isotropic), prt_aniso.h (crystals).
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC prt.hip -o libprt.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <new>
#include <mutex>
#include <vector>
#include <algorithm>
#include <initializer_list>
#include "prt_kernels.h"


// ---------------------------------------------------------------------------
// host-side state
// ---------------------------------------------------------------------------
struct prt_system {
    int32_t device;
    int32_t n_surfaces;
    int32_t all_isotropic;
    int32_t all_conic;
    int32_t shape_level;       // PRT_SHAPES_*: the least kernel instantiation that covers the table's shapes
    prt_dev_surface *d_table;  // device records (prt_device.h: the caller's records repacked, 504 B each)
    prt_surface_t *h_table;    // host copy of the caller's records (dispatch decisions)
    void *d_side;              // one device array: the coefficients / term powers / spline data in use
    int32_t complex_eps;       // some crystal of the table has a complex (absorbing) epsilon tensor
    double *d_eps_im;          // (n_surfaces, 9): imaginary parts of the epsilon tensors, only if complex_eps
    // tables with crystals (at most PRT_FUSED_MAX_CRYSTALS interfaces): what k_trace_general executes
    prt_hot_surface *d_hot;    // (n_surfaces) hot blocks of the records (prt_device.h)
    walk_step *d_walk;         // the walk program (prt_kernels.h), n_walk entries + a sentinel
    int32_t n_walk;
    int32_t n_aniso;           // crystal interfaces of the table
    // prt_system_update: capacities of the device arrays, and the ring of page-locked staging slots of its copies
    size_t side_capacity, walk_capacity;
    char *stage;
    size_t stage_slot_bytes;
    hipEvent_t stage_ev[4];
    uint32_t stage_used;
    int stage_next;
    // set when prt_system_update failed AFTER some of its copies were enqueued: the device holds a mix of two tables
    // while the host-side facts still describe the old one -- every entry point refuses the system from then on
    int32_t poisoned;
};
#define PRT_STAGE_SLOTS 4

static void free_system(prt_system *sys) {
    if (!sys) return;
    if (sys->d_side) (void)hipFree(sys->d_side);
    if (sys->d_eps_im) (void)hipFree(sys->d_eps_im);
    if (sys->d_hot) (void)hipFree(sys->d_hot);
    if (sys->d_walk) (void)hipFree(sys->d_walk);
    for (int q = 0; q < 4; ++q)
        if (sys->stage_ev[q]) {
            (void)hipEventSynchronize(sys->stage_ev[q]);
            (void)hipEventDestroy(sys->stage_ev[q]);
        }
    if (sys->stage) (void)hipHostFree(sys->stage);
    if (sys->d_table) (void)hipFree(sys->d_table);
    delete[] sys->h_table;
    delete sys;
}

static thread_local char g_err[512] = "";

static int32_t fail(int32_t code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess)
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

// Entry points run on `device` but leave the calling thread's current device untouched (the
// host framework -- torch -- tracks its own notion of the current device).
struct device_guard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit device_guard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err == hipSuccess) prev = -1;  // nothing to restore
    }
    ~device_guard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define PRT_ON_DEVICE(dev)                                                            \
    device_guard guard_((dev));                                                       \
    if (guard_.err != hipSuccess) return fail(PRT_ERR_DEVICE, "hipSetDevice", guard_.err)

#define HIP_TRY(call)                                                   \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return fail(PRT_ERR_DEVICE, #call, e_);   \
    } while (0)

// a system whose in-place update failed half way (prt_system_update) is refused by everything that would read it
#define PRT_SYS_USABLE(sys)                                                                                  \
    if ((sys) && (sys)->poisoned)                                                                            \
    return fail(PRT_ERR_DEVICE, "this system was left unusable by a failed prt_system_update: destroy it")

// Stream-ordered scratch memory that is returned on every exit path (also the error returns).
struct stream_scratch {
    hipStream_t st;
    void *ptrs[16];
    int n = 0;
    explicit stream_scratch(hipStream_t s) : st(s) {}
    template <typename T>
    hipError_t get(T **out, size_t bytes) {
        *out = nullptr;
        if (n >= 16) return hipErrorOutOfMemory;
        hipError_t e = hipMallocAsync((void **)out, bytes ? bytes : 1, st);
        if (e == hipSuccess) ptrs[n++] = (void *)*out;
        return e;
    }
    ~stream_scratch() {
        for (int i = 0; i < n; ++i) (void)hipFreeAsync(ptrs[i], st);
    }
};

struct event_pair {
    hipEvent_t a = nullptr, b = nullptr;
    ~event_pair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

#include "prt_placed.h"

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static inline unsigned nblocks(int64_t n, int per_block) {
    return (unsigned)((n + per_block - 1) / per_block);
}

static int32_t e_mode_of(const double *e_re, int32_t use_default_e) {
    if (e_re) return 2;
    return use_default_e ? 1 : 0;
}

// ---- launch of the fused isotropic march: picks the instantiation ---------------------------------
struct iso_launch {
    const prt_system_t *sys;
    int64_t n0, in_pitch;
    const double *x0, *k0, *e_re, *e_im;
    int32_t e_mode;
    int64_t out_pitch;
    double *x_hit, *k_out;
    uint8_t *valid, *valid_out;
    int32_t packed_flags;
    uint8_t *nonconv;
    bool uni;          // uniform first segment (k0 == NULL): fu holds k and E / the direction
    first_uniform fu;
    bool moments;      // also reduce the image-plane moments (partials: one row of 7 per block)
    double rx, ry, rz;
    double *partials;
    bool redirect;     // the last surface's record goes to img (path mode, aligned rows)
    img_redirect img;
    hipStream_t st;
};

template <int MODE, bool VI, bool VO, int SH, bool MOM, bool UNI, bool IMG = false>
static void launch_iso_inst(const iso_launch &a) {
    // (a multiple of 8 blocks: the kernel deals contiguous eighths of the bundle to the 8 XCDs, prt_kernels.h)
    const dim3 grid((nblocks(a.n0, PRT_MARCH_BLOCK * 2) + 7u) / 8u * 8u), block(PRT_MARCH_BLOCK);
    hipLaunchKernelGGL((k_trace_iso<MODE, VI, VO, SH, MOM, UNI, IMG>), grid, block, 0, a.st, a.sys->d_table,
                       a.sys->n_surfaces, a.n0, a.in_pitch, a.x0, a.k0, a.e_re, a.e_im, a.e_mode, a.out_pitch,
                       a.x_hit, a.k_out, a.valid, a.valid_out, a.rx, a.ry, a.rz, a.partials, a.packed_flags,
                       a.nonconv, a.fu, (int32_t)(a.uni ? 1 : 0), a.img);
}

// The aligned case (16-B rows in and out: what prt_recommended_pitch gives) has compile-time variants for the
// uniform first segment and the fused moments; unaligned buffers run the general instantiations, which take
// the uniform first segment as a run-time flag (moments: two-kernel reduction behind the trace, see trace_launch).
template <int MODE, int SH>
static void launch_iso_shape(const iso_launch &a, bool vi, bool vo) {
    if (a.redirect) {  // (path mode, aligned rows: checked by the caller)
        if (MODE == PRT_MODE_PATH) {
            if (a.moments && a.uni) launch_iso_inst<PRT_MODE_PATH, true, true, SH, true, true, true>(a);
            else if (a.moments) launch_iso_inst<PRT_MODE_PATH, true, true, SH, true, false, true>(a);
            else if (a.uni) launch_iso_inst<PRT_MODE_PATH, true, true, SH, false, true, true>(a);
            else launch_iso_inst<PRT_MODE_PATH, true, true, SH, false, false, true>(a);
        }
    } else if (a.moments) {
        if (a.uni) launch_iso_inst<MODE, true, true, SH, true, true>(a);
        else launch_iso_inst<MODE, true, true, SH, true, false>(a);
    } else if (vi && vo) {
        if (a.uni) launch_iso_inst<MODE, true, true, SH, false, true>(a);
        else launch_iso_inst<MODE, true, true, SH, false, false>(a);
    } else if (vi) {
        launch_iso_inst<MODE, true, false, SH, false, false>(a);
    } else if (vo) {
        launch_iso_inst<MODE, false, true, SH, false, false>(a);
    } else {
        launch_iso_inst<MODE, false, false, SH, false, false>(a);
    }
}

template <int MODE>
static void launch_trace_iso(const iso_launch &a, bool vi, bool vo) {
    switch (a.sys->shape_level) {
        case PRT_SHAPES_CONIC: launch_iso_shape<MODE, PRT_SHAPES_CONIC>(a, vi, vo); break;
        case PRT_SHAPES_ASPHERE: launch_iso_shape<MODE, PRT_SHAPES_ASPHERE>(a, vi, vo); break;
        case PRT_SHAPES_POLY: launch_iso_shape<MODE, PRT_SHAPES_POLY>(a, vi, vo); break;
        default: launch_iso_shape<MODE, PRT_SHAPES_ALL>(a, vi, vo); break;
    }
}

// ---- what the fused crystal march executes: hot blocks and the walk program (prt_kernels.h) ---------------
static prt_hot_surface hot_block(const prt_surface_t &r) {
    prt_hot_surface h;
    memset(&h, 0, sizeof h);
    const uint32_t bits = (uint32_t)(r.shape_type & 15) | (uint32_t)(r.ap_type & 3) << 4 | (uint32_t)(r.interaction & 1) << 6 |
                          (uint32_t)(r.mat_type & 1) << 7 | (uint32_t)(r.aniso_class & 3) << 8 | (uint32_t)(r.frame_flags & 7) << 10;
    const uint64_t w = (uint64_t)bits | (uint64_t)(uint32_t)r.newton_maxit << 32;
    memcpy(&h.v[0], &w, 8);
    h.v[1] = r.curv;
    h.v[2] = r.cc;
    for (int q = 0; q < 3; ++q) h.v[3 + q] = r.g_shape[q];
    h.v[6] = r.ap_p0;
    h.v[7] = r.ap_p1;
    h.v[8] = r.n_after;
    h.v[9] = r.aniso_eo;
    h.v[10] = r.aniso_ee;
    for (int q = 0; q < 3; ++q) h.v[11 + q] = r.aniso_axis[q];
    return h;
}

// The depth-first walk of k_trace_general, written down: at a crystal interface child 1 is parked at the level of
// that interface and child 0 goes on; at the end of the table the deepest parked child is taken up behind its
// interface.  One entry per (surface, branch) = per record of the concatenated layout; a sentinel at the end.
static std::vector<walk_step> build_walk_program(const prt_surface_t *table, int S) {
    std::vector<int32_t> cum_in(S + 1, 0), cum_out(S + 1, 0), level(S + 1, 0);
    for (int s = 0, a = 0; s < S; ++s) {
        level[s] = a;
        cum_in[s + 1] = cum_in[s] + (1 << a);
        if (table[s].mat_type == PRT_MAT_ANISOTROPIC) ++a;
        cum_out[s + 1] = cum_out[s] + (1 << a);
        level[s + 1] = a;
    }
    std::vector<walk_step> prog;
    struct parked { int level, s_next; int64_t L; };   // (s_next - 1: the surface that parked it)
    std::vector<parked> stack;   // at most one entry per level, deepest last
    int s = 0;
    int64_t L = 0;
    int resume = 0, s_park = 0;
    for (;;) {
        for (; s < S; ++s) {
            const int a = level[s];
            walk_step w;
            const int32_t lp = (int32_t)(L & (((int64_t)1 << a) - 1));
            w.s = s;
            w.bits = a | resume << 8 | lp << 16 | ((s == S - 1) ? 1 : 0) << 24;
            w.cum_in = cum_in[s];
            w.s_park = resume ? s_park : 0;
            resume = 0;
            prog.push_back(w);
            if (table[s].mat_type == PRT_MAT_ANISOTROPIC) stack.push_back({a, s + 1, L | ((int64_t)1 << a)});
        }
        // the deepest parked child whose walk is not empty (a crystal behind the LAST surface parks children that
        // have nowhere to go)
        bool found = false;
        while (!stack.empty()) {
            const parked c = stack.back();
            stack.pop_back();
            if (c.s_next < S) {
                s = c.s_next;
                L = c.L;
                resume = c.level + 1;
                s_park = c.s_next - 1;
                found = true;
                break;
            }
        }
        if (!found) break;
    }
    walk_step end;
    memset(&end, 0, sizeof end);
    prog.push_back(end);  // sentinel: what the last step prefetches
    return prog;
}

extern "C" {

int32_t prt_abi_version(void) { return PRT_ABI_VERSION; }
int32_t prt_sizeof_surface(void) { return (int32_t)sizeof(prt_surface_t); }

int32_t prt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PRT_ERR_NO_DEVICE, "hipGetDeviceCount", e);
    return n;
}

const char *prt_strerror(int32_t code) {
    switch (code) {
        case PRT_OK: return "ok";
        case PRT_ERR_INVALID_ARG: return "invalid argument";
        case PRT_ERR_UNSUPPORTED: return "unsupported shape/material";
        case PRT_ERR_DEVICE: return "HIP runtime error";
        case PRT_ERR_NO_DEVICE: return "no HIP device";
        case PRT_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

const char *prt_last_error(void) { return g_err; }

// The polynomial part of a surface (xypoly: all terms; combination: those behind the asphere part) as the
// dense Horner rows xypoly_eval reads: header ints, then 32-byte chunks (layout: prt_device.h).
#define PRT_MAX_POLY_POWER 64
static std::vector<char> poly_rows(const prt_surface_t &r, bool *dense_out) {
    std::vector<char> blob;
    *dense_out = false;
    int t0 = 0;
    if (r.shape_type == PRT_SHAPE_COMBO) t0 = r.n_asphere;
    else if (r.shape_type != PRT_SHAPE_XYPOLY) return blob;
    int D = -1;
    for (int t = t0; t < r.n_coeffs; ++t) D = r.xpow[t] > D ? r.xpow[t] : D;
    const int nrows = D + 1;
    std::vector<std::vector<double>> row(nrows);  // row[i][j] = c_ij (terms with equal powers add up)
    for (int t = t0; t < r.n_coeffs; ++t) {
        std::vector<double> &w = row[r.xpow[t]];
        if ((int)w.size() <= r.ypow[t]) w.resize(r.ypow[t] + 1, 0.0);
        w[r.ypow[t]] += r.coeffs[t];
    }
    std::vector<int32_t> head(2 + nrows + ((2 + nrows) & 1), 0);
    std::vector<double> data;
    for (int i = D; i >= 0; --i) {
        const int len = (int)row[i].size(), nch = (len + 3) / 4;
        head[2 + (D - i)] = nch;
        for (int q = 0; q < 4 * nch - len; ++q) data.push_back(0.0);  // leading zeros: p, dp stay 0
        for (int j = len - 1; j >= 0; --j) data.push_back(row[i][j]);
    }
    head[0] = nrows;
    head[1] = (int32_t)(data.size() / 4);
    // the dense triangle of total degree <= PRT_POLY_DENSE_DEG in front (dense_poly_eval); zeros if it does not fit
    double dense[PRT_POLY_DENSE_SLOTS] = {0};
    bool fits = true;
    for (int i = 0; i <= D; ++i)
        for (size_t j = 0; j < row[i].size(); ++j)
            if (row[i][j] != 0.0 && i + (int)j > PRT_POLY_DENSE_DEG) fits = false;
    if (fits) {
        int q = 0;
        for (int i = PRT_POLY_DENSE_DEG; i >= 0; --i)
            for (int j = PRT_POLY_DENSE_DEG - i; j >= 0; --j, ++q)
                dense[q] = (i <= D && j < (int)row[i].size()) ? row[i][j] : 0.0;
    }
    *dense_out = fits;
    blob.resize(sizeof dense + 4 * head.size() + 8 * data.size());
    memcpy(blob.data(), dense, sizeof dense);
    memcpy(blob.data() + sizeof dense, head.data(), 4 * head.size());
    if (!data.empty()) memcpy(blob.data() + sizeof dense + 4 * head.size(), data.data(), 8 * data.size());
    return blob;
}

static int32_t check_record(const prt_surface_t *r, int idx) {
    char msg[128];
    if (r->shape_type < PRT_SHAPE_CONIC || r->shape_type > PRT_SHAPE_GRIDSAG) {
        snprintf(msg, sizeof msg, "surface %d: unknown shape_type %d", idx, r->shape_type);
        return fail(PRT_ERR_UNSUPPORTED, msg);
    }
    if (r->n_coeffs < 0 || r->n_coeffs > PRT_MAX_COEFFS ||
        (r->shape_type == PRT_SHAPE_BICONIC && 2 * r->n_coeffs > PRT_MAX_COEFFS)) {
        snprintf(msg, sizeof msg, "surface %d: n_coeffs %d out of range", idx, r->n_coeffs);
        return fail(PRT_ERR_INVALID_ARG, msg);
    }
    if (r->shape_type == PRT_SHAPE_GRIDSAG && (r->grid_nx < 8 || r->grid_ny < 8 || !r->aux)) {
        snprintf(msg, sizeof msg, "surface %d: grid sag needs >= 8 knots per direction and its data", idx);
        return fail(PRT_ERR_INVALID_ARG, msg);
    }
    if (r->shape_type == PRT_SHAPE_COMBO && (r->n_asphere < 0 || r->n_asphere > r->n_coeffs)) {
        snprintf(msg, sizeof msg, "surface %d: n_asphere %d out of range", idx, r->n_asphere);
        return fail(PRT_ERR_INVALID_ARG, msg);
    }
    if (r->shape_type == PRT_SHAPE_XYPOLY || r->shape_type == PRT_SHAPE_COMBO) {
        for (int t = (r->shape_type == PRT_SHAPE_COMBO ? r->n_asphere : 0); t < r->n_coeffs; ++t)
            if (r->xpow[t] < 0 || r->ypow[t] < 0 || r->xpow[t] > PRT_MAX_POLY_POWER || r->ypow[t] > PRT_MAX_POLY_POWER) {
                snprintf(msg, sizeof msg, "surface %d: polynomial term %d has powers (%d, %d) outside 0..%d", idx, t,
                         r->xpow[t], r->ypow[t], PRT_MAX_POLY_POWER);
                return fail(PRT_ERR_INVALID_ARG, msg);
            }
    }
    if (r->ap_type < PRT_AP_NONE || r->ap_type > PRT_AP_RECTANGULAR ||
        r->interaction < PRT_REFRACT || r->interaction > PRT_MIRROR ||
        r->mat_type < PRT_MAT_ISOTROPIC || r->mat_type > PRT_MAT_ANISOTROPIC) {
        snprintf(msg, sizeof msg, "surface %d: bad aperture/interaction/material enum", idx);
        return fail(PRT_ERR_INVALID_ARG, msg);
    }
    return PRT_OK;
}

// What prt_system_create uploads and prt_system_update re-uploads, built on the host: the device records (coefficient
// pointers as OFFSETS into the side array until `relocate` adds its base), the side array, the hot blocks and the
// walk program of tables with crystals, the imaginary parts of absorbing media, and the dispatch facts.
struct table_image {
    std::vector<prt_dev_surface> recs;
    std::vector<char> side;            // side_bytes + slack
    std::vector<prt_hot_surface> hot;
    std::vector<walk_step> walk;       // empty: no fused crystal march for this table
    std::vector<double> eps_im;        // (S, 9), only if complex_eps
    int32_t all_isotropic = 1, all_conic = 1, shape_level = PRT_SHAPES_CONIC, complex_eps = 0, n_aniso = 0;
    void relocate(const char *d_side) {
        for (prt_dev_surface &d : recs) {
            d.coeffs = (const double *)(d_side + (uintptr_t)d.coeffs);
            d.pows = (const void *)(d_side + (uintptr_t)d.pows);
        }
    }
};

static int32_t build_table_image(const prt_surface_t *table, int32_t n_surfaces, table_image &im) {
    for (int s = 0; s < n_surfaces; ++s) {
        int32_t rc = check_record(table + s, s);
        if (rc != PRT_OK) return rc;
    }
    // Absorbing media -- complex epsilon tensors (material_anisotropic.py:52-56), complex refractive indices (Im n in
    // eps_im[0] of an isotropic record) -- are supported wherever the reference's result is defined: inside crystals,
    // and for an isotropic medium behind the LAST surface (its complex k = k_inplane + xi n is unique).  Behind an
    // EARLIER isotropic interface the reference takes E from an SVD whose null space is two-dimensional for a complex
    // k (material_isotropic.py:72-128): the Poynting direction, and with it every later hit point, is LAPACK's
    // arbitrary pick -- there is nothing to be compatible with.
    {
        auto complex_medium = [](const prt_surface_t &r) {
            if (r.mat_type == PRT_MAT_ANISOTROPIC) {
                for (int q = 0; q < 9; ++q)
                    if (r.eps_im[q] != 0.0) return true;
                return false;
            }
            return r.eps_im[0] != 0.0;
        };
        int first_complex = -1;
        for (int s = 0; s < n_surfaces && first_complex < 0; ++s)
            if (complex_medium(table[s])) first_complex = s;
        if (first_complex >= 0) {
            for (int s = first_complex; s < n_surfaces; ++s) {
                const bool iso = table[s].mat_type != PRT_MAT_ANISOTROPIC;
                const bool bad_place = iso && s != n_surfaces - 1;
                const bool bad_mirror = iso && table[s].eps_im[0] != 0.0 && table[s].interaction == PRT_MIRROR;
                if (bad_place || bad_mirror) {
                    char msg[256];
                    if (bad_mirror)
                        snprintf(msg, sizeof msg, "surface %d: a mirror inside an absorbing isotropic medium", s);
                    else
                        snprintf(msg, sizeof msg, "surface %d: an isotropic medium behind the absorbing medium of surface "
                                                  "%d before the last surface of the sequence -- complex wave vectors are "
                                                  "defined inside crystals and behind the last surface only", s, first_complex);
                    return fail(PRT_ERR_UNSUPPORTED, msg);
                }
            }
            im.complex_eps = 1;
        }
    }
    // side array: per surface its doubles (coefficients -- for an asphere part followed by the products
    // (n+1) a_n --, or the grid-sag spline), then the polynomial part as dense Horner rows (poly_rows)
    // coefficients of the even-asphere part of a surface: its products (n+1) a_n follow the doubles of the
    // host table in the side array (asphere_prefetch)
    auto n_asphere_part = [](const prt_surface_t &r) -> size_t {
        if (r.shape_type == PRT_SHAPE_ASPHERE) return (size_t)r.n_coeffs;
        if (r.shape_type == PRT_SHAPE_COMBO) return (size_t)r.n_asphere;
        return 0;
    };
    auto n_doubles = [n_asphere_part](const prt_surface_t &r) -> size_t {
        if (r.shape_type == PRT_SHAPE_GRIDSAG)
            return (size_t)r.grid_nx + (size_t)r.grid_ny + (size_t)(r.grid_nx - 4) * (size_t)(r.grid_ny - 4);
        if (r.shape_type == PRT_SHAPE_BICONIC) return 2 * (size_t)r.n_coeffs;
        return (size_t)r.n_coeffs + n_asphere_part(r);
    };
    // slack behind the last entry: asphere_prefetch reads PRT_ASPHERE_PREFETCH doubles from the start of a
    // surface's coefficients whatever their number, xypoly_eval one 32-byte chunk beyond the last
    const size_t PRT_SIDE_SLACK = 8 + 8 * PRT_ASPHERE_PREFETCH;
    try {
        size_t side_bytes = 0;
        std::vector<std::vector<char>> poly((size_t)n_surfaces);
        std::vector<char> poly_dense((size_t)n_surfaces, 0);
        for (int s = 0; s < n_surfaces; ++s) {
            bool dense = false;
            poly[s] = poly_rows(table[s], &dense);
            poly_dense[s] = (dense && table[s].shape_type == PRT_SHAPE_XYPOLY) ? 1 : 0;
            side_bytes += 8 * n_doubles(table[s]) + poly[s].size();
        }
        im.side.assign(side_bytes + PRT_SIDE_SLACK, 0);
        im.recs.resize((size_t)n_surfaces);
        char *h_side = im.side.data();
        size_t off = 0;
        for (int s = 0; s < n_surfaces; ++s) {
            const prt_surface_t &r = table[s];
            prt_dev_surface &d = im.recs[s];
            if (r.mat_type != PRT_MAT_ISOTROPIC || im.complex_eps) im.all_isotropic = 0;
            if (r.mat_type == PRT_MAT_ANISOTROPIC) ++im.n_aniso;
            if (r.shape_type != PRT_SHAPE_CONIC) im.all_conic = 0;
            if (r.shape_type == PRT_SHAPE_ASPHERE && im.shape_level < PRT_SHAPES_ASPHERE) im.shape_level = PRT_SHAPES_ASPHERE;
            if ((r.shape_type == PRT_SHAPE_XYPOLY || r.shape_type == PRT_SHAPE_BICONIC) && im.shape_level < PRT_SHAPES_POLY)
                im.shape_level = PRT_SHAPES_POLY;
            if (r.shape_type == PRT_SHAPE_COMBO || r.shape_type == PRT_SHAPE_GRIDSAG) im.shape_level = PRT_SHAPES_ALL;
            memset(&d, 0, sizeof d);
            d.shape_type = r.shape_type;
            d.n_coeffs = r.n_coeffs;
            d.ap_type = r.ap_type;
            d.interaction = r.interaction;
            d.mat_type = r.mat_type;
            d.frame_flags = r.frame_flags;
            d.newton_maxit = r.newton_maxit;
            d.aniso_class = r.aniso_class;
            d.n_asphere = r.n_asphere;
            d.grid_nx = r.grid_nx;
            d.grid_ny = r.grid_ny;
            d.poly_dense = poly_dense[s];
            d.curv = r.curv;
            d.cc = r.cc;
            memcpy(d.B_shape, r.B_shape, sizeof d.B_shape);
            memcpy(d.g_shape, r.g_shape, sizeof d.g_shape);
            memcpy(d.B_ap, r.B_ap, sizeof d.B_ap);
            memcpy(d.g_ap, r.g_ap, sizeof d.g_ap);
            d.ap_p0 = r.ap_p0;
            d.ap_p1 = r.ap_p1;
            memcpy(d.B_mat, r.B_mat, sizeof d.B_mat);
            d.n_after = r.n_after;
            memcpy(d.eps_re, r.eps_re, sizeof d.eps_re);
            if (r.mat_type == PRT_MAT_ANISOTROPIC && !im.complex_eps) {
                // A lossless crystal's tensor is symmetric; one that a caller ROTATES into place (R diag R^T, the usual
                // way to build it) comes out symmetric up to an ulp.  The solver's cheapest route -- flux and ray
                // direction of the leaving pair from the adjugate of W, no eigenvectors (prt_aniso.h) -- asks for
                // eps[i][j] == eps[j][i] exactly, so such a tensor is stored as its symmetric part: an antisymmetric
                // part below 1e-14 of the largest entry is rounding, not physics (a change of <= 1e-14 relative in the
                // tensor against the 1e-10 parity bar; the uniaxial / isotropic classes are recognised with 1e-12,
                // surface_table.classify_eps).  Anything more asymmetric is taken as given.
                double scale = 0.0, asym = 0.0;
                for (int q = 0; q < 9; ++q) scale = fmax(scale, fabs(r.eps_re[q]));
                for (int i = 0; i < 3; ++i)
                    for (int j = i + 1; j < 3; ++j) asym = fmax(asym, fabs(r.eps_re[3 * i + j] - r.eps_re[3 * j + i]));
                if (asym > 0.0 && asym <= 1e-14 * scale)
                    for (int i = 0; i < 3; ++i)
                        for (int j = i + 1; j < 3; ++j)
                            d.eps_re[3 * i + j] = d.eps_re[3 * j + i] = 0.5 * (r.eps_re[3 * i + j] + r.eps_re[3 * j + i]);
            }
            d.aniso_eo = r.aniso_eo;
            d.aniso_ee = r.aniso_ee;
            memcpy(d.aniso_axis, r.aniso_axis, sizeof d.aniso_axis);
            d.curv_y = r.curv_y;
            d.cc_y = r.cc_y;
            d.asphere_scale = r.asphere_scale;
            const size_t nd = n_doubles(r);
            d.coeffs = (const double *)(uintptr_t)off;          // (offset: table_image::relocate)
            const size_t na = n_asphere_part(r);
            memcpy(h_side + off, r.shape_type == PRT_SHAPE_GRIDSAG ? (const void *)r.aux : (const void *)r.coeffs,
                   8 * (nd - na));
            for (size_t n = 0; n < na; ++n) ((double *)(h_side + off))[nd - na + n] = (double)(n + 1) * r.coeffs[n];
            off += 8 * nd;
            d.pows = (const void *)(uintptr_t)off;
            if (!poly[s].empty()) memcpy(h_side + off, poly[s].data(), poly[s].size());
            off += poly[s].size();
        }
        if (im.complex_eps) {
            im.eps_im.assign((size_t)n_surfaces * 9, 0.0);
            for (int s = 0; s < n_surfaces; ++s)
                for (int q = 0; q < 9; ++q)   // (isotropic records: Im n in slot 0)
                    im.eps_im[(size_t)s * 9 + q] = (table[s].mat_type == PRT_MAT_ANISOTROPIC || q == 0) ? table[s].eps_im[q] : 0.0;
        }
        if (im.n_aniso > 0 && im.n_aniso <= PRT_FUSED_MAX_CRYSTALS && !im.complex_eps) {
            // the fused crystal march (k_trace_general): hot blocks + walk program
            im.hot.resize((size_t)n_surfaces);
            for (int s = 0; s < n_surfaces; ++s) im.hot[s] = hot_block(table[s]);
            im.walk = build_walk_program(table, n_surfaces);
        }
    } catch (...) {  // std::bad_alloc: the ABI never throws
        return fail(PRT_ERR_NOMEM, "host alloc");
    }
    return PRT_OK;
}

static void adopt_image_facts(prt_system *sys, const table_image &im) {
    sys->all_isotropic = im.all_isotropic;
    sys->all_conic = im.all_conic;
    sys->shape_level = im.shape_level;
    sys->complex_eps = im.complex_eps;
    sys->n_aniso = im.n_aniso;
    sys->n_walk = im.walk.empty() ? 0 : (int32_t)im.walk.size() - 1;
}

int32_t prt_system_create(const prt_surface_t *table, int32_t n_surfaces, int32_t device,
                          prt_system_t **out) {
    if (!table || !out || n_surfaces <= 0) return fail(PRT_ERR_INVALID_ARG, "prt_system_create: null/empty");
    *out = nullptr;
    table_image im;
    int32_t rc = build_table_image(table, n_surfaces, im);
    if (rc != PRT_OK) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(PRT_ERR_NO_DEVICE, "no HIP device visible", e);
    if (device < 0 || device >= ndev) return fail(PRT_ERR_INVALID_ARG, "device index out of range");
    PRT_ON_DEVICE(device);
    prt_system *sys = new (std::nothrow) prt_system();
    if (!sys) return fail(PRT_ERR_NOMEM, "host alloc");
    memset((void *)sys, 0, sizeof *sys);
    sys->device = device;
    sys->n_surfaces = n_surfaces;
    sys->h_table = new (std::nothrow) prt_surface_t[n_surfaces];
    if (!sys->h_table) {
        free_system(sys);
        return fail(PRT_ERR_NOMEM, "host alloc");
    }
    memcpy(sys->h_table, table, sizeof(prt_surface_t) * n_surfaces);
    adopt_image_facts(sys, im);
    sys->side_capacity = im.side.size();
    e = hipMalloc(&sys->d_side, im.side.size());
    if (e != hipSuccess) {
        free_system(sys);
        return fail(PRT_ERR_NOMEM, "hipMalloc(coefficients)", e);
    }
    im.relocate((const char *)sys->d_side);
    e = hipMemcpy(sys->d_side, im.side.data(), im.side.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && im.complex_eps) {
        e = hipMalloc((void **)&sys->d_eps_im, sizeof(double) * im.eps_im.size());
        if (e == hipSuccess)
            e = hipMemcpy(sys->d_eps_im, im.eps_im.data(), sizeof(double) * im.eps_im.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&sys->d_table, sizeof(prt_dev_surface) * n_surfaces);
    if (e == hipSuccess)
        e = hipMemcpy(sys->d_table, im.recs.data(), sizeof(prt_dev_surface) * n_surfaces, hipMemcpyHostToDevice);
    if (e == hipSuccess && !im.walk.empty()) {
        e = hipMalloc((void **)&sys->d_hot, sizeof(prt_hot_surface) * im.hot.size());
        if (e == hipSuccess)
            e = hipMemcpy(sys->d_hot, im.hot.data(), sizeof(prt_hot_surface) * im.hot.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&sys->d_walk, sizeof(walk_step) * im.walk.size());
        if (e == hipSuccess)
            e = hipMemcpy(sys->d_walk, im.walk.data(), sizeof(walk_step) * im.walk.size(), hipMemcpyHostToDevice);
        sys->walk_capacity = im.walk.size();
    }
    if (e != hipSuccess) {
        free_system(sys);
        return fail(PRT_ERR_DEVICE, "prt_system_create: table upload", e);
    }
    *out = sys;
    return PRT_OK;
}

// The table of an existing system replaced IN PLACE, stream-ordered: the optimiser's pattern (SURVEY.md 3.5 -- one
// curvature moved, trace, the next one moved, trace) costs one small asynchronous copy per trace instead of the three
// allocations, three blocking copies and three frees of a destroy + create.  The new table must have the shape of the
// old one: the same number of surfaces, a side array that fits the one allocated, the same absorbing-media status
// and -- with crystals -- a walk program that fits; otherwise PRT_ERR_UNSUPPORTED, nothing touched, and the caller
// creates a new system.  Ordering: the copies are enqueued on `stream` -- launches enqueued there before see the old
// table, launches enqueued there afterwards the new one; traces of this system on OTHER streams must have completed.
// The host side of the copies is a ring of page-locked staging slots (a slot is reused only after its copy has
// finished), so the call never waits for the device in steady state.
int32_t prt_system_update(prt_system_t *sys, const prt_surface_t *table, int32_t n_surfaces, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || !table) return fail(PRT_ERR_INVALID_ARG, "prt_system_update: null argument");
    if (n_surfaces != sys->n_surfaces) return fail(PRT_ERR_UNSUPPORTED, "prt_system_update: another number of surfaces");
    table_image im;
    int32_t rc = build_table_image(table, n_surfaces, im);
    if (rc != PRT_OK) return rc;
    if (im.side.size() > sys->side_capacity || im.complex_eps != sys->complex_eps ||
        im.walk.empty() != (sys->d_walk == nullptr) || im.walk.size() > sys->walk_capacity)
        return fail(PRT_ERR_UNSUPPORTED, "prt_system_update: the new table does not fit the system's device arrays");
    PRT_ON_DEVICE(sys->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t b_recs = sizeof(prt_dev_surface) * (size_t)n_surfaces, b_side = im.side.size();
    const size_t b_hot = sizeof(prt_hot_surface) * im.hot.size(), b_walk = sizeof(walk_step) * im.walk.size();
    const size_t b_im = sizeof(double) * im.eps_im.size();
    const size_t need = b_recs + b_side + b_hot + b_walk + b_im;
    if (!sys->stage || need > sys->stage_slot_bytes) {
        if (sys->stage) {
            for (int q = 0; q < PRT_STAGE_SLOTS; ++q)
                if (sys->stage_ev[q]) (void)hipEventSynchronize(sys->stage_ev[q]);
            (void)hipHostFree(sys->stage);
            sys->stage = nullptr;
        }
        sys->stage_slot_bytes = (need + 4095) / 4096 * 4096;
        HIP_TRY(hipHostMalloc((void **)&sys->stage, sys->stage_slot_bytes * PRT_STAGE_SLOTS, hipHostMallocDefault));
        for (int q = 0; q < PRT_STAGE_SLOTS; ++q)
            if (!sys->stage_ev[q]) HIP_TRY(hipEventCreateWithFlags(&sys->stage_ev[q], hipEventDisableTiming));
        sys->stage_used = 0;
    }
    const int slot = sys->stage_next;
    sys->stage_next = (slot + 1) % PRT_STAGE_SLOTS;
    if (sys->stage_used & (1u << slot)) HIP_TRY(hipEventSynchronize(sys->stage_ev[slot]));   // its last copy is done
    char *h = sys->stage + (size_t)slot * sys->stage_slot_bytes;
    im.relocate((const char *)sys->d_side);
    memcpy(h, im.recs.data(), b_recs);
    memcpy(h + b_recs, im.side.data(), b_side);
    if (b_hot) memcpy(h + b_recs + b_side, im.hot.data(), b_hot);
    if (b_walk) memcpy(h + b_recs + b_side + b_hot, im.walk.data(), b_walk);
    if (b_im) memcpy(h + b_recs + b_side + b_hot + b_walk, im.eps_im.data(), b_im);
    // From the first enqueue on there is no way back: a failure leaves the device with pieces of two tables.  The
    // system is then POISONED (every later call on it fails with PRT_ERR_DEVICE; the caller destroys it).
    // (test hook, only in processes started with PRT_TEST_HOOKS in the environment -- read once --: PRT_TEST_FAIL_UPDATE=k
    //  makes the k-th enqueue of this call fail INSTEAD of being issued, so that the ones before it are on the stream
    //  and the ones behind it are not: the partial update the poisoning exists for)
    static const bool test_hooks = getenv("PRT_TEST_HOOKS") != nullptr;
    int fail_at = 0, step_no = 0;
    if (test_hooks)
        if (const char *v = getenv("PRT_TEST_FAIL_UPDATE")) fail_at = std::max(1, atoi(v));
    auto enqueue = [&](void *dst, const void *src, size_t bytes) {
        return (++step_no == fail_at) ? hipErrorUnknown : hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    hipError_t ce = enqueue(sys->d_table, h, b_recs);
    if (ce == hipSuccess && b_side) ce = enqueue(sys->d_side, h + b_recs, b_side);
    if (ce == hipSuccess && b_hot) ce = enqueue(sys->d_hot, h + b_recs + b_side, b_hot);
    if (ce == hipSuccess && b_walk) ce = enqueue(sys->d_walk, h + b_recs + b_side + b_hot, b_walk);
    if (ce == hipSuccess && b_im) ce = enqueue(sys->d_eps_im, h + b_recs + b_side + b_hot + b_walk, b_im);
    if (ce == hipSuccess) ce = (++step_no == fail_at) ? hipErrorUnknown : hipEventRecord(sys->stage_ev[slot], st);
    if (ce != hipSuccess) {
        sys->poisoned = 1;
        sys->stage_used &= ~(1u << slot);
        return fail(PRT_ERR_DEVICE, "prt_system_update: a copy of the new table failed after others were enqueued; the "
                                    "system is unusable (destroy it)", ce);
    }
    sys->stage_used |= 1u << slot;
    memcpy(sys->h_table, table, sizeof(prt_surface_t) * n_surfaces);
    adopt_image_facts(sys, im);
    return PRT_OK;
}

int32_t prt_system_destroy(prt_system_t *sys) {
    if (!sys) return PRT_OK;
    device_guard guard_(sys->device);
    free_system(sys);
    return PRT_OK;
}

// PRT_GENERAL_PER_SURFACE (present in the environment, whatever its value): tables with crystals take the per-surface
// march also where the fused walk would do -- the independent implementation, for cross-checks
static bool general_per_surface_forced() {
    static const bool forced = getenv("PRT_GENERAL_PER_SURFACE") != nullptr;
    return forced;
}

int32_t prt_system_layout(const prt_system_t *sys) {
    if (!sys) return fail(PRT_ERR_INVALID_ARG, "null system");
    if (sys->all_isotropic) return PRT_LAYOUT_ROW_PITCHED;
    return (sys->d_walk && !general_per_surface_forced()) ? PRT_LAYOUT_CONCATENATED_PITCHED : PRT_LAYOUT_CONCATENATED_TIGHT;
}

int32_t prt_system_num_surfaces(const prt_system_t *sys) {
    if (!sys) return fail(PRT_ERR_INVALID_ARG, "null system");
    return sys->n_surfaces;
}

int32_t prt_system_ray_counts(const prt_system_t *sys, int64_t n0, int64_t *n_in, int64_t *n_out) {
    if (!sys || n0 < 0 || !n_in || !n_out) return fail(PRT_ERR_INVALID_ARG, "prt_system_ray_counts");
    int64_t n = n0;
    for (int s = 0; s < sys->n_surfaces; ++s) {
        n_in[s] = n;
        if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) n *= 2;
        n_out[s] = n;
    }
    return PRT_OK;
}

static bool aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// march through a table that contains anisotropic media: one launch pair per surface
static int32_t trace_general(const prt_system_t *sys, int64_t n0, const double *x0,
                             const double *k0, const double *e_re, const double *e_im,
                             int32_t mode, double *x_hit, double *k_out, double *e_out,
                             double *e_out_im, uint8_t *valid, uint8_t *valid_out, uint8_t *nonconv,
                             int32_t e_mode_first, hipStream_t st, double *k_out_im = nullptr) {
    const int S = sys->n_surfaces;
    // scratch: directions after anisotropic interfaces, plus ping-pong state in IMAGE mode
    int64_t n_final = n0;
    for (int s = 0; s < S; ++s)
        if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) n_final *= 2;
    double *dirbuf[2] = {nullptr, nullptr};
    double *xbuf[2] = {nullptr, nullptr}, *kbuf[2] = {nullptr, nullptr}, *kimbuf[2] = {nullptr, nullptr};
    uint8_t *vbuf[2] = {nullptr, nullptr}, *wbuf[2] = {nullptr, nullptr};
    stream_scratch scratch(st);
    HIP_TRY(scratch.get(&dirbuf[0], sizeof(double) * 3 * n_final));
    HIP_TRY(scratch.get(&dirbuf[1], sizeof(double) * 3 * n_final));
    if (mode == PRT_MODE_IMAGE) {
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(scratch.get(&xbuf[b], sizeof(double) * 3 * n_final));
            HIP_TRY(scratch.get(&kbuf[b], sizeof(double) * 3 * n_final));
            if (sys->complex_eps) HIP_TRY(scratch.get(&kimbuf[b], sizeof(double) * 3 * n_final));
            HIP_TRY(scratch.get(&vbuf[b], n_final));
            HIP_TRY(scratch.get(&wbuf[b], n_final));
        }
    }
    uint8_t *vo_scratch = nullptr;  // valid_out storage when the caller passed NULL in PATH mode
    int64_t tot_out = 0;
    {
        int64_t n = n0;
        for (int s = 0; s < S; ++s) {
            if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) n *= 2;
            tot_out += n;
        }
    }
    if (mode == PRT_MODE_PATH && !valid_out) {
        HIP_TRY(scratch.get(&vo_scratch, tot_out));
        valid_out = vo_scratch;
    }

    const double *cur_x = x0, *cur_k = k0, *cur_dir = nullptr;
    const double *cur_k_im = nullptr;  // imaginary part of the wave vectors (tables with a complex epsilon)
    const uint8_t *cur_valid = nullptr;
    int64_t n_src = n0;  // number of distinct points in cur_x
    int64_t n = n0;
    int64_t off_in = 0, off_out = 0;  // element offsets into the concatenated outputs
    int e_mode = e_mode_first;
    if (e_mode == 3) {  // the caller's unit directions for the first segment
        cur_dir = e_re;
        e_re = nullptr;
        e_mode = 0;
    }
    for (int s = 0; s < S; ++s) {
        const prt_surface_t *rec = sys->h_table + s;
        const bool last = (s == S - 1);
        const bool aniso = rec->mat_type == PRT_MAT_ANISOTROPIC;
        const int64_t n_o = aniso ? 2 * n : n;
        double *xh_dst, *k_dst;
        uint8_t *v_dst, *vo_dst;
        if (mode == PRT_MODE_PATH) {
            xh_dst = x_hit + 3 * off_in;
            k_dst = k_out + 3 * off_out;
            v_dst = valid + off_in;
            vo_dst = valid_out + off_out;
        } else if (last) {
            xh_dst = x_hit;
            k_dst = k_out;
            v_dst = valid;
            vo_dst = valid_out ? valid_out : wbuf[s & 1];
        } else {
            xh_dst = xbuf[s & 1];
            k_dst = kbuf[s & 1];
            v_dst = vbuf[s & 1];
            vo_dst = wbuf[s & 1];
        }
        hipLaunchKernelGGL(k_propagate, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st,
                           sys->d_table + s, n, n_src, cur_x, cur_k, cur_dir,
                           (s == 0) ? e_re : nullptr, (s == 0) ? e_im : nullptr,
                           (s == 0) ? e_mode : 0, cur_valid, xh_dst, v_dst,
                           !nonconv ? (uint8_t *)nullptr
                                    : (mode == PRT_MODE_PATH ? nonconv + off_in : (last ? nonconv : (uint8_t *)nullptr)));
        double *dir_dst = dirbuf[s & 1];
        double *kim_dst = nullptr;
        if (sys->complex_eps && k_out_im)
            kim_dst = (mode == PRT_MODE_PATH) ? k_out_im + 3 * off_out : (last ? k_out_im : kimbuf[s & 1]);
        if (aniso && sys->complex_eps) {
            hipLaunchKernelGGL(k_interact_aniso_cplx, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st,
                               sys->d_table + s, sys->d_eps_im + (size_t)s * 9, n, xh_dst, cur_k, cur_k_im, cur_valid,
                               k_dst, kim_dst, dir_dst,
                               (e_out && mode == PRT_MODE_PATH) ? e_out + 3 * off_out
                                                               : ((e_out && last) ? e_out : (double *)nullptr),
                               (e_out_im && mode == PRT_MODE_PATH)
                                   ? e_out_im + 3 * off_out
                                   : ((e_out_im && last) ? e_out_im : (double *)nullptr),
                               vo_dst);
            cur_dir = dir_dst;
            cur_k_im = kim_dst;
        } else if (aniso) {
            hipLaunchKernelGGL(k_interact_aniso, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                               st, sys->d_table + s, n, xh_dst, cur_k, cur_valid, k_dst, dir_dst,
                               (e_out && mode == PRT_MODE_PATH) ? e_out + 3 * off_out
                                                               : ((e_out && last) ? e_out : (double *)nullptr),
                               (e_out_im && mode == PRT_MODE_PATH)
                                   ? e_out_im + 3 * off_out
                                   : ((e_out_im && last) ? e_out_im : (double *)nullptr),
                               vo_dst);
            cur_dir = dir_dst;
        } else if (sys->complex_eps && last && (cur_k_im || rec->eps_im[0] != 0.0)) {
            // the last surface of a table with absorbing media: complex k in and / or a complex index behind it
            hipLaunchKernelGGL(k_interact_iso_cplx, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st,
                               sys->d_table + s, rec->eps_im[0], n, xh_dst, cur_k, cur_k_im, v_dst, k_dst, kim_dst,
                               vo_dst);
            cur_dir = nullptr;
            cur_k_im = kim_dst;
        } else {
            hipLaunchKernelGGL(k_interact_iso, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st,
                               sys->d_table + s, n, xh_dst, cur_k, v_dst, k_dst,
                               (double *)nullptr, vo_dst);
            cur_dir = nullptr;  // k/|k|
        }
        cur_x = xh_dst;
        n_src = n;
        cur_k = k_dst;
        cur_valid = vo_dst;
        off_in += n;
        off_out += n_o;
        n = n_o;
    }
    HIP_TRY(hipGetLastError());
    return PRT_OK;  // `scratch` is returned to the stream's pool here (stream-ordered)
}

int64_t prt_recommended_pitch(int64_t n) {
    if (n <= 0) return 0;
    return (n + 511) / 512 * 512;  // 4 KiB of doubles: every row starts on a 128-B line
}

int64_t prt_crystal_pitch(int64_t n) {
    if (n <= 0) return 0;
    return (n + 127) / 128 * 128;  // rows of doubles AND rows of mask bytes start on 128-B lines at every level
}

// a uniform first segment for the code paths that read per-ray arrays (the per-surface march): broadcast
__global__ __launch_bounds__(PRT_BLOCK) void k_broadcast_rows(int64_t n, double v0, double v1, double v2,
                                                              double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= n) return;
    out[i] = v0;
    out[n + i] = v1;
    out[2 * n + i] = v2;
}

static prt_trace_args_t blank_args() {
    prt_trace_args_t a;
    memset(&a, 0, sizeof a);
    a.struct_bytes = (int32_t)sizeof a;
    return a;
}

// One trace call with every option (include/prt.h, prt_trace_args_t); no timing loop here.
static int32_t trace_launch(const prt_system_t *sys, const prt_trace_args_t &a) {
    PRT_SYS_USABLE(sys);
    if (!sys || a.n0 < 0) return fail(PRT_ERR_INVALID_ARG, "prt_trace: null system / negative count");
    const int64_t n0 = a.n0;
    int32_t mode = a.mode;
    const int32_t packed_flags = (mode >= 0 && (mode & PRT_MODE_FLAGS)) ? 1 : 0;
    if (mode >= 0) mode &= ~PRT_MODE_FLAGS;
    if (mode != PRT_MODE_PATH && mode != PRT_MODE_IMAGE) return fail(PRT_ERR_INVALID_ARG, "prt_trace: bad mode");
    if (packed_flags && !sys->all_isotropic)
        return fail(PRT_ERR_UNSUPPORTED, "prt_trace: PRT_MODE_FLAGS needs an all-isotropic table");
    const bool moments = a.moments_out7_dev != nullptr;
    if (moments && !a.moments_scratch_dev) return fail(PRT_ERR_INVALID_ARG, "prt_trace: moments need their scratch array");
    if (moments && !sys->all_isotropic)
        return fail(PRT_ERR_UNSUPPORTED, "prt_trace_moments: isotropic tables only (use prt_trace + prt_bundle_moments)");
    // ---- the first segment ----
    const bool uni = (a.k0 == nullptr);
    int32_t e_mode;
    const double *e_re = a.e0_re, *e_im = a.e0_im;
    first_uniform fu;
    memset(&fu, 0, sizeof fu);
    for (int q = 0; q < 3; ++q) fu.k[q] = a.k_uniform[q];
    switch (a.first_dir) {
        case PRT_FIRST_E:
            if (uni && e_re) return fail(PRT_ERR_INVALID_ARG, "prt_trace: E arrays need k arrays (k0 is NULL)");
            e_mode = e_re ? 2 : 1;
            break;
        case PRT_FIRST_K:
            e_mode = 0;
            e_re = e_im = nullptr;
            break;
        case PRT_FIRST_DIR:
            if (uni || !e_re) return fail(PRT_ERR_INVALID_ARG, "prt_trace: PRT_FIRST_DIR needs k0 and the directions in e0_re");
            e_mode = 3;
            e_im = nullptr;
            break;
        case PRT_FIRST_E_UNIFORM:
        case PRT_FIRST_DIR_UNIFORM:
            if (!uni) return fail(PRT_ERR_INVALID_ARG, "prt_trace: a uniform E / direction goes with a uniform k (k0 = NULL)");
            e_mode = (a.first_dir == PRT_FIRST_E_UNIFORM) ? 2 : 3;
            e_re = e_im = nullptr;
            for (int q = 0; q < 3; ++q) {
                fu.er[q] = a.e_uniform_re[q];
                fu.ei[q] = (a.first_dir == PRT_FIRST_E_UNIFORM) ? a.e_uniform_im[q] : 0.0;
            }
            break;
        default:
            return fail(PRT_ERR_INVALID_ARG, "prt_trace: bad first_dir");
    }
    PRT_ON_DEVICE(sys->device);
    hipStream_t st = (hipStream_t)a.stream;
    if (n0 == 0) {  // empty bundle: nothing to trace (buffers may be NULL); the moments are zero
        if (moments) HIP_TRY(hipMemsetAsync(a.moments_out7_dev, 0, sizeof(double) * MOM_VALUES, st));
        return PRT_OK;
    }
    if (!a.x0 || !a.x_hit || !a.k_out || !a.valid) return fail(PRT_ERR_INVALID_ARG, "prt_trace: null pointer");
    int64_t in_pitch = a.in_pitch, out_pitch = a.out_pitch;
    if (in_pitch == 0) in_pitch = n0;
    if (in_pitch < n0 || (out_pitch != 0 && out_pitch < n0))
        return fail(PRT_ERR_INVALID_ARG, "prt_trace: pitch smaller than the ray count");
    uint8_t *valid_out = a.valid_out;
    if (!sys->all_isotropic && a.x_img)
        return fail(PRT_ERR_UNSUPPORTED, "prt_trace: the image-plane redirect is for all-isotropic tables");
    if (!sys->all_isotropic) {
        // concatenated layout with ray pitch out_pitch (0 = n0, tight)
        if (out_pitch == 0) out_pitch = n0;
        const bool per_surface = general_per_surface_forced();
        const int n_aniso = sys->n_aniso;
        // The fused march walks the tree of split rays depth first inside one launch (k_trace_general);
        // the per-surface march (one launch pair per surface, intermediate arrays) remains for
        // sequences with more crystal interfaces than the kernel has parking slots, and as the
        // independent implementation PRT_GENERAL_PER_SURFACE=1 selects for cross-checks.
        if (per_surface || !sys->d_walk) {  // (more than 8 crystal interfaces, complex epsilon: no walk program)
            if (out_pitch != n0 || in_pitch != n0)
                return fail(PRT_ERR_UNSUPPORTED, "prt_trace: the per-surface march through crystals (more than 8 crystal "
                                                 "interfaces) takes tight arrays (pitch 0)");
            if (a.k_out_im && !sys->complex_eps)
                return fail(PRT_ERR_UNSUPPORTED, "prt_trace: k_out_im (complex k of evanescent modes) comes from the fused "
                                                 "crystal march (at most 8 crystal interfaces)");
            if (sys->complex_eps) {
                // absorbing crystals: every wave vector behind the first interface is complex
                if (!a.k_out_im)
                    return fail(PRT_ERR_INVALID_ARG, "prt_trace: a table with a complex epsilon tensor needs k_out_im "
                                                     "(the wave vectors are complex)");
                int64_t cnt = 0, nn = n0;
                for (int s = 0; s < sys->n_surfaces; ++s) {
                    if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) nn *= 2;
                    cnt += nn;
                }
                HIP_TRY(hipMemsetAsync(a.k_out_im, 0, sizeof(double) * 3 * (size_t)(mode == PRT_MODE_PATH ? cnt : nn), st));
            }
            stream_scratch rows(st);
            const double *k0 = a.k0;
            if (uni) {  // the per-surface kernels read arrays: broadcast the uniform vectors once
                double *kb = nullptr, *eb = nullptr, *ib = nullptr;
                HIP_TRY(rows.get(&kb, sizeof(double) * 3 * n0));
                hipLaunchKernelGGL(k_broadcast_rows, dim3(nblocks(n0, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, n0, fu.k[0],
                                   fu.k[1], fu.k[2], kb);
                k0 = kb;
                if (e_mode >= 2) {
                    HIP_TRY(rows.get(&eb, sizeof(double) * 3 * n0));
                    hipLaunchKernelGGL(k_broadcast_rows, dim3(nblocks(n0, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, n0,
                                       fu.er[0], fu.er[1], fu.er[2], eb);
                    e_re = eb;
                    if (e_mode == 2 && (fu.ei[0] != 0.0 || fu.ei[1] != 0.0 || fu.ei[2] != 0.0)) {
                        HIP_TRY(rows.get(&ib, sizeof(double) * 3 * n0));
                        hipLaunchKernelGGL(k_broadcast_rows, dim3(nblocks(n0, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, n0,
                                           fu.ei[0], fu.ei[1], fu.ei[2], ib);
                        e_im = ib;
                    }
                }
            }
            return trace_general(sys, n0, a.x0, k0, e_re, e_im, mode, a.x_hit, a.k_out, a.e_out_re, a.e_out_im,
                                 a.valid, valid_out, a.nonconv, e_mode, st, a.k_out_im);
        }
        const dim3 grid(nblocks(n0, PRT_GENERAL_BLOCK)), block(PRT_GENERAL_BLOCK);
        bool general_eps = false;
        for (int s = 0; s < sys->n_surfaces; ++s)
            if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC &&
                sys->h_table[s].aniso_class == PRT_ANISO_GENERAL)
                general_eps = true;
        // few crystal interfaces: the parking slots of the depth-first walk fit into LDS
        const bool park_lds = n_aniso <= PRT_PARK_LDS_LEVELS;
        // (slot: 9 doubles + 1 byte with biaxial crystals, 6 + 1 without -- prt_kernels.h "What is parked")
        const size_t park_bytes = park_lds ? (size_t)n_aniso * PRT_GENERAL_BLOCK * ((general_eps ? 9 : 6) * sizeof(double) + 1) : 0;
        const bool conics = sys->all_conic != 0;
#define PRT_LAUNCH_GS(MODE_, GEN_, LDS_, E_, SH_)                                                                   \
    hipLaunchKernelGGL((k_trace_general<MODE_, GEN_, LDS_, E_, SH_>), grid, block, park_bytes, st, sys->d_table,   \
                       sys->d_hot, sys->d_walk, sys->n_walk, n_aniso, n0, in_pitch, out_pitch, a.x0, a.k0, e_re,  \
                       e_im, e_mode, a.x_hit, a.k_out,                                                             \
                       a.e_out_re, a.e_out_im, a.valid, valid_out, a.nonconv, fu, (int32_t)(uni ? 1 : 0))
#define PRT_LAUNCH_GU(MODE_, GEN_, LDS_, E_)                                   \
    do {                                                                       \
        if (conics) PRT_LAUNCH_GS(MODE_, GEN_, LDS_, E_, PRT_SHAPES_CONIC);    \
        else PRT_LAUNCH_GS(MODE_, GEN_, LDS_, E_, PRT_SHAPES_ALL);             \
    } while (0)
#define PRT_LAUNCH_GP(MODE_, GEN_, LDS_)                      \
    do {                                                      \
        if (want_e) PRT_LAUNCH_GU(MODE_, GEN_, LDS_, true);   \
        else PRT_LAUNCH_GU(MODE_, GEN_, LDS_, false);         \
    } while (0)
#define PRT_LAUNCH_G(MODE_, GEN_)                        \
    do {                                                 \
        if (park_lds) PRT_LAUNCH_GP(MODE_, GEN_, true);  \
        else PRT_LAUNCH_GP(MODE_, GEN_, false);          \
    } while (0)
        const bool want_e = a.e_out_re != nullptr;
        if (mode == PRT_MODE_PATH) {
            if (general_eps) PRT_LAUNCH_G(PRT_MODE_PATH, true);
            else PRT_LAUNCH_G(PRT_MODE_PATH, false);
        } else {
            if (general_eps) PRT_LAUNCH_G(PRT_MODE_IMAGE, true);
            else PRT_LAUNCH_G(PRT_MODE_IMAGE, false);
        }
#undef PRT_LAUNCH_GS
#undef PRT_LAUNCH_GU
#undef PRT_LAUNCH_GP
#undef PRT_LAUNCH_G
        if (a.k_out_im) {
            // complex wave vectors of the evanescent modes: post-pass over the crystal surfaces (k_evanescent_fill)
            if (mode != PRT_MODE_PATH)
                return fail(PRT_ERR_INVALID_ARG, "prt_trace: k_out_im needs PRT_MODE_PATH (the post-pass reads the path)");
            int64_t tot_out = 0, br = 1;
            for (int s = 0; s < sys->n_surfaces; ++s) {
                if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) br *= 2;
                tot_out += br * out_pitch;
            }
            HIP_TRY(hipMemsetAsync(a.k_out_im, 0, sizeof(double) * 3 * (size_t)tot_out, st));
            int64_t off_in = 0, off_out = 0;
            br = 1;
            for (int s = 0; s < sys->n_surfaces; ++s) {
                const bool cr = sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC;
                const int64_t bp = br * out_pitch;
                if (cr) {
                    const double *k_par = (s == 0) ? a.k0 : a.k_out + 3 * (off_out - bp);
                    const int64_t par_pitch = (s == 0) ? in_pitch : bp;
                    hipLaunchKernelGGL(k_evanescent_fill, dim3(nblocks(bp, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st,
                                       sys->d_table + s, n0, out_pitch, bp, a.x_hit + 3 * off_in, k_par, par_pitch, fu,
                                       a.k_out + 3 * off_out, a.k_out_im + 3 * off_out);
                    br *= 2;
                }
                off_in += bp;
                off_out += br * out_pitch;
            }
        }
        HIP_TRY(hipGetLastError());
        return PRT_OK;
    }
    // ---- all-isotropic table: the fused march ----
    if (out_pitch == 0) out_pitch = n0;
    const bool vec_in = (in_pitch % 2 == 0) && aligned16(a.x0) && (uni || aligned16(a.k0)) &&
                        (!e_re || aligned16(e_re)) && (!e_im || aligned16(e_im));
    if (packed_flags) valid_out = nullptr;
    uint8_t *nonconv = a.nonconv;
    const bool vec_out = (out_pitch % 2 == 0) && aligned16(a.x_hit) && aligned16(a.k_out) &&
                         ((((uintptr_t)a.valid) & 1u) == 0) &&
                         (!valid_out || (((uintptr_t)valid_out) & 1u) == 0) &&
                         (n0 % 2 == 0 || out_pitch > n0);  // odd N: the tail lane's 2nd ray lands in the padding
    if (nonconv && sys->all_conic)  // closed-form intersections only: nothing can hit an iteration cap
        HIP_TRY(hipMemsetAsync(nonconv, 0, (size_t)(mode == PRT_MODE_PATH ? sys->n_surfaces : 1) * (size_t)out_pitch, st));
    const bool vec_nc = !nonconv || (((uintptr_t)nonconv) & 1u) == 0;
    const bool vec_all = vec_in && vec_out && vec_nc;
    iso_launch L;
    L.sys = sys;
    L.n0 = n0;
    L.in_pitch = in_pitch;
    L.x0 = a.x0;
    L.k0 = a.k0;
    L.e_re = e_re;
    L.e_im = e_im;
    L.e_mode = e_mode;
    L.out_pitch = out_pitch;
    L.x_hit = a.x_hit;
    L.k_out = a.k_out;
    L.valid = a.valid;
    L.valid_out = valid_out;
    L.packed_flags = packed_flags;
    L.nonconv = nonconv;
    L.uni = uni;
    L.fu = fu;
    L.moments = moments && vec_all && !nonconv;
    L.rx = L.ry = L.rz = 0.0;
    L.partials = nullptr;
    L.st = st;
    L.redirect = a.x_img != nullptr;
    memset(&L.img, 0, sizeof L.img);
    if (L.redirect) {
        const bool img_ok = a.k_img && a.valid_img && a.img_pitch >= n0 && (a.img_pitch % 2 == 0) && aligned16(a.x_img) &&
                            aligned16(a.k_img) && ((((uintptr_t)a.valid_img) & 1u) == 0) &&
                            (!a.valid_out_img || (((uintptr_t)a.valid_out_img) & 1u) == 0);
        if (mode != PRT_MODE_PATH || !vec_all || nonconv || !img_ok)
            return fail(PRT_ERR_INVALID_ARG, "prt_trace: the image-plane redirect needs PRT_MODE_PATH, 16-B aligned rows "
                                             "everywhere (even pitches) and no nonconv array");
        L.img.x = a.x_img;
        L.img.k = a.k_img;
        L.img.valid = a.valid_img;
        L.img.valid_out = packed_flags ? nullptr : a.valid_out_img;
        L.img.pitch = a.img_pitch;
    }
    if (moments) {
        // reference point of the sums: the vertex of the last surface unless the caller names one
        const prt_surface_t *last = sys->h_table + (sys->n_surfaces - 1);
        L.rx = a.moments_ref3 ? a.moments_ref3[0] : last->g_shape[0];
        L.ry = a.moments_ref3 ? a.moments_ref3[1] : last->g_shape[1];
        L.rz = a.moments_ref3 ? a.moments_ref3[2] : last->g_shape[2];
        L.partials = a.moments_scratch_dev;
    }
    if (mode == PRT_MODE_PATH) launch_trace_iso<PRT_MODE_PATH>(L, vec_in, vec_out && vec_nc);
    else launch_trace_iso<PRT_MODE_IMAGE>(L, vec_in, vec_out && vec_nc);
    if (L.moments) {
        const unsigned nb = (unsigned)nblocks(n0, PRT_MARCH_BLOCK * 2);
        const unsigned ng = (nb + PRT_BLOCK - 1) / PRT_BLOCK;
        if (ng == 1) {
            // up to 256 march blocks (131072 rays): the final kernel adds the rows itself -- thread t takes row t, then
            // the same fixed tree the staging kernel would have used: the same bits, one launch less per call
            hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(PRT_BLOCK), 0, st, (int)nb, a.moments_scratch_dev,
                               a.moments_out7_dev);
        } else {
            double *stage = a.moments_scratch_dev + (int64_t)MOM_VALUES * nb;
            hipLaunchKernelGGL(k_moments_stage, dim3(ng), dim3(PRT_BLOCK), 0, st, (int)nb, a.moments_scratch_dev, stage);
            hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(PRT_BLOCK), 0, st, (int)ng, stage, a.moments_out7_dev);
        }
    } else if (moments) {
        // unaligned or odd-pitch buffers: the plain trace above followed by the two-kernel reduction (with
        // packed flags the selecting mask is bit 1 of the flags byte)
        const int64_t row = (mode == PRT_MODE_PATH) ? (int64_t)(sys->n_surfaces - 1) : 0;
        const uint8_t *mask = packed_flags ? a.valid + row * out_pitch : (valid_out ? valid_out + row * out_pitch : nullptr);
        if (!mask) return fail(PRT_ERR_INVALID_ARG, "prt_trace_moments: valid_out is required for unaligned buffers");
        int nbk = (int)((n0 + PRT_BLOCK * 8 - 1) / (PRT_BLOCK * 8));
        if (nbk > 2048) nbk = 2048;
        if (nbk < 1) nbk = 1;
        hipLaunchKernelGGL(k_moments_partial, dim3(nbk), dim3(PRT_BLOCK), 0, st, n0, out_pitch,
                           a.x_hit + row * 3 * out_pitch, mask, 0, L.rx, L.ry, L.rz, (const double *)nullptr, 0,
                           a.moments_scratch_dev, packed_flags ? 2 : 0xff);
        hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(PRT_BLOCK), 0, st, nbk, a.moments_scratch_dev,
                           a.moments_out7_dev);
    }
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_sizeof_trace_args(void) { return (int32_t)sizeof(prt_trace_args_t); }

int32_t prt_trace_ex(const prt_system_t *sys, const prt_trace_args_t *args) {
    if (!args) return fail(PRT_ERR_INVALID_ARG, "prt_trace_ex: null argument struct");
    if (args->struct_bytes != (int32_t)sizeof(prt_trace_args_t))
        return fail(PRT_ERR_INVALID_ARG, "prt_trace_ex: struct_bytes is not sizeof(prt_trace_args_t) of this library");
    if (args->timed_iters <= 0) return trace_launch(sys, *args);
    if (!args->ms_avg) return fail(PRT_ERR_INVALID_ARG, "prt_trace_ex: timed_iters without ms_avg");
    if (!sys) return fail(PRT_ERR_INVALID_ARG, "null system");
    PRT_ON_DEVICE(sys->device);
    hipStream_t st = (hipStream_t)args->stream;
    event_pair ev;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    HIP_TRY(hipEventRecord(ev.a, st));
    for (int it = 0; it < args->timed_iters; ++it) {
        int32_t rc = trace_launch(sys, *args);
        if (rc != PRT_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev.b, st));
    HIP_TRY(hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    *args->ms_avg = (double)ms / args->timed_iters;
    return PRT_OK;
}

static prt_trace_args_t legacy_args(int64_t n0, int64_t in_pitch, const double *x0, const double *k0,
                                    const double *e0_re, const double *e0_im, int32_t mode, int64_t out_pitch,
                                    double *x_hit, double *k_out, uint8_t *valid, uint8_t *valid_out,
                                    uint8_t *nonconv, void *stream) {
    prt_trace_args_t a = blank_args();
    a.mode = mode;
    a.n0 = n0;
    a.in_pitch = in_pitch;
    a.x0 = x0;
    a.k0 = k0;
    a.e0_re = e0_re;
    a.e0_im = e0_im;
    a.first_dir = PRT_FIRST_E;
    a.out_pitch = out_pitch;
    a.x_hit = x_hit;
    a.k_out = k_out;
    a.valid = valid;
    a.valid_out = valid_out;
    a.nonconv = nonconv;
    a.stream = stream;
    return a;
}

int32_t prt_trace(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                  const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                  int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                  uint8_t *valid_out, uint8_t *nonconv, void *stream) {
    if (n0 > 0 && !k0) return fail(PRT_ERR_INVALID_ARG, "prt_trace: null pointer (a uniform k0 is prt_trace_ex's)");
    const prt_trace_args_t a = legacy_args(n0, in_pitch, x0, k0, e0_re, e0_im, mode, out_pitch, x_hit, k_out, valid,
                                           valid_out, nonconv, stream);
    return trace_launch(sys, a);
}

// ---- one-call form: no handle for the caller to keep -----------------------
// Tables are uploaded once and kept by content (the last PRT_SEQ_CACHE of them per process): an optimiser
// loop that re-traces a changing prescription pays one table upload per distinct table.
#define PRT_SEQ_CACHE 8
struct seq_cache_entry {
    int32_t device, n_surfaces;
    prt_surface_t *table;  // host copy, the key
    prt_system *sys;
    uint64_t stamp;
};
static std::mutex g_seq_mu;
static seq_cache_entry g_seq_cache[PRT_SEQ_CACHE];
static uint64_t g_seq_clock = 0;

// caller holds g_seq_mu (cacheable tables) -- the lock is kept until the launch is enqueued, so an entry cannot be
// evicted between the lookup and the launch that uses its device table
static int32_t seq_system_locked(const prt_surface_t *table, int32_t S, int32_t device, prt_system **out, bool *owned) {
    *owned = false;
    int victim = 0;
    for (int i = 0; i < PRT_SEQ_CACHE; ++i) {
        seq_cache_entry &e = g_seq_cache[i];
        if (e.sys && e.device == device && e.n_surfaces == S &&
            memcmp(e.table, table, sizeof(prt_surface_t) * (size_t)S) == 0) {
            e.stamp = ++g_seq_clock;
            *out = e.sys;
            return PRT_OK;
        }
        if (g_seq_cache[i].stamp < g_seq_cache[victim].stamp) victim = i;
    }
    prt_system *sys = nullptr;
    int32_t rc = prt_system_create(table, S, device, &sys);
    if (rc != PRT_OK) return rc;
    prt_surface_t *copy = new (std::nothrow) prt_surface_t[S];
    if (!copy) {
        *owned = true;  // no room to remember it: the caller's call owns it
        *out = sys;
        return PRT_OK;
    }
    memcpy(copy, table, sizeof(prt_surface_t) * (size_t)S);
    seq_cache_entry &v = g_seq_cache[victim];
    if (v.sys) {
        // work of earlier calls may still use the evicted table (enqueued under this lock, so it is visible)
        device_guard guard_(v.device);
        (void)hipDeviceSynchronize();
        free_system(v.sys);
        delete[] v.table;
    }
    v.device = device;
    v.n_surfaces = S;
    v.table = copy;
    v.sys = sys;
    v.stamp = ++g_seq_clock;
    *out = sys;
    return PRT_OK;
}

int32_t prt_trace_seq(const prt_surface_t *table, int32_t n_surfaces, int64_t n, const double *x0,
                      const double *k0, const double *d0, const int64_t *ray_id, int32_t mode, double *x_hit,
                      double *k_out, uint8_t *valid, uint8_t *nonconv, int32_t device, void *stream) {
    (void)ray_id;  // outputs are dense: column i belongs to input ray i, whatever the caller calls it
    if (!table || n_surfaces <= 0 || n < 0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_seq: bad table / count");
    if (mode != PRT_MODE_PATH && mode != PRT_MODE_IMAGE) return fail(PRT_ERR_INVALID_ARG, "prt_trace_seq: bad mode");
    if (n > 0 && !k0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_seq: null pointer");
    prt_trace_args_t a = legacy_args(n, 0, x0, k0, d0, nullptr, mode, 0, x_hit, k_out, valid, nullptr, nonconv, stream);
    a.first_dir = d0 ? PRT_FIRST_DIR : PRT_FIRST_K;
    bool cacheable = true;  // records that point to host data (grid-sag splines) are not compared by content
    for (int s = 0; s < n_surfaces; ++s)
        if (table[s].aux) cacheable = false;
    prt_system *sys = nullptr;
    if (cacheable) {
        std::lock_guard<std::mutex> lock(g_seq_mu);
        bool owned = false;
        int32_t rc = seq_system_locked(table, n_surfaces, device, &sys, &owned);
        if (rc != PRT_OK) return rc;
        rc = trace_launch(sys, a);
        if (owned) {
            device_guard guard_(device);
            (void)hipStreamSynchronize((hipStream_t)stream);
            free_system(sys);
        }
        return rc;
    }
    int32_t rc = prt_system_create(table, n_surfaces, device, &sys);
    if (rc != PRT_OK) return rc;
    rc = trace_launch(sys, a);
    {
        device_guard guard_(device);
        (void)hipStreamSynchronize((hipStream_t)stream);
        free_system(sys);
    }
    return rc;
}

int32_t prt_trace_fields(const prt_system_t *sys, int64_t n0, const double *x0, const double *k0,
                         const double *e0_re, const double *e0_im, int32_t mode, double *x_hit,
                         double *k_out, double *e_out_re, double *e_out_im, uint8_t *valid,
                         uint8_t *valid_out, void *stream) {
    if (!e_out_re) return fail(PRT_ERR_INVALID_ARG, "prt_trace_fields: e_out_re is NULL");
    if (n0 > 0 && !k0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_fields: null pointer");
    prt_trace_args_t a = legacy_args(n0, 0, x0, k0, e0_re, e0_im, mode, 0, x_hit, k_out, valid, valid_out, nullptr, stream);
    a.e_out_re = e_out_re;
    a.e_out_im = e_out_im;
    return trace_launch(sys, a);
}

int64_t prt_trace_moments_scratch_doubles(int64_t n0) {
    if (n0 < 0) return 0;
    // one row per march block (2 * PRT_MARCH_BLOCK rays) + the rows of k_moments_stage
    const int64_t rows = (n0 + 2 * PRT_MARCH_BLOCK - 1) / (2 * PRT_MARCH_BLOCK);
    const int64_t fused = (int64_t)MOM_VALUES * (rows + (rows + PRT_BLOCK - 1) / PRT_BLOCK);
    const int64_t two_kernel = prt_moments_scratch_doubles(n0);
    return fused > two_kernel ? fused : two_kernel;
}

int32_t prt_trace_moments(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                          const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                          int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                          uint8_t *valid_out, const double *ref3, double *out7_dev,
                          double *scratch_dev, void *stream) {
    if (!sys || n0 < 0 || !out7_dev || !scratch_dev)
        return fail(PRT_ERR_INVALID_ARG, "prt_trace_moments: bad argument");
    if (n0 > 0 && !k0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_moments: null pointer");
    prt_trace_args_t a = legacy_args(n0, in_pitch, x0, k0, e0_re, e0_im, mode, out_pitch, x_hit, k_out, valid,
                                     valid_out, nullptr, stream);
    a.moments_ref3 = ref3;
    a.moments_out7_dev = out7_dev;
    a.moments_scratch_dev = scratch_dev;
    return trace_launch(sys, a);
}

int32_t prt_trace_timed(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                        const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                        int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                        uint8_t *valid_out, void *stream, int32_t iters, double *ms_avg) {
    if (!ms_avg || iters <= 0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_timed");
    if (n0 > 0 && !k0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_timed: null pointer");
    prt_trace_args_t a = legacy_args(n0, in_pitch, x0, k0, e0_re, e0_im, mode, out_pitch, x_hit, k_out, valid,
                                     valid_out, nullptr, stream);
    a.timed_iters = iters;
    a.ms_avg = ms_avg;
    return prt_trace_ex(sys, &a);
}

int32_t prt_propagate(const prt_system_t *sys, int32_t surface, int64_t n, const double *x,
                      const double *k, const double *dir, const double *e_re, const double *e_im,
                      int32_t use_default_e, const uint8_t *valid_in, double *x_hit,
                      uint8_t *valid, uint8_t *nonconv, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_propagate: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x || (!k && !dir) || !x_hit || !valid)
        return fail(PRT_ERR_INVALID_ARG, "prt_propagate: null pointer");
    PRT_ON_DEVICE(sys->device);
    hipLaunchKernelGGL(k_propagate, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                       (hipStream_t)stream, sys->d_table + surface, n, n, x, k, dir, e_re, e_im,
                       e_mode_of(e_re, use_default_e), valid_in, x_hit, valid, nonconv);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_interact(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit,
                     const double *k, const uint8_t *valid_in, double *k_out, double *dir_out,
                     double *e_out_re, double *e_out_im, uint8_t *valid_out, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_interact: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x_hit || !k || !k_out) return fail(PRT_ERR_INVALID_ARG, "prt_interact: null pointer");
    PRT_ON_DEVICE(sys->device);
    const prt_surface_t *rec = sys->h_table + surface;
    // Absorbing media (a complex epsilon tensor, a complex refractive index -- also on an ISOTROPIC record, also as the
    // only record of a one-surface table): the wave vectors are complex, and this entry point has neither a complex k
    // to take nor one to give back.  Dropping Im(n) silently would return real-index wave vectors and masks that
    // disagree with the fused path and with the reference.
    if (sys->complex_eps)
        return fail(PRT_ERR_UNSUPPORTED, "prt_interact: a table with absorbing media (complex epsilon / complex index) has "
                                         "complex wave vectors: prt_interact_cplx (one surface) or prt_trace_ex with "
                                         "k_out_im (the whole sequence)");
    if (rec->mat_type == PRT_MAT_ANISOTROPIC) {
        if (!dir_out) return fail(PRT_ERR_INVALID_ARG, "prt_interact: anisotropic needs dir_out");
        hipLaunchKernelGGL(k_interact_aniso, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                           (hipStream_t)stream, sys->d_table + surface, n, x_hit, k,
                           (const uint8_t *)nullptr, k_out, dir_out, e_out_re, e_out_im, valid_out);
    } else {
        hipLaunchKernelGGL(k_interact_iso, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                           (hipStream_t)stream, sys->d_table + surface, n, x_hit, k, valid_in,
                           k_out, dir_out, valid_out);
    }
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

// every pointer 16-B aligned and every pitch even: one dwordx4 access moves the two rays a thread owns
static bool rows_vectorisable(std::initializer_list<const void *> ptrs, std::initializer_list<int64_t> pitches) {
    for (const void *p : ptrs)
        if (p && (reinterpret_cast<uintptr_t>(p) & 15u)) return false;
    for (int64_t q : pitches)
        if (q & 1) return false;
    return true;
}

static int rows_shape_level(int32_t shape_type) {
    if (shape_type == PRT_SHAPE_CONIC) return PRT_SHAPES_CONIC;
    if (shape_type == PRT_SHAPE_ASPHERE) return PRT_SHAPES_ASPHERE;
    if (shape_type == PRT_SHAPE_XYPOLY || shape_type == PRT_SHAPE_BICONIC) return PRT_SHAPES_POLY;
    return PRT_SHAPES_ALL;
}

int32_t prt_propagate_rows(const prt_system_t *sys, int32_t surface, int64_t n, const double *x, int64_t x_pitch,
                           const double *k, int64_t k_pitch, const double *dir, const double *e_re, const double *e_im,
                           int32_t use_default_e, const uint8_t *valid_in, double *x_hit, int64_t out_pitch,
                           uint8_t *valid, uint8_t *nonconv, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_propagate_rows: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x || (!k && !dir) || !x_hit || !valid) return fail(PRT_ERR_INVALID_ARG, "prt_propagate_rows: null pointer");
    if (!x_pitch) x_pitch = n;
    if (!k_pitch) k_pitch = n;
    if (!out_pitch) out_pitch = n;
    if (x_pitch < n || k_pitch < n || out_pitch < n)
        return fail(PRT_ERR_INVALID_ARG, "prt_propagate_rows: a row pitch shorter than the bundle");
    PRT_ON_DEVICE(sys->device);
    const bool vec = rows_vectorisable({x, k, dir, e_re, e_im, x_hit}, {x_pitch, k_pitch, out_pitch}) &&
                     !((reinterpret_cast<uintptr_t>(valid_in) | reinterpret_cast<uintptr_t>(valid) |
                        reinterpret_cast<uintptr_t>(nonconv)) & 1u);
    const dim3 grid(nblocks((n + 1) / 2, PRT_MARCH_BLOCK)), block(PRT_MARCH_BLOCK);
    const int e_mode = e_mode_of(e_re, use_default_e);
#define PRT_LAUNCH_PROPAGATE_ROWS(VEC_, SH_)                                                                          \
    hipLaunchKernelGGL((k_propagate_rows<VEC_, SH_>), grid, block, 0, (hipStream_t)stream, sys->d_table + surface, n, x, \
                       x_pitch, k, k_pitch, dir, e_re, e_im, e_mode, valid_in, x_hit, out_pitch, valid, nonconv)
    // (the instantiation with this surface's shape code alone: a conic needs 60-odd registers, the general case 128)
    const int level = rows_shape_level(sys->h_table[surface].shape_type);
    if (!vec) PRT_LAUNCH_PROPAGATE_ROWS(false, PRT_SHAPES_ALL);
    else if (level == PRT_SHAPES_CONIC) PRT_LAUNCH_PROPAGATE_ROWS(true, PRT_SHAPES_CONIC);
    else if (level == PRT_SHAPES_ASPHERE) PRT_LAUNCH_PROPAGATE_ROWS(true, PRT_SHAPES_ASPHERE);
    else if (level == PRT_SHAPES_POLY) PRT_LAUNCH_PROPAGATE_ROWS(true, PRT_SHAPES_POLY);
    else PRT_LAUNCH_PROPAGATE_ROWS(true, PRT_SHAPES_ALL);
#undef PRT_LAUNCH_PROPAGATE_ROWS
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_interact_rows(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit, int64_t x_pitch,
                          const double *k, int64_t k_pitch, const uint8_t *valid_in, double *k_out, int64_t out_pitch,
                          double *dir_out, uint8_t *valid_out, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_interact_rows: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x_hit || !k || !k_out) return fail(PRT_ERR_INVALID_ARG, "prt_interact_rows: null pointer");
    if (sys->complex_eps || sys->h_table[surface].mat_type == PRT_MAT_ANISOTROPIC)
        return fail(PRT_ERR_UNSUPPORTED, "prt_interact_rows: isotropic, lossless media only (crystals double the rays: "
                                         "prt_interact; absorbing media: prt_interact_cplx)");
    if (!x_pitch) x_pitch = n;
    if (!k_pitch) k_pitch = n;
    if (!out_pitch) out_pitch = n;
    if (x_pitch < n || k_pitch < n || out_pitch < n)
        return fail(PRT_ERR_INVALID_ARG, "prt_interact_rows: a row pitch shorter than the bundle");
    PRT_ON_DEVICE(sys->device);
    const bool vec = rows_vectorisable({x_hit, k, k_out, dir_out}, {x_pitch, k_pitch, out_pitch}) &&
                     !((reinterpret_cast<uintptr_t>(valid_in) | reinterpret_cast<uintptr_t>(valid_out)) & 1u);
    const dim3 grid(nblocks((n + 1) / 2, PRT_MARCH_BLOCK)), block(PRT_MARCH_BLOCK);
#define PRT_LAUNCH_INTERACT_ROWS(VEC_, SH_)                                                                            \
    hipLaunchKernelGGL((k_interact_iso_rows<VEC_, SH_>), grid, block, 0, (hipStream_t)stream, sys->d_table + surface, n, \
                       x_hit, x_pitch, k, k_pitch, valid_in, k_out, out_pitch, dir_out, valid_out)
    if (!vec) PRT_LAUNCH_INTERACT_ROWS(false, PRT_SHAPES_ALL);
    else if (sys->h_table[surface].shape_type == PRT_SHAPE_CONIC) PRT_LAUNCH_INTERACT_ROWS(true, PRT_SHAPES_CONIC);
    else PRT_LAUNCH_INTERACT_ROWS(true, PRT_SHAPES_ALL);
#undef PRT_LAUNCH_INTERACT_ROWS
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_surface_step_rows(const prt_system_t *sys, int32_t surface, int64_t n, const double *x, int64_t x_pitch,
                              const double *k, int64_t k_pitch, const double *dir, const double *e_re, const double *e_im,
                              int32_t use_default_e, const uint8_t *valid_in, double *x_hit, double *k_out,
                              int64_t out_pitch, uint8_t *valid, uint8_t *valid_out, uint8_t *nonconv, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_surface_step_rows: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x || !k || !x_hit || !k_out || !valid || !valid_out)
        return fail(PRT_ERR_INVALID_ARG, "prt_surface_step_rows: null pointer");
    if (sys->complex_eps || sys->h_table[surface].mat_type == PRT_MAT_ANISOTROPIC)
        return fail(PRT_ERR_UNSUPPORTED, "prt_surface_step_rows: isotropic, lossless media only (crystals double the rays: "
                                         "prt_propagate + prt_interact; absorbing media: prt_interact_cplx)");
    if (!x_pitch) x_pitch = n;
    if (!k_pitch) k_pitch = n;
    if (!out_pitch) out_pitch = n;
    if (x_pitch < n || k_pitch < n || out_pitch < n)
        return fail(PRT_ERR_INVALID_ARG, "prt_surface_step_rows: a row pitch shorter than the bundle");
    PRT_ON_DEVICE(sys->device);
    const bool vec = rows_vectorisable({x, k, dir, e_re, e_im, x_hit, k_out}, {x_pitch, k_pitch, out_pitch}) &&
                     !((reinterpret_cast<uintptr_t>(valid_in) | reinterpret_cast<uintptr_t>(valid) |
                        reinterpret_cast<uintptr_t>(valid_out) | reinterpret_cast<uintptr_t>(nonconv)) & 1u);
    const dim3 grid(nblocks((n + 1) / 2, PRT_MARCH_BLOCK)), block(PRT_MARCH_BLOCK);
    const int e_mode = e_mode_of(e_re, use_default_e);
#define PRT_LAUNCH_STEP_ROWS(VEC_, SH_)                                                                                  \
    hipLaunchKernelGGL((k_surface_step_rows<VEC_, SH_>), grid, block, 0, (hipStream_t)stream, sys->d_table + surface, n, \
                       x, x_pitch, k, k_pitch, dir, e_re, e_im, e_mode, valid_in, x_hit, k_out, out_pitch, valid,        \
                       valid_out, nonconv)
    const int level = rows_shape_level(sys->h_table[surface].shape_type);
    if (!vec) PRT_LAUNCH_STEP_ROWS(false, PRT_SHAPES_ALL);
    else if (level == PRT_SHAPES_CONIC) PRT_LAUNCH_STEP_ROWS(true, PRT_SHAPES_CONIC);
    else if (level == PRT_SHAPES_ASPHERE) PRT_LAUNCH_STEP_ROWS(true, PRT_SHAPES_ASPHERE);
    else if (level == PRT_SHAPES_POLY) PRT_LAUNCH_STEP_ROWS(true, PRT_SHAPES_POLY);
    else PRT_LAUNCH_STEP_ROWS(true, PRT_SHAPES_ALL);
#undef PRT_LAUNCH_STEP_ROWS
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_interact_cplx(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit,
                          const double *k_re, const double *k_im, const uint8_t *valid_in, double *k_out_re,
                          double *k_out_im, double *dir_out, double *e_out_re, double *e_out_im, uint8_t *valid_out,
                          void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_interact_cplx: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x_hit || !k_re || !k_out_re || !k_out_im) return fail(PRT_ERR_INVALID_ARG, "prt_interact_cplx: null pointer");
    PRT_ON_DEVICE(sys->device);
    const prt_surface_t *rec = sys->h_table + surface;
    if (rec->mat_type == PRT_MAT_ANISOTROPIC) {
        if (!dir_out) return fail(PRT_ERR_INVALID_ARG, "prt_interact_cplx: anisotropic needs dir_out");
        // (every ray leaves a crystal interface, like in k_interact_aniso: AnisotropicMaterial.refract restarts validity)
        hipLaunchKernelGGL(k_interact_aniso_cplx, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, (hipStream_t)stream,
                           sys->d_table + surface, sys->d_eps_im ? sys->d_eps_im + (size_t)surface * 9 : (const double *)nullptr,
                           n, x_hit, k_re, k_im, (const uint8_t *)nullptr, k_out_re, k_out_im, dir_out, e_out_re, e_out_im,
                           valid_out);
    } else {
        if (rec->eps_im[0] != 0.0 && rec->interaction == PRT_MIRROR)
            return fail(PRT_ERR_UNSUPPORTED, "prt_interact_cplx: a mirror inside an absorbing isotropic medium");
        hipLaunchKernelGGL(k_interact_iso_cplx, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, (hipStream_t)stream,
                           sys->d_table + surface, rec->eps_im[0], n, x_hit, k_re, k_im, valid_in, k_out_re, k_out_im,
                           valid_out);
    }
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_shape_eval(const prt_system_t *sys, int32_t surface, int64_t n, const double *x,
                       const double *y, double *sag, double *grad, void *stream) {
    PRT_SYS_USABLE(sys);
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_shape_eval: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x || !y) return fail(PRT_ERR_INVALID_ARG, "prt_shape_eval: null pointer");
    PRT_ON_DEVICE(sys->device);
    hipLaunchKernelGGL(k_shape_eval, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                       (hipStream_t)stream, sys->d_table + surface, n, x, y, sag, grad);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_efield_perp(int32_t device, int64_t n, const double *k, double *e_out, void *stream) {
    if (n < 0) return fail(PRT_ERR_INVALID_ARG, "prt_efield_perp: negative count");
    if (n == 0) return PRT_OK;
    if (!k || !e_out) return fail(PRT_ERR_INVALID_ARG, "prt_efield_perp: null pointer");
    PRT_ON_DEVICE(device);
    hipLaunchKernelGGL(k_efield_perp, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                       (hipStream_t)stream, n, k, e_out);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

static void rect_grid_params(int64_t nray, int64_t *n_per_dim, double *start, double *step,
                             double *stop) {
    // nPerDim = int(round(sqrt(nray*4/pi))); dx = 1/nPerDim; linspace(-1+.25dx, 1-.25dx, nPerDim)
    const double v = sqrt((double)nray * 4.0 / 3.14159265358979323846);
    int64_t n = (int64_t)nearbyint(v);  // Python round(): half to even, like nearbyint
    if (n < 1) n = 1;
    const double dx = 1.0 / (double)n;
    *start = -1.0 + 0.25 * dx;
    *stop = 1.0 - 0.25 * dx;
    *step = (n > 1) ? (*stop - *start) / (double)(n - 1) : 0.0;
    *n_per_dim = n;
}

// mask + block counts + scan for the raster; returns device scratch (caller frees)
static int32_t rect_grid_scan(int64_t nray, hipStream_t st, int64_t *n_per_dim, double *start,
                              double *step, double *stop, uint8_t **d_mask, int64_t **d_sums,
                              int64_t *nb_out, int64_t *total_out) {
    rect_grid_params(nray, n_per_dim, start, step, stop);
    const int64_t n = *n_per_dim, pts = n * n;
    const int64_t nb = (pts + CMP_TILE - 1) / CMP_TILE;
    HIP_TRY(hipMallocAsync((void **)d_mask, (size_t)pts, st));
    HIP_TRY(hipMallocAsync((void **)d_sums, sizeof(int64_t) * (size_t)(nb + 1), st));
    hipLaunchKernelGGL(k_rectgrid_mask, dim3(nblocks(pts, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, n, *start,
                       *step, *stop, *d_mask);
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nb), dim3(PRT_BLOCK), 0, st, *d_mask, pts, *d_sums);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(PRT_BLOCK), 0, st, *d_sums, nb);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(total_out, *d_sums + nb, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *nb_out = nb;
    return PRT_OK;
}

int32_t prt_rect_grid_count(int32_t device, int64_t nray, int64_t *n_per_dim, int64_t *n_in_disk,
                            void *stream) {
    if (nray < 1 || !n_per_dim || !n_in_disk) return fail(PRT_ERR_INVALID_ARG, "prt_rect_grid_count");
    PRT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    double start, step, stop;
    uint8_t *d_mask = nullptr;
    int64_t *d_sums = nullptr;
    int64_t nb = 0;
    int32_t rc = rect_grid_scan(nray, st, n_per_dim, &start, &step, &stop, &d_mask, &d_sums, &nb, n_in_disk);
    if (d_mask) (void)hipFreeAsync(d_mask, st);
    if (d_sums) (void)hipFreeAsync(d_sums, st);
    return rc;
}

int32_t prt_collimated_bundle(int32_t device, int64_t nray, int64_t lo, int64_t hi,
                              const prt_collimated_t *prm, int64_t pitch, double *x_out,
                              double *k_out, double *e_out, void *stream) {
    if (nray < 1 || !prm || lo < 0 || hi < lo) return fail(PRT_ERR_INVALID_ARG, "prt_collimated_bundle");
    if (hi == lo) return PRT_OK;
    if (!x_out) return fail(PRT_ERR_INVALID_ARG, "prt_collimated_bundle: null pointer");
    if (pitch == 0) pitch = hi - lo;
    if (pitch < hi - lo) return fail(PRT_ERR_INVALID_ARG, "prt_collimated_bundle: pitch < hi - lo");
    PRT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    double start, step, stop;
    uint8_t *d_mask = nullptr;
    int64_t *d_sums = nullptr;
    int64_t nb = 0, n = 0, total = 0;
    int32_t rc = rect_grid_scan(nray, st, &n, &start, &step, &stop, &d_mask, &d_sums, &nb, &total);
    if (rc == PRT_OK && hi > total) rc = fail(PRT_ERR_INVALID_ARG, "prt_collimated_bundle: hi beyond the raster");
    if (rc == PRT_OK) {
        collimated_params cp;
        cp.radius = prm->radius;
        cp.startx = prm->startx;
        cp.starty = prm->starty;
        cp.startz = prm->startz;
        for (int q = 0; q < 3; ++q) {
            cp.k[q] = prm->k[q];
            cp.e[q] = prm->e[q];
        }
        hipLaunchKernelGGL(k_rectgrid_scatter, dim3((unsigned)nb), dim3(PRT_BLOCK), 0, st, d_mask, n, start,
                           step, stop, d_sums, lo, hi, cp, pitch, x_out, k_out, e_out);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail(PRT_ERR_DEVICE, "k_rectgrid_scatter", e);
    }
    if (d_mask) (void)hipFreeAsync(d_mask, st);
    if (d_sums) (void)hipFreeAsync(d_sums, st);
    return rc;
}

// device copies of a raster's tables + mask + block offsets; released on every exit path
struct raster_scan {
    hipStream_t st;
    double *d_tab = nullptr;
    uint8_t *d_mask = nullptr;
    int64_t *d_sums = nullptr;
    int64_t nb = 0, total = 0;
    raster_tables t;
    explicit raster_scan(hipStream_t s) : st(s) {}
    ~raster_scan() {
        if (d_tab) (void)hipFreeAsync(d_tab, st);
        if (d_mask) (void)hipFreeAsync(d_mask, st);
        if (d_sums) (void)hipFreeAsync(d_sums, st);
    }
    int32_t run(const prt_raster_t *r) {
        if (!r || r->ni < 1 || r->nj < 1 || !r->xa || !r->xb || !r->ya || !r->yb)
            return fail(PRT_ERR_INVALID_ARG, "raster: null table / empty raster");
        if (r->ni > ((int64_t)1 << 40) / r->nj) return fail(PRT_ERR_INVALID_ARG, "raster: too many points");
        const int64_t pts = r->ni * r->nj;
        nb = (pts + CMP_TILE - 1) / CMP_TILE;
        const size_t nt = (size_t)(2 * r->ni + 2 * r->nj);
        HIP_TRY(hipMallocAsync((void **)&d_tab, sizeof(double) * nt, st));
        HIP_TRY(hipMallocAsync((void **)&d_mask, (size_t)pts, st));
        HIP_TRY(hipMallocAsync((void **)&d_sums, sizeof(int64_t) * (size_t)(nb + 1), st));
        t.ni = r->ni;
        t.nj = r->nj;
        t.xa = d_tab;
        t.ya = d_tab + r->nj;
        t.xb = d_tab + 2 * r->nj;
        t.yb = d_tab + 2 * r->nj + r->ni;
        HIP_TRY(hipMemcpyAsync((void *)t.xa, r->xa, sizeof(double) * r->nj, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)t.ya, r->ya, sizeof(double) * r->nj, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)t.xb, r->xb, sizeof(double) * r->ni, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)t.yb, r->yb, sizeof(double) * r->ni, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_raster_mask, dim3(nblocks(pts, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, t, r->clip, d_mask);
        hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nb), dim3(PRT_BLOCK), 0, st, d_mask, pts, d_sums);
        hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(PRT_BLOCK), 0, st, d_sums, nb);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&total, d_sums + nb, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));   // the count is needed on the host; the caller's tables may go
        return PRT_OK;
    }
};

int32_t prt_raster_count(int32_t device, const prt_raster_t *raster, int64_t *n_points, void *stream) {
    if (!n_points) return fail(PRT_ERR_INVALID_ARG, "prt_raster_count: null output");
    PRT_ON_DEVICE(device);
    raster_scan scan((hipStream_t)stream);
    int32_t rc = scan.run(raster);
    if (rc == PRT_OK) *n_points = scan.total;
    return rc;
}

int32_t prt_raster_bundle(int32_t device, const prt_raster_t *raster, int64_t lo, int64_t hi,
                          const prt_bundle_t *prm, int64_t pitch, double *x_out, double *k_out,
                          double *e_out, double *p_out, void *stream) {
    if (!prm || lo < 0 || hi < lo || prm->kind < 0 || prm->kind > 1)
        return fail(PRT_ERR_INVALID_ARG, "prt_raster_bundle: bad argument");
    if (hi == lo) return PRT_OK;
    if (!x_out || (!k_out && prm->kind != 0)) return fail(PRT_ERR_INVALID_ARG, "prt_raster_bundle: null pointer");
    if (pitch == 0) pitch = hi - lo;
    if (pitch < hi - lo) return fail(PRT_ERR_INVALID_ARG, "prt_raster_bundle: pitch < hi - lo");
    PRT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    raster_scan scan(st);
    int32_t rc = scan.run(raster);
    if (rc != PRT_OK) return rc;
    if (hi > scan.total) return fail(PRT_ERR_INVALID_ARG, "prt_raster_bundle: hi beyond the raster");
    bundle_params bp;
    bp.kind = prm->kind;
    bp.radius = prm->radius;
    bp.startx = prm->start[0];
    bp.starty = prm->start[1];
    bp.startz = prm->start[2];
    bp.anglex = prm->anglex;
    bp.angley = prm->angley;
    bp.index = prm->index;
    for (int q = 0; q < 3; ++q) {
        bp.k[q] = prm->k[q];
        bp.e[q] = prm->e[q];
    }
    hipLaunchKernelGGL(k_raster_bundle, dim3((unsigned)scan.nb), dim3(PRT_BLOCK), 0, st, scan.d_mask, scan.t,
                       scan.d_sums, lo, hi, bp, pitch, x_out, k_out, e_out, p_out);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_poynting_dir(int32_t device, int64_t n, const double *k, const double *e_re,
                         const double *e_im, int32_t use_default_e, double *d_out, void *stream) {
    if (n < 0) return fail(PRT_ERR_INVALID_ARG, "prt_poynting_dir: negative count");
    if (n == 0) return PRT_OK;
    if (!k || !d_out) return fail(PRT_ERR_INVALID_ARG, "prt_poynting_dir: null pointer");
    PRT_ON_DEVICE(device);
    hipLaunchKernelGGL(k_poynting_dir, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                       (hipStream_t)stream, n, k, e_re, e_im, e_mode_of(e_re, use_default_e), d_out);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_path_sums(int32_t device, int32_t n_points, int64_t n, const double *const *xs,
                      const double *const *ks, int32_t mode, double *out, void *stream) {
    if (n < 0 || n_points < 1 || n_points > 64 || mode < 0 || mode > 1)
        return fail(PRT_ERR_INVALID_ARG, "prt_path_sums: bad argument");
    if (n == 0) return PRT_OK;
    if (!xs || !out || (mode == 1 && !ks)) return fail(PRT_ERR_INVALID_ARG, "prt_path_sums: null pointer");
    PRT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const double **d_tab = nullptr;
    stream_scratch scratch_(st);
    HIP_TRY(scratch_.get(&d_tab, sizeof(void *) * 2 * n_points));
    HIP_TRY(hipMemcpyAsync((void *)d_tab, xs, sizeof(void *) * n_points, hipMemcpyHostToDevice, st));
    if (ks)
        HIP_TRY(hipMemcpyAsync((void *)(d_tab + n_points), ks, sizeof(void *) * n_points,
                               hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_path_sums, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, n_points, n,
                       (const double *const *)d_tab, (const double *const *)(d_tab + n_points), mode, out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // the host pointer tables must outlive the copies
    return PRT_OK;
}

int32_t prt_bundle_moments(int32_t device, int64_t n, int64_t pitch, const double *x,
                           const uint8_t *mask, int32_t mode, const double *ref, double *out7,
                           void *stream) {
    if (n < 0 || !out7 || mode < 0 || mode > 2)
        return fail(PRT_ERR_INVALID_ARG, "prt_bundle_moments: bad argument");
    for (int q = 0; q < MOM_VALUES; ++q) out7[q] = 0.0;
    if (n == 0) return PRT_OK;
    if (!x) return fail(PRT_ERR_INVALID_ARG, "prt_bundle_moments: null pointer");
    if (pitch == 0) pitch = n;
    if (pitch < n) return fail(PRT_ERR_INVALID_ARG, "prt_bundle_moments: pitch < n");
    PRT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)((n + PRT_BLOCK * 8 - 1) / (PRT_BLOCK * 8));
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    double *scratch = nullptr;
    stream_scratch scratch_(st);
    HIP_TRY(scratch_.get(&scratch, sizeof(double) * MOM_VALUES * (nb + 1)));
    const double rx = ref ? ref[0] : 0.0, ry = ref ? ref[1] : 0.0, rz = ref ? ref[2] : 0.0;
    hipLaunchKernelGGL(k_moments_partial, dim3(nb), dim3(PRT_BLOCK), 0, st, n, pitch, x, mask, mode, rx,
                       ry, rz, (const double *)nullptr, 0, scratch);
    hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(PRT_BLOCK), 0, st, nb, scratch,
                       scratch + (int64_t)nb * MOM_VALUES);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out7, scratch + (int64_t)nb * MOM_VALUES, sizeof(double) * MOM_VALUES,
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PRT_OK;
}

int64_t prt_moments_scratch_doubles(int64_t n) {
    (void)n;
    return (int64_t)MOM_VALUES * 2048;
}

int32_t prt_bundle_moments_async(int32_t device, int64_t n, int64_t pitch, const double *x,
                                 const uint8_t *mask, int32_t mode, const double *ref_dev,
                                 int32_t ref_kind, double *out7_dev, double *scratch_dev,
                                 void *stream) {
    if (n < 0 || !out7_dev || !scratch_dev || mode < 0 || mode > 2 || ref_kind < 0 || ref_kind > 2 ||
        (ref_kind != 0 && !ref_dev))
        return fail(PRT_ERR_INVALID_ARG, "prt_bundle_moments_async: bad argument");
    if (n > 0 && !x) return fail(PRT_ERR_INVALID_ARG, "prt_bundle_moments_async: null pointer");
    if (pitch == 0) pitch = n;
    if (pitch < n) return fail(PRT_ERR_INVALID_ARG, "prt_bundle_moments_async: pitch < n");
    PRT_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)((n + PRT_BLOCK * 8 - 1) / (PRT_BLOCK * 8));
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_moments_partial, dim3(nb), dim3(PRT_BLOCK), 0, st, n, pitch, x, mask, mode, 0.0,
                       0.0, 0.0, ref_dev, ref_kind, scratch_dev);
    hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(PRT_BLOCK), 0, st, nb, scratch_dev, out7_dev);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int64_t prt_compact_scratch_bytes(int64_t n) {
    if (n < 0) return 0;
    const int64_t nb = (n + CMP_TILE - 1) / CMP_TILE;
    // block sums (+1 total) and the two pointer tables (up to PRT_COMPACT_MAX_ROWS arrays each)
    return (nb + 1) * (int64_t)sizeof(int64_t) + 2 * PRT_COMPACT_MAX_ROWS * (int64_t)sizeof(void *) + 64;
}

int32_t prt_compact(int64_t n, const uint8_t *mask, int32_t n_arrays, const double *const *src,
                    double *const *dst, const int64_t *id_src, int64_t *id_dst,
                    const uint8_t *u8_src, uint8_t *u8_dst, void *scratch, int64_t *n_kept,
                    void *stream) {
    if (n < 0 || (n > 0 && !mask) || n_arrays < 0 || n_arrays > PRT_COMPACT_MAX_ROWS || (n_arrays && (!src || !dst)) ||
        (n > 0 && !scratch) || !n_kept || (id_src && !id_dst) || (u8_src && !u8_dst))
        return fail(PRT_ERR_INVALID_ARG, "prt_compact: bad argument");
    *n_kept = 0;
    if (n == 0) return PRT_OK;
    // no device argument: the arrays say where they live
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, mask) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(PRT_ERR_INVALID_ARG, "prt_compact: mask is not a device pointer");
    }
    PRT_ON_DEVICE(attr.device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (n + CMP_TILE - 1) / CMP_TILE;
    int64_t *sums = (int64_t *)scratch;
    uintptr_t pbase = ((uintptr_t)(sums + nb + 1) + 15u) & ~(uintptr_t)15u;
    const double **d_src = (const double **)pbase;
    double **d_dst = (double **)(pbase + PRT_COMPACT_MAX_ROWS * sizeof(void *));
    if (n_arrays) {
        HIP_TRY(hipMemcpyAsync((void *)d_src, src, sizeof(void *) * n_arrays, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void *)d_dst, dst, sizeof(void *) * n_arrays, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_compact_count, dim3((unsigned)nb), dim3(PRT_BLOCK), 0, st, mask, n, sums);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(PRT_BLOCK), 0, st, sums, nb);
    hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)nb), dim3(PRT_BLOCK), 0, st, mask, n, sums,
                       n_arrays, (const double *const *)d_src, (double *const *)d_dst, id_src, id_dst,
                       u8_src, u8_dst);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(n_kept, sums + nb, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PRT_OK;
}

}  // extern "C"
