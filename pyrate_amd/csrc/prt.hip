// prt.hip -- kernels + C ABI (include/prt.h) of the gfx950 sequential raytrace engine.
//
// Kernels
//   k_trace_iso<RPT,MODE>  whole isotropic sequence in one launch: a thread loads its
//                          ray(s) once (x0,k0: 48 B/ray), marches all S surfaces with
//                          the state in VGPRs and streams out hit point / outgoing k /
//                          valid per surface (49 B/ray/surface).  HBM-bound by design:
//                          algorithmic bytes = 48 N (+E0) + 50 N S.
//   k_propagate            one Material.propagate (intersect + aperture)
//   k_interact_iso         one IsotropicMaterial.refract / reflect
//   k_interact_aniso       one AnisotropicMaterial.refract / reflect (N -> 2N rays)
//   k_shape_eval           Shape.getSag / getGrad
//   k_compact_*            order-preserving compaction by mask
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC prt.hip -o libprt.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <new>
#include "prt_device.h"
#include "prt_aniso.h"

#define PRT_BLOCK 256
#define CMP_ITEMS 4  // mask bytes per thread in the compaction / raster kernels
#define CMP_TILE (PRT_BLOCK * CMP_ITEMS)

// ---------------------------------------------------------------------------
// host-side state
// ---------------------------------------------------------------------------
struct prt_system {
    int32_t device;
    int32_t n_surfaces;
    int32_t all_isotropic;
    int32_t all_conic;
    prt_surface_t *d_table;  // device copy
    prt_surface_t *h_table;  // host copy (for dispatch decisions)
};

static thread_local char g_err[512] = "";

static int32_t fail(int32_t code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess)
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

#define HIP_TRY(call)                                                   \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return fail(PRT_ERR_DEVICE, #call, e_);   \
    } while (0)

// ---------------------------------------------------------------------------
// ray load / store helpers for the fused march.  Arrays are component-major rows
// with a row PITCH (in elements): element (row r, ray i) lives at r*pitch + i.
// A thread owns the two adjacent rays i, i+1.  With an even pitch and a 16-byte
// aligned base every row is 16-B aligned and one dwordx4 access moves both rays
// (VEC = true); otherwise two 8-B accesses with a tail guard.
//
// Row alignment decides the achievable HBM WRITE bandwidth: with rows that do not
// start on a 128-B line a wave's 1-KiB store is split over partial lines and the
// 72-stream path write drops from ~6.3 to ~3.9-4.7 TB/s (measured, DESIGN.md
// "row pitch").  prt_recommended_pitch() rounds the pitch to 4 KiB.
// ---------------------------------------------------------------------------
typedef double prt_double2 __attribute__((ext_vector_type(2)));

template <bool VEC>
struct rayio {
    // second = false: ray i+1 does not exist (odd N tail) -> duplicate ray i
    static PRT_DEV void load(const double *__restrict__ a, int64_t pitch, int64_t i, bool second,
                             vec3 v[2]) {
        if (VEC) {
            const prt_double2 x = *reinterpret_cast<const prt_double2 *>(a + i);
            const prt_double2 y = *reinterpret_cast<const prt_double2 *>(a + pitch + i);
            const prt_double2 z = *reinterpret_cast<const prt_double2 *>(a + 2 * pitch + i);
            v[0] = v3(x.x, y.x, z.x);
            v[1] = v3(x.y, y.y, z.y);
        } else {
            v[0] = v3(a[i], a[pitch + i], a[2 * pitch + i]);
            v[1] = second ? v3(a[i + 1], a[pitch + i + 1], a[2 * pitch + i + 1]) : v[0];
        }
    }
    static PRT_DEV void store(double *__restrict__ a, int64_t pitch, int64_t i, bool second,
                              const vec3 v[2]) {
        if (VEC) {
            *reinterpret_cast<prt_double2 *>(a + i) = prt_double2{v[0].x, v[1].x};
            *reinterpret_cast<prt_double2 *>(a + pitch + i) = prt_double2{v[0].y, v[1].y};
            *reinterpret_cast<prt_double2 *>(a + 2 * pitch + i) = prt_double2{v[0].z, v[1].z};
        } else {
            a[i] = v[0].x;
            a[pitch + i] = v[0].y;
            a[2 * pitch + i] = v[0].z;
            if (second) {
                a[i + 1] = v[1].x;
                a[pitch + i + 1] = v[1].y;
                a[2 * pitch + i + 1] = v[1].z;
            }
        }
    }
    static PRT_DEV void store_mask(uint8_t *__restrict__ m, int64_t i, bool second, const bool b[2]) {
        if (VEC) {
            *reinterpret_cast<uint16_t *>(m + i) = (uint16_t)((b[0] ? 1u : 0u) | (b[1] ? 0x100u : 0u));
        } else {
            m[i] = b[0] ? 1 : 0;
            if (second) m[i + 1] = b[1] ? 1 : 0;
        }
    }
};

// first-segment direction selector
//   e_mode 0: d = k/|k|        1: E = (0,1,0) (ray.py:71-73)     2: E given (re [, im])
template <bool VEC>
PRT_DEV void first_direction(int e_mode, const double *__restrict__ e_re,
                             const double *__restrict__ e_im, int64_t pitch, int64_t i, bool second,
                             const vec3 k[2], vec3 d[2]) {
    if (e_mode == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) d[r] = normalized(k[r]);
    } else if (e_mode == 1) {
#pragma unroll
        for (int r = 0; r < 2; ++r) d[r] = poynting_dir(k[r], v3(0, 1, 0), v3(0, 0, 0));
    } else {
        vec3 er[2], ei[2];
        rayio<VEC>::load(e_re, pitch, i, second, er);
        if (e_im) {
            rayio<VEC>::load(e_im, pitch, i, second, ei);
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) ei[r] = v3(0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) d[r] = poynting_dir(k[r], er[r], ei[r]);
    }
}

// ---------------------------------------------------------------------------
// fused isotropic march: OpticalElement.seqtrace's loop (optical_element.py:336-375)
// ---------------------------------------------------------------------------
// LDS_TAB = true is the measured alternative of DESIGN.md ("surface table placement"): the block
// first copies the table into LDS and the march reads the records from there (ds_read broadcast
// into VGPRs) instead of through the scalar cache (s_load into SGPRs).  Kept only for that A/B
// (PRT_LDS_TABLE=1); it is slower and uses more VGPRs.
#define PRT_LDS_TAB_MAX 16
// EXPLICIT = false: the host guarantees that every shape of the table is a Conic, and the
// Newton / polynomial code of the explicit shapes is compiled out (fewer VGPRs: one more wave
// per SIMD for the all-conic systems such as the double Gauss).
template <int MODE, bool VEC_IN, bool VEC_OUT, bool EXPLICIT = true, bool LDS_TAB = false>
__global__ __launch_bounds__(PRT_BLOCK) void k_trace_iso(
    const prt_surface_t *__restrict__ tab_g, int32_t S, int64_t N, int64_t in_pitch,
    const double *__restrict__ x0, const double *__restrict__ k0, const double *__restrict__ e_re,
    const double *__restrict__ e_im, int32_t e_mode, int64_t out_pitch,
    double *__restrict__ xh_out, double *__restrict__ k_out, uint8_t *__restrict__ valid_out_hit,
    uint8_t *__restrict__ valid_out_refr) {
    const prt_surface_t *__restrict__ tab = tab_g;
    if (LDS_TAB) {
        __shared__ prt_surface_t lds_tab[PRT_LDS_TAB_MAX];
        const int words = S * (int)(sizeof(prt_surface_t) / 8);
        const double *src = reinterpret_cast<const double *>(tab_g);
        double *dst = reinterpret_cast<double *>(lds_tab);
        for (int w = threadIdx.x; w < words; w += PRT_BLOCK) dst[w] = src[w];
        __syncthreads();
        tab = lds_tab;
    }
    const int64_t i = ((int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x) * 2;
    if (i >= N) return;
    const bool second = (i + 1 < N);

    vec3 x[2], k[2], d[2];
    bool valid[2] = {true, true};
    rayio<VEC_IN>::load(x0, in_pitch, i, second, x);
    rayio<VEC_IN>::load(k0, in_pitch, i, second, k);
    first_direction<VEC_IN>(e_mode, e_re, e_im, in_pitch, i, second, k, d);
    double d2 = 1.0;  // |d|^2: unit Poynting direction on the first segment

    for (int32_t s = 0; s < S; ++s) {
        const prt_surface_t *__restrict__ sf = tab + s;
        bool vhit[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            vec3 xh, p, g;
            double g2;
            propagate_step<EXPLICIT>(sf, x[r], d[r], d2, xh, p, g, g2, valid[r]);
            vhit[r] = valid[r];
            const vec3 n = normal_from_grad<EXPLICIT>(sf, g, g2);
            interact_isotropic(sf, n, k[r], valid[r]);
            x[r] = xh;
            // after an isotropic interaction E is perpendicular to k, so the Poynting
            // direction (ray.py:136-152) is parallel to k; the next intersection takes the
            // unnormalised k with |k|^2 = n_after^2 (conic_t / explicit_t are homogeneous in d)
            d[r] = k[r];
        }
        d2 = sf->n_after * sf->n_after;

        if (MODE == PRT_MODE_PATH || s == S - 1) {
            const int64_t so = (MODE == PRT_MODE_PATH) ? (int64_t)s : 0;
            rayio<VEC_OUT>::store(xh_out + so * 3 * out_pitch, out_pitch, i, second, x);
            rayio<VEC_OUT>::store(k_out + so * 3 * out_pitch, out_pitch, i, second, k);
            rayio<VEC_OUT>::store_mask(valid_out_hit + so * out_pitch, i, second, vhit);
            if (valid_out_refr)
                rayio<VEC_OUT>::store_mask(valid_out_refr + so * out_pitch, i, second, valid);
        }
    }
}

// ---------------------------------------------------------------------------
// fused march through tables that contain anisotropic media (ray doubling).  Thread i owns
// input ray i and ALL its descendants: with A anisotropic interfaces there are 2^A leaves;
// leaf L (bit j = which of the two transmitted solutions is followed at the j-th crystal
// interface) is traced from the start, so no per-ray stack is needed (recomputation instead
// of 2^A live states; 20 instead of 12 surface steps for the doublet of config 4, but one
// launch, no intermediate arrays, no direction buffers).  A prefix shared by several leaves
// is WRITTEN only by the leaf whose remaining bits are zero.  Outputs use the concatenated
// layout of include/prt.h (rays of a split bundle stacked [sol2, sol3] like np.hstack,
// material_anisotropic.py:89): at a level with a doublings, leaf L sits at i + N (L mod 2^a).
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(PRT_BLOCK) void k_trace_general(
    const prt_surface_t *__restrict__ tab, int32_t S, int32_t A, int64_t N,
    const double *__restrict__ x0, const double *__restrict__ k0, const double *__restrict__ e_re,
    const double *__restrict__ e_im, int32_t e_mode, double *__restrict__ xh_out,
    double *__restrict__ k_out, uint8_t *__restrict__ valid_out_hit,
    uint8_t *__restrict__ valid_out_refr) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xs = v3(x0[i], x0[N + i], x0[2 * N + i]);
    const vec3 ks = v3(k0[i], k0[N + i], k0[2 * N + i]);
    vec3 ds;
    {
        vec3 kk[2] = {ks, ks};
        vec3 dd[2];
        first_direction<false>(e_mode, e_re, e_im, N, i, false, kk, dd);
        ds = dd[0];
    }
    const int64_t leaves = (int64_t)1 << A;
    for (int64_t L = 0; L < leaves; ++L) {
        vec3 x = xs, k = ks, d = ds;
        double d2 = 1.0;
        bool valid = true;  // cumulative mask carried into the next propagate
        int a = 0;          // doublings so far
        int64_t off_in = 0, off_out = 0;
        for (int32_t s = 0; s < S; ++s) {
            const prt_surface_t *__restrict__ sf = tab + s;
            const int64_t n_in = N << a;
            const int64_t idx_in = i + N * (L & (((int64_t)1 << a) - 1));
            const bool alive = valid;
            vec3 xh, p, g;
            double g2;
            propagate_step(sf, x, d, d2, xh, p, g, g2, valid);
            const bool last = (s == S - 1);
            if ((L >> a) == 0 && (MODE == PRT_MODE_PATH || last)) {
                double *xo = xh_out + ((MODE == PRT_MODE_PATH) ? 3 * off_in : 0);
                xo[idx_in] = xh.x;
                xo[n_in + idx_in] = xh.y;
                xo[2 * n_in + idx_in] = xh.z;
                valid_out_hit[((MODE == PRT_MODE_PATH) ? off_in : 0) + idx_in] = valid ? 1 : 0;
            }
            int a_out = a;
            if (sf->mat_type == PRT_MAT_ANISOTROPIC) {
                aniso_solution sol[2];
                interact_anisotropic(sf, p, k, sol);
                const bool second = ((L >> a) & 1) != 0;
                k = second ? sol[1].k : sol[0].k;
                d = second ? sol[1].d : sol[0].d;
                d2 = 1.0;
                valid = alive;  // no validity filtering at a crystal interface (ray.py:68)
                a_out = a + 1;
            } else {
                const vec3 n = normal_from_grad(sf, g, g2);
                interact_isotropic(sf, n, k, valid);
                d = k;
                d2 = sf->n_after * sf->n_after;
            }
            const int64_t n_out = N << a_out;
            if ((L >> a_out) == 0 && (MODE == PRT_MODE_PATH || last)) {
                const int64_t idx_out = i + N * (L & (((int64_t)1 << a_out) - 1));
                double *ko = k_out + ((MODE == PRT_MODE_PATH) ? 3 * off_out : 0);
                ko[idx_out] = k.x;
                ko[n_out + idx_out] = k.y;
                ko[2 * n_out + idx_out] = k.z;
                if (valid_out_refr)
                    valid_out_refr[((MODE == PRT_MODE_PATH) ? off_out : 0) + idx_out] = valid ? 1 : 0;
            }
            x = xh;
            off_in += n_in;
            off_out += n_out;
            a = a_out;
        }
    }
}

// ---------------------------------------------------------------------------
// per-surface kernels (the plugin-granular API, and the march through
// anisotropic systems).  x is read modulo n_src so that the two children of a
// split ray share their parent's hit point without a copy.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PRT_BLOCK) void k_propagate(
    const prt_surface_t *__restrict__ sf, int64_t N, int64_t n_src,
    const double *__restrict__ x_in, const double *__restrict__ k_in,
    const double *__restrict__ dir_in, const double *__restrict__ e_re,
    const double *__restrict__ e_im, int32_t e_mode, const uint8_t *__restrict__ valid_in,
    double *__restrict__ xh_out, uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const int64_t j = (n_src == N) ? i : (i % n_src);
    const vec3 x = v3(x_in[j], x_in[n_src + j], x_in[2 * n_src + j]);
    vec3 d;
    if (dir_in) {
        d = v3(dir_in[i], dir_in[N + i], dir_in[2 * N + i]);
    } else {
        const vec3 kk = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
        vec3 k[2] = {kk, kk};
        vec3 dd[2];
        first_direction<false>(e_mode, e_re, e_im, N, i, false, k, dd);
        d = dd[0];
    }
    bool valid = valid_in ? (valid_in[i] != 0) : true;
    vec3 xh, p, g;
    double g2;
    propagate_step(sf, x, d, 1.0, xh, p, g, g2, valid);
    xh_out[i] = xh.x;
    xh_out[N + i] = xh.y;
    xh_out[2 * N + i] = xh.z;
    valid_out[i] = valid ? 1 : 0;
}

__global__ __launch_bounds__(PRT_BLOCK) void k_interact_iso(
    const prt_surface_t *__restrict__ sf, int64_t N, const double *__restrict__ xh_in,
    const double *__restrict__ k_in, const uint8_t *__restrict__ valid_in,
    double *__restrict__ k_out, double *__restrict__ dir_out, uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xh = v3(xh_in[i], xh_in[N + i], xh_in[2 * N + i]);
    vec3 k = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    bool valid = valid_in ? (valid_in[i] != 0) : true;
    const vec3 p = to_shape_frame(sf, xh);
    interact_isotropic(sf, normal_in_material_frame(sf, p), k, valid);
    k_out[i] = k.x;
    k_out[N + i] = k.y;
    k_out[2 * N + i] = k.z;
    if (dir_out) {
        const vec3 d = normalized(k);
        dir_out[i] = d.x;
        dir_out[N + i] = d.y;
        dir_out[2 * N + i] = d.z;
    }
    if (valid_out) valid_out[i] = valid ? 1 : 0;
}

__global__ __launch_bounds__(PRT_BLOCK) void k_interact_aniso(
    const prt_surface_t *__restrict__ sf, int64_t N, const double *__restrict__ xh_in,
    const double *__restrict__ k_in, const uint8_t *__restrict__ alive_in,
    double *__restrict__ k_out, double *__restrict__ dir_out, double *__restrict__ e_re_out,
    double *__restrict__ e_im_out, uint8_t *__restrict__ valid_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 xh = v3(xh_in[i], xh_in[N + i], xh_in[2 * N + i]);
    const vec3 k = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    const vec3 p = to_shape_frame(sf, xh);
    // The reference's anisotropic refract does no validity filtering: every ray that is
    // still IN the bundle gets two children in a fresh all-valid bundle (ray.py:68).
    // In the dense representation "in the bundle" = alive_in (the mask the previous
    // compaction used); rays compacted away earlier must stay dead.
    const uint8_t alive = alive_in ? alive_in[i] : (uint8_t)1;
    aniso_solution sol[2];
    interact_anisotropic(sf, p, k, sol);
    const int64_t M = 2 * N;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t o = i + b * N;  // np.hstack((sol2, sol3)), material_anisotropic.py:89
        k_out[o] = sol[b].k.x;
        k_out[M + o] = sol[b].k.y;
        k_out[2 * M + o] = sol[b].k.z;
        dir_out[o] = sol[b].d.x;
        dir_out[M + o] = sol[b].d.y;
        dir_out[2 * M + o] = sol[b].d.z;
        if (e_re_out) {
            e_re_out[o] = sol[b].er.x;
            e_re_out[M + o] = sol[b].er.y;
            e_re_out[2 * M + o] = sol[b].er.z;
        }
        if (e_im_out) {
            e_im_out[o] = sol[b].ei.x;
            e_im_out[M + o] = sol[b].ei.y;
            e_im_out[2 * M + o] = sol[b].ei.z;
        }
        if (valid_out) valid_out[o] = alive;
    }
}

__global__ __launch_bounds__(PRT_BLOCK) void k_shape_eval(const prt_surface_t *__restrict__ sf,
                                                          int64_t N, const double *__restrict__ x,
                                                          const double *__restrict__ y,
                                                          double *__restrict__ sag,
                                                          double *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const double xx = x[i], yy = y[i];
    if (sag) sag[i] = shape_sag(sf, xx, yy);
    if (grad) {
        vec3 g;
        if (sf->shape_type == PRT_SHAPE_CONIC) {
            // Conic.getGrad as the reference evaluates it (surface_shape.py:229-235)
            const double z = conic_sag(sf->curv, sf->cc, xx * xx + yy * yy);
            g = v3(-sf->curv * xx, -sf->curv * yy, 1.0 - sf->curv * z * (1.0 + sf->cc));
        } else {
            g = shape_grad(sf, xx, yy);
        }
        grad[i] = g.x;
        grad[N + i] = g.y;
        grad[2 * N + i] = g.z;
    }
}

// A unit E field perpendicular to k for bundles leaving an isotropic interface.  The
// reference takes the singular vector of the smallest singular value of
// -k^2 I + k k^T + n^2 I (material_isotropic.py:72-128), which is an ARBITRARY unit
// vector of the 2-d null space {E : E.k = 0}; this picks E = unit(k x a), a = the
// coordinate axis least aligned with k.  Not on the x/k parity contract.
__global__ __launch_bounds__(PRT_BLOCK) void k_efield_perp(int64_t N, const double *__restrict__ k_in,
                                                           double *__restrict__ e_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 k = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    const double ax = fabs(k.x), ay = fabs(k.y), az = fabs(k.z);
    const vec3 a = (ay <= ax && ay <= az) ? v3(0, 1, 0) : ((ax <= az) ? v3(1, 0, 0) : v3(0, 0, 1));
    const vec3 e = normalized(cross(k, a));
    e_out[i] = e.x;
    e_out[N + i] = e.y;
    e_out[2 * N + i] = e.z;
}

// RayBundle.returnKtoD (ray.py:136-152) for one stored point: unit Poynting direction from
// (k, E); e_mode as in first_direction (0: k/|k|, 1: E = ey, 2: E given).
__global__ __launch_bounds__(PRT_BLOCK) void k_poynting_dir(int64_t N, const double *__restrict__ k_in,
                                                            const double *__restrict__ e_re,
                                                            const double *__restrict__ e_im,
                                                            int32_t e_mode,
                                                            double *__restrict__ d_out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    const vec3 kk = v3(k_in[i], k_in[N + i], k_in[2 * N + i]);
    vec3 k[2] = {kk, kk};
    vec3 dd[2];
    first_direction<false>(e_mode, e_re, e_im, N, i, false, k, dd);
    d_out[i] = dd[0].x;
    d_out[N + i] = dd[0].y;
    d_out[2 * N + i] = dd[0].z;
}

// RayBundleAnalysis.get_arc_length / get_phase_difference (analysis/ray_analysis.py:136-163):
// per ray, sum over consecutive stored points p of |x_{p+1} - x_p|  (mode 0) or of
// x_{p+1}.k_{p+1} - x_p.k_p (mode 1).  xs / ks: device tables of P pointers to tight (3,N) arrays.
__global__ __launch_bounds__(PRT_BLOCK) void k_path_sums(int32_t P, int64_t N,
                                                         const double *const *__restrict__ xs,
                                                         const double *const *__restrict__ ks,
                                                         int32_t mode, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (i >= N) return;
    double acc = 0.0;
    vec3 xa = v3(xs[0][i], xs[0][N + i], xs[0][2 * N + i]);
    double pa = 0.0;
    if (mode == 1) pa = dot(xa, v3(ks[0][i], ks[0][N + i], ks[0][2 * N + i]));
    for (int p = 1; p < P; ++p) {
        const vec3 xb = v3(xs[p][i], xs[p][N + i], xs[p][2 * N + i]);
        if (mode == 0) {
            const vec3 dlt = v3(xb.x - xa.x, xb.y - xa.y, xb.z - xa.z);
            acc += sqrt(dot(dlt, dlt));
        } else {
            const double pb = dot(xb, v3(ks[p][i], ks[p][N + i], ks[p][2 * N + i]));
            acc += pb - pa;
            pa = pb;
        }
        xa = xb;
    }
    out[i] = acc;
}

// ---------------------------------------------------------------------------
// bundle moments: count, sum (x - ref), sum (x - ref)^2 per component over the rays whose
// mask byte is non-zero (all rays if mask == NULL).  Two deterministic stages (fixed
// summation order, no atomics): grid-stride partials per block, then one block adds the
// partials.  Feeds RayBundleAnalysis.get_centroid_position / get_rms_spot_size
// (analysis/ray_analysis.py:44-86) and turns the multi-GPU image-plane exchange into a
// 7-double all-reduce.
// ---------------------------------------------------------------------------
#define MOM_VALUES 7
__global__ __launch_bounds__(PRT_BLOCK) void k_moments_partial(int64_t N, int64_t pitch,
                                                               const double *__restrict__ x,
                                                               const uint8_t *__restrict__ mask,
                                                               int32_t mode, double rx, double ry,
                                                               double rz,
                                                               const double *__restrict__ ref_dev,
                                                               int32_t ref_kind,
                                                               double *__restrict__ partials) {
    // ref_kind 1: ref_dev holds the reference point (3 doubles); 2: ref_dev holds a moments
    // vector {count, sum x, ...} (e.g. all-reduced over the ranks) -> reference = centroid
    if (ref_kind == 1) {
        rx = ref_dev[0];
        ry = ref_dev[1];
        rz = ref_dev[2];
    } else if (ref_kind == 2) {
        const double inv = 1.0 / (ref_dev[0] + 1e-17);  // numerical_tolerance, ray_analysis.py:55
        rx = ref_dev[1] * inv;
        ry = ref_dev[2] * inv;
        rz = ref_dev[3] * inv;
    }
    double acc[MOM_VALUES] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * PRT_BLOCK) {
        if (mask && !mask[i]) continue;
        double dx = x[i], dy = x[pitch + i], dz = x[2 * pitch + i];
        if (mode == 0) {  // points relative to ref
            dx -= rx;
            dy -= ry;
            dz -= rz;
        } else {  // unit directions (ray.py:136-152 for E perpendicular to k), optionally x ref
            const vec3 u = normalized(v3(dx, dy, dz));
            dx = u.x;
            dy = u.y;
            dz = u.z;
            if (mode == 2) {
                const vec3 c = cross(u, v3(rx, ry, rz));
                dx = c.x;
                dy = c.y;
                dz = c.z;
            }
        }
        acc[0] += 1.0;
        acc[1] += dx;
        acc[2] += dy;
        acc[3] += dz;
        acc[4] += dx * dx;
        acc[5] += dy * dy;
        acc[6] += dz * dz;
    }
    __shared__ double sh[PRT_BLOCK / 64][MOM_VALUES];
#pragma unroll
    for (int q = 0; q < MOM_VALUES; ++q) {
        double v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < MOM_VALUES) {
        double v = 0.0;
        for (int w = 0; w < PRT_BLOCK / 64; ++w) v += sh[w][threadIdx.x];
        partials[(int64_t)blockIdx.x * MOM_VALUES + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(PRT_BLOCK) void k_moments_final(int nblocks_,
                                                             const double *__restrict__ partials,
                                                             double *__restrict__ out) {
    // fixed summation order: thread t adds blocks t, t+256, ...; then a fixed-shape tree
    __shared__ double sh[PRT_BLOCK][MOM_VALUES];
    double acc[MOM_VALUES] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks_; b += PRT_BLOCK)
#pragma unroll
        for (int q = 0; q < MOM_VALUES; ++q) acc[q] += partials[(int64_t)b * MOM_VALUES + q];
#pragma unroll
    for (int q = 0; q < MOM_VALUES; ++q) sh[threadIdx.x][q] = acc[q];
    __syncthreads();
    for (int stride = PRT_BLOCK / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int q = 0; q < MOM_VALUES; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + stride][q];
        __syncthreads();
    }
    if (threadIdx.x < MOM_VALUES) out[threadIdx.x] = sh[0][threadIdx.x];
}

// Order-preserving slot assignment inside one CMP_TILE (= 4 sub-tiles of PRT_BLOCK consecutive
// elements; thread t owns elements q*PRT_BLOCK + t, so loads and stores are lane-consecutive):
// ballot + popcount per wave, then a prefix over the 4 x 4 (sub-tile, wave) counts.
// keep[q] in -> pos[q] out (offset of the element among the tile's survivors).
PRT_DEV void tile_slots(const bool keep[CMP_ITEMS], int pos[CMP_ITEMS]) {
    __shared__ int cnt[CMP_ITEMS][PRT_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const unsigned long long b = __ballot(keep[q]);
        rank[q] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) cnt[q][wave] = __popcll(b);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
#pragma unroll
        for (int w = 0; w < PRT_BLOCK / 64; ++w) {
            if (w == wave) pos[q] = run + rank[q];
            run += cnt[q][w];
        }
    }
}

// ---------------------------------------------------------------------------
// compaction: per-block popcount -> single-block scan of block totals -> scatter
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(PRT_BLOCK) void k_compact_count(const uint8_t *__restrict__ mask,
                                                             int64_t N,
                                                             int64_t *__restrict__ block_sums) {
    __shared__ int wsum[PRT_BLOCK / 64];
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    int c = 0;
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        if (idx < N && mask[idx]) ++c;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < PRT_BLOCK / 64; ++w) t += wsum[w];
        block_sums[blockIdx.x] = t;
    }
}

// exclusive scan of block_sums in place; total -> block_sums[nb]
__global__ __launch_bounds__(PRT_BLOCK) void k_compact_scan(int64_t *__restrict__ block_sums,
                                                            int64_t nb) {
    __shared__ int64_t part[PRT_BLOCK];
    const int64_t chunk = (nb + PRT_BLOCK - 1) / PRT_BLOCK;
    const int64_t lo = (int64_t)threadIdx.x * chunk;
    const int64_t hi = (lo + chunk < nb) ? lo + chunk : nb;
    int64_t s = 0;
    for (int64_t q = lo; q < hi; ++q) s += block_sums[q];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int q = 0; q < PRT_BLOCK; ++q) {
            const int64_t v = part[q];
            part[q] = run;
            run += v;
        }
        block_sums[nb] = run;
    }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int64_t q = lo; q < hi; ++q) {
        const int64_t v = block_sums[q];
        block_sums[q] = run;
        run += v;
    }
}

__global__ __launch_bounds__(PRT_BLOCK) void k_compact_scatter(
    const uint8_t *__restrict__ mask, int64_t N, const int64_t *__restrict__ block_offs,
    int32_t n_arrays, const double *const *__restrict__ src, double *const *__restrict__ dst,
    const int64_t *__restrict__ id_src, int64_t *__restrict__ id_dst,
    const uint8_t *__restrict__ u8_src, uint8_t *__restrict__ u8_dst) {
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    bool keep[CMP_ITEMS];
    int pos[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        keep[q] = (idx < N) && mask[idx];
    }
    tile_slots(keep, pos);
    const int64_t off = block_offs[blockIdx.x];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        if (!keep[q]) continue;
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        const int64_t o = off + pos[q];
        for (int a = 0; a < n_arrays; ++a) dst[a][o] = src[a][idx];
        if (id_src) id_dst[o] = id_src[idx];
        if (u8_src) u8_dst[o] = u8_src[idx];
    }
}

// ---------------------------------------------------------------------------
// device-side bundle generation: RectGrid.getGrid (sampling2d/raster.py:40-60) +
// OpticalSystemAnalysis.collimated_bundle (analysis/optical_system_analysis.py:83-122).
// The raster is reproduced BIT-EXACTLY: numpy.linspace is i*step + start with two
// roundings and the last sample forced to `stop`; the disk test is x*x + y*y <= 1 with
// separate roundings -- hence the explicit _rn intrinsics (no FMA contraction).
// ---------------------------------------------------------------------------
// a*b and a+b rounded separately: HIP's __dmul_rn/__dadd_rn are plain operators that hipcc
// contracts into FMAs, so contraction is switched off per statement instead
PRT_DEV double mul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
PRT_DEV double add_rn(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
PRT_DEV double lin_sample(int64_t i, int64_t n, double start, double step, double stop) {
    return (i == n - 1) ? stop : add_rn(mul_rn((double)i, step), start);
}

__global__ __launch_bounds__(PRT_BLOCK) void k_rectgrid_mask(int64_t n, double start, double step,
                                                             double stop,
                                                             uint8_t *__restrict__ mask) {
    const int64_t idx = (int64_t)blockIdx.x * PRT_BLOCK + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t iy = idx / n, ix = idx - iy * n;  // np.meshgrid(x1d, x1d): x varies fastest
    const double x = lin_sample(ix, n, start, step, stop), y = lin_sample(iy, n, start, step, stop);
    mask[idx] = (add_rn(mul_rn(x, x), mul_rn(y, y)) <= 1.0) ? 1 : 0;
}

struct collimated_params {
    double radius, startx, starty, startz;
    double k[3], e[3];
};

__global__ __launch_bounds__(PRT_BLOCK) void k_rectgrid_scatter(
    const uint8_t *__restrict__ mask, int64_t n, double start, double step, double stop,
    const int64_t *__restrict__ block_offs, int64_t lo, int64_t hi, collimated_params prm,
    int64_t pitch, double *__restrict__ x_out, double *__restrict__ k_out,
    double *__restrict__ e_out) {
    const int64_t total = n * n;
    const int64_t tile = (int64_t)blockIdx.x * CMP_TILE;
    bool keep[CMP_ITEMS];
    int pos[CMP_ITEMS];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        keep[q] = (idx < total) && mask[idx];
    }
    tile_slots(keep, pos);
    const int64_t off = block_offs[blockIdx.x];
#pragma unroll
    for (int q = 0; q < CMP_ITEMS; ++q) {
        if (!keep[q]) continue;
        const int64_t p = off + pos[q];
        if (p < lo || p >= hi) continue;
        const int64_t idx = tile + q * PRT_BLOCK + threadIdx.x;
        const int64_t iy = idx / n, ix = idx - iy * n;
        const double px = lin_sample(ix, n, start, step, stop);
        const double py = lin_sample(iy, n, start, step, stop);
        const int64_t o = p - lo;
        // origin = radius * p + start (optical_system_analysis.py:106-108), two roundings
        x_out[o] = add_rn(mul_rn(prm.radius, px), prm.startx);
        x_out[pitch + o] = add_rn(mul_rn(prm.radius, py), prm.starty);
        x_out[2 * pitch + o] = prm.startz;
        k_out[o] = prm.k[0];
        k_out[pitch + o] = prm.k[1];
        k_out[2 * pitch + o] = prm.k[2];
        if (e_out) {
            e_out[o] = prm.e[0];
            e_out[pitch + o] = prm.e[1];
            e_out[2 * pitch + o] = prm.e[2];
        }
    }
}

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

template <int MODE>
static void launch_trace_iso(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                             const double *k0, const double *e_re, const double *e_im,
                             int32_t e_mode, int64_t out_pitch, double *x_hit, double *k_out,
                             uint8_t *valid, uint8_t *valid_out, bool vec_in, bool vec_out,
                             hipStream_t st) {
    const dim3 grid(nblocks(n0, PRT_BLOCK * 2)), block(PRT_BLOCK);
#define PRT_LAUNCH_E(VI, VO, EX)                                                                 \
    hipLaunchKernelGGL((k_trace_iso<MODE, VI, VO, EX>), grid, block, 0, st, sys->d_table,        \
                       sys->n_surfaces, n0, in_pitch, x0, k0, e_re, e_im, e_mode, out_pitch,     \
                       x_hit, k_out, valid, valid_out)
#define PRT_LAUNCH(VI, VO)                 \
    do {                                   \
        if (sys->all_conic)                \
            PRT_LAUNCH_E(VI, VO, false);   \
        else                               \
            PRT_LAUNCH_E(VI, VO, true);    \
    } while (0)
    static const bool lds_table = getenv("PRT_LDS_TABLE") != nullptr;
    if (vec_in && vec_out && lds_table && sys->all_conic && sys->n_surfaces <= PRT_LDS_TAB_MAX)
        hipLaunchKernelGGL((k_trace_iso<MODE, true, true, false, true>), grid, block, 0, st, sys->d_table,
                           sys->n_surfaces, n0, in_pitch, x0, k0, e_re, e_im, e_mode, out_pitch, x_hit,
                           k_out, valid, valid_out);
    else if (vec_in && vec_out)
        PRT_LAUNCH(true, true);
    else if (vec_in)
        PRT_LAUNCH(true, false);
    else if (vec_out)
        PRT_LAUNCH(false, true);
    else
        PRT_LAUNCH(false, false);
#undef PRT_LAUNCH
#undef PRT_LAUNCH_E
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

static int32_t check_record(const prt_surface_t *r, int idx) {
    char msg[128];
    if (r->shape_type < PRT_SHAPE_CONIC || r->shape_type > PRT_SHAPE_BICONIC) {
        snprintf(msg, sizeof msg, "surface %d: unknown shape_type %d", idx, r->shape_type);
        return fail(PRT_ERR_UNSUPPORTED, msg);
    }
    if (r->n_coeffs < 0 || r->n_coeffs > PRT_MAX_COEFFS ||
        (r->shape_type == PRT_SHAPE_BICONIC && 2 * r->n_coeffs > PRT_MAX_COEFFS)) {
        snprintf(msg, sizeof msg, "surface %d: n_coeffs %d out of range", idx, r->n_coeffs);
        return fail(PRT_ERR_INVALID_ARG, msg);
    }
    if (r->ap_type < PRT_AP_NONE || r->ap_type > PRT_AP_RECTANGULAR ||
        r->interaction < PRT_REFRACT || r->interaction > PRT_MIRROR ||
        r->mat_type < PRT_MAT_ISOTROPIC || r->mat_type > PRT_MAT_ANISOTROPIC) {
        snprintf(msg, sizeof msg, "surface %d: bad aperture/interaction/material enum", idx);
        return fail(PRT_ERR_INVALID_ARG, msg);
    }
    if (r->mat_type == PRT_MAT_ANISOTROPIC) {
        for (int q = 0; q < 9; ++q)
            if (r->eps_im[q] != 0.0) {
                snprintf(msg, sizeof msg, "surface %d: complex epsilon tensor not supported", idx);
                return fail(PRT_ERR_UNSUPPORTED, msg);
            }
    }
    return PRT_OK;
}

int32_t prt_system_create(const prt_surface_t *table, int32_t n_surfaces, int32_t device,
                          prt_system_t **out) {
    if (!table || !out || n_surfaces <= 0) return fail(PRT_ERR_INVALID_ARG, "prt_system_create: null/empty");
    *out = nullptr;
    for (int s = 0; s < n_surfaces; ++s) {
        int32_t rc = check_record(table + s, s);
        if (rc != PRT_OK) return rc;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(PRT_ERR_NO_DEVICE, "no HIP device visible", e);
    if (device < 0 || device >= ndev) return fail(PRT_ERR_INVALID_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    prt_system *sys = new (std::nothrow) prt_system();
    if (!sys) return fail(PRT_ERR_NOMEM, "host alloc");
    sys->device = device;
    sys->n_surfaces = n_surfaces;
    sys->h_table = new (std::nothrow) prt_surface_t[n_surfaces];
    if (!sys->h_table) {
        delete sys;
        return fail(PRT_ERR_NOMEM, "host alloc");
    }
    memcpy(sys->h_table, table, sizeof(prt_surface_t) * n_surfaces);
    sys->all_isotropic = 1;
    sys->all_conic = 1;
    for (int s = 0; s < n_surfaces; ++s) {
        if (table[s].mat_type != PRT_MAT_ISOTROPIC) sys->all_isotropic = 0;
        if (table[s].shape_type != PRT_SHAPE_CONIC) sys->all_conic = 0;
    }
    e = hipMalloc((void **)&sys->d_table, sizeof(prt_surface_t) * n_surfaces);
    if (e != hipSuccess) {
        delete[] sys->h_table;
        delete sys;
        return fail(PRT_ERR_NOMEM, "hipMalloc(table)", e);
    }
    e = hipMemcpy(sys->d_table, table, sizeof(prt_surface_t) * n_surfaces, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(sys->d_table);
        delete[] sys->h_table;
        delete sys;
        return fail(PRT_ERR_DEVICE, "hipMemcpy(table)", e);
    }
    *out = sys;
    return PRT_OK;
}

int32_t prt_system_destroy(prt_system_t *sys) {
    if (!sys) return PRT_OK;
    (void)hipSetDevice(sys->device);
    (void)hipFree(sys->d_table);
    delete[] sys->h_table;
    delete sys;
    return PRT_OK;
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
                             int32_t mode, double *x_hit, double *k_out, uint8_t *valid,
                             uint8_t *valid_out, hipStream_t st) {
    const int S = sys->n_surfaces;
    // scratch: directions after anisotropic interfaces, plus ping-pong state in IMAGE mode
    int64_t n_final = n0;
    for (int s = 0; s < S; ++s)
        if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) n_final *= 2;
    double *dirbuf[2] = {nullptr, nullptr};
    double *xbuf[2] = {nullptr, nullptr}, *kbuf[2] = {nullptr, nullptr};
    uint8_t *vbuf[2] = {nullptr, nullptr}, *wbuf[2] = {nullptr, nullptr};
    HIP_TRY(hipMallocAsync((void **)&dirbuf[0], sizeof(double) * 3 * n_final, st));
    HIP_TRY(hipMallocAsync((void **)&dirbuf[1], sizeof(double) * 3 * n_final, st));
    if (mode == PRT_MODE_IMAGE) {
        for (int b = 0; b < 2; ++b) {
            HIP_TRY(hipMallocAsync((void **)&xbuf[b], sizeof(double) * 3 * n_final, st));
            HIP_TRY(hipMallocAsync((void **)&kbuf[b], sizeof(double) * 3 * n_final, st));
            HIP_TRY(hipMallocAsync((void **)&vbuf[b], n_final, st));
            HIP_TRY(hipMallocAsync((void **)&wbuf[b], n_final, st));
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
        HIP_TRY(hipMallocAsync((void **)&vo_scratch, tot_out, st));
        valid_out = vo_scratch;
    }

    const double *cur_x = x0, *cur_k = k0, *cur_dir = nullptr;
    const uint8_t *cur_valid = nullptr;
    int64_t n_src = n0;  // number of distinct points in cur_x
    int64_t n = n0;
    int64_t off_in = 0, off_out = 0;  // element offsets into the concatenated outputs
    int e_mode = e_mode_of(e_re, 1);
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
                           (s == 0) ? e_mode : 0, cur_valid, xh_dst, v_dst);
        double *dir_dst = dirbuf[s & 1];
        if (aniso) {
            hipLaunchKernelGGL(k_interact_aniso, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                               st, sys->d_table + s, n, xh_dst, cur_k, cur_valid, k_dst, dir_dst,
                               (double *)nullptr, (double *)nullptr, vo_dst);
            cur_dir = dir_dst;
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
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipFreeAsync(dirbuf[b], st));
        if (xbuf[b]) HIP_TRY(hipFreeAsync(xbuf[b], st));
        if (kbuf[b]) HIP_TRY(hipFreeAsync(kbuf[b], st));
        if (vbuf[b]) HIP_TRY(hipFreeAsync(vbuf[b], st));
        if (wbuf[b]) HIP_TRY(hipFreeAsync(wbuf[b], st));
    }
    if (vo_scratch) HIP_TRY(hipFreeAsync(vo_scratch, st));
    return PRT_OK;
}

int64_t prt_recommended_pitch(int64_t n) {
    if (n <= 0) return 0;
    return (n + 511) / 512 * 512;  // 4 KiB of doubles: every row starts on a 128-B line
}

int32_t prt_trace(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                  const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                  int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                  uint8_t *valid_out, void *stream) {
    if (!sys || n0 < 0) return fail(PRT_ERR_INVALID_ARG, "prt_trace: null system / negative count");
    if (mode != PRT_MODE_PATH && mode != PRT_MODE_IMAGE)
        return fail(PRT_ERR_INVALID_ARG, "prt_trace: bad mode");
    if (n0 == 0) return PRT_OK;  // empty bundle: nothing to do (buffers may be NULL)
    if (!x0 || !k0 || !x_hit || !k_out || !valid)
        return fail(PRT_ERR_INVALID_ARG, "prt_trace: null pointer");
    if (in_pitch == 0) in_pitch = n0;
    if (in_pitch < n0 || (out_pitch != 0 && out_pitch < n0))
        return fail(PRT_ERR_INVALID_ARG, "prt_trace: pitch smaller than the ray count");
    HIP_TRY(hipSetDevice(sys->device));
    hipStream_t st = (hipStream_t)stream;
    if (!sys->all_isotropic) {
        if (out_pitch != 0 || in_pitch != n0)
            return fail(PRT_ERR_INVALID_ARG,
                        "prt_trace: tables with anisotropic media use the concatenated layout (pitch 0)");
        static const bool per_surface = getenv("PRT_GENERAL_PER_SURFACE") != nullptr;
        int n_aniso = 0;
        bool general_eps = false;
        for (int s = 0; s < sys->n_surfaces; ++s)
            if (sys->h_table[s].mat_type == PRT_MAT_ANISOTROPIC) {
                ++n_aniso;
                if (sys->h_table[s].aniso_class == PRT_ANISO_GENERAL) general_eps = true;
            }
        // The fused march re-traces shared prefixes (2^A leaves per thread).  That pays for the
        // closed-form crystal classes (measured, 1e6 rays through the doublet of config 4:
        // 0.44 vs 0.54 ms) but not when every interface runs the iterative quartic solver
        // (biaxial: 2.1 vs 1.2 ms), nor for many interfaces.
        if (per_surface || general_eps || n_aniso > 4)
            return trace_general(sys, n0, x0, k0, e0_re, e0_im, mode, x_hit, k_out, valid, valid_out, st);
        const dim3 grid(nblocks(n0, PRT_BLOCK)), block(PRT_BLOCK);
        const int32_t e_mode_g = e_mode_of(e0_re, 1);
        if (mode == PRT_MODE_PATH)
            hipLaunchKernelGGL((k_trace_general<PRT_MODE_PATH>), grid, block, 0, st, sys->d_table,
                               sys->n_surfaces, n_aniso, n0, x0, k0, e0_re, e0_im, e_mode_g, x_hit, k_out,
                               valid, valid_out);
        else
            hipLaunchKernelGGL((k_trace_general<PRT_MODE_IMAGE>), grid, block, 0, st, sys->d_table,
                               sys->n_surfaces, n_aniso, n0, x0, k0, e0_re, e0_im, e_mode_g, x_hit, k_out,
                               valid, valid_out);
        HIP_TRY(hipGetLastError());
        return PRT_OK;
    }
    if (out_pitch == 0) out_pitch = n0;
    const int32_t e_mode = e_mode_of(e0_re, 1);
    const bool vec_in = (in_pitch % 2 == 0) && aligned16(x0) && aligned16(k0) &&
                        (!e0_re || aligned16(e0_re)) && (!e0_im || aligned16(e0_im));
    const bool vec_out = (out_pitch % 2 == 0) && aligned16(x_hit) && aligned16(k_out) &&
                         ((((uintptr_t)valid) & 1u) == 0) &&
                         (!valid_out || (((uintptr_t)valid_out) & 1u) == 0) &&
                         (n0 % 2 == 0 || out_pitch > n0);  // odd N: the tail lane's 2nd ray lands in the padding
    if (mode == PRT_MODE_PATH)
        launch_trace_iso<PRT_MODE_PATH>(sys, n0, in_pitch, x0, k0, e0_re, e0_im, e_mode, out_pitch,
                                        x_hit, k_out, valid, valid_out, vec_in, vec_out, st);
    else
        launch_trace_iso<PRT_MODE_IMAGE>(sys, n0, in_pitch, x0, k0, e0_re, e0_im, e_mode, out_pitch,
                                         x_hit, k_out, valid, valid_out, vec_in, vec_out, st);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_trace_timed(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                        const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                        int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                        uint8_t *valid_out, void *stream, int32_t iters, double *ms_avg) {
    if (!ms_avg || iters <= 0) return fail(PRT_ERR_INVALID_ARG, "prt_trace_timed");
    if (!sys) return fail(PRT_ERR_INVALID_ARG, "null system");
    HIP_TRY(hipSetDevice(sys->device));
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, st));
    for (int it = 0; it < iters; ++it) {
        int32_t rc = prt_trace(sys, n0, in_pitch, x0, k0, e0_re, e0_im, mode, out_pitch, x_hit, k_out,
                               valid, valid_out, stream);
        if (rc != PRT_OK) {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
            return rc;
        }
    }
    HIP_TRY(hipEventRecord(b, st));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_avg = (double)ms / iters;
    return PRT_OK;
}

int32_t prt_propagate(const prt_system_t *sys, int32_t surface, int64_t n, const double *x,
                      const double *k, const double *dir, const double *e_re, const double *e_im,
                      int32_t use_default_e, const uint8_t *valid_in, double *x_hit,
                      uint8_t *valid, void *stream) {
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_propagate: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x || (!k && !dir) || !x_hit || !valid)
        return fail(PRT_ERR_INVALID_ARG, "prt_propagate: null pointer");
    HIP_TRY(hipSetDevice(sys->device));
    hipLaunchKernelGGL(k_propagate, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                       (hipStream_t)stream, sys->d_table + surface, n, n, x, k, dir, e_re, e_im,
                       e_mode_of(e_re, use_default_e), valid_in, x_hit, valid);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_interact(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit,
                     const double *k, const uint8_t *valid_in, double *k_out, double *dir_out,
                     double *e_out_re, double *e_out_im, uint8_t *valid_out, void *stream) {
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_interact: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x_hit || !k || !k_out) return fail(PRT_ERR_INVALID_ARG, "prt_interact: null pointer");
    HIP_TRY(hipSetDevice(sys->device));
    const prt_surface_t *rec = sys->h_table + surface;
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

int32_t prt_shape_eval(const prt_system_t *sys, int32_t surface, int64_t n, const double *x,
                       const double *y, double *sag, double *grad, void *stream) {
    if (!sys || surface < 0 || surface >= sys->n_surfaces || n < 0)
        return fail(PRT_ERR_INVALID_ARG, "prt_shape_eval: bad system / surface / count");
    if (n == 0) return PRT_OK;
    if (!x || !y) return fail(PRT_ERR_INVALID_ARG, "prt_shape_eval: null pointer");
    HIP_TRY(hipSetDevice(sys->device));
    hipLaunchKernelGGL(k_shape_eval, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0,
                       (hipStream_t)stream, sys->d_table + surface, n, x, y, sag, grad);
    HIP_TRY(hipGetLastError());
    return PRT_OK;
}

int32_t prt_efield_perp(int32_t device, int64_t n, const double *k, double *e_out, void *stream) {
    if (n < 0) return fail(PRT_ERR_INVALID_ARG, "prt_efield_perp: negative count");
    if (n == 0) return PRT_OK;
    if (!k || !e_out) return fail(PRT_ERR_INVALID_ARG, "prt_efield_perp: null pointer");
    HIP_TRY(hipSetDevice(device));
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
    HIP_TRY(hipSetDevice(device));
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
    if (!x_out || !k_out) return fail(PRT_ERR_INVALID_ARG, "prt_collimated_bundle: null pointer");
    if (pitch == 0) pitch = hi - lo;
    if (pitch < hi - lo) return fail(PRT_ERR_INVALID_ARG, "prt_collimated_bundle: pitch < hi - lo");
    HIP_TRY(hipSetDevice(device));
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

int32_t prt_poynting_dir(int32_t device, int64_t n, const double *k, const double *e_re,
                         const double *e_im, int32_t use_default_e, double *d_out, void *stream) {
    if (n < 0) return fail(PRT_ERR_INVALID_ARG, "prt_poynting_dir: negative count");
    if (n == 0) return PRT_OK;
    if (!k || !d_out) return fail(PRT_ERR_INVALID_ARG, "prt_poynting_dir: null pointer");
    HIP_TRY(hipSetDevice(device));
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
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const double **d_tab = nullptr;
    HIP_TRY(hipMallocAsync((void **)&d_tab, sizeof(void *) * 2 * n_points, st));
    HIP_TRY(hipMemcpyAsync((void *)d_tab, xs, sizeof(void *) * n_points, hipMemcpyHostToDevice, st));
    if (ks)
        HIP_TRY(hipMemcpyAsync((void *)(d_tab + n_points), ks, sizeof(void *) * n_points,
                               hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_path_sums, dim3(nblocks(n, PRT_BLOCK)), dim3(PRT_BLOCK), 0, st, n_points, n,
                       (const double *const *)d_tab, (const double *const *)(d_tab + n_points), mode, out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));   // the host pointer tables must outlive the copies
    HIP_TRY(hipFreeAsync((void *)d_tab, st));
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
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)((n + PRT_BLOCK * 8 - 1) / (PRT_BLOCK * 8));
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    double *scratch = nullptr;
    HIP_TRY(hipMallocAsync((void **)&scratch, sizeof(double) * MOM_VALUES * (nb + 1), st));
    const double rx = ref ? ref[0] : 0.0, ry = ref ? ref[1] : 0.0, rz = ref ? ref[2] : 0.0;
    hipLaunchKernelGGL(k_moments_partial, dim3(nb), dim3(PRT_BLOCK), 0, st, n, pitch, x, mask, mode, rx,
                       ry, rz, (const double *)nullptr, 0, scratch);
    hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(PRT_BLOCK), 0, st, nb, scratch,
                       scratch + (int64_t)nb * MOM_VALUES);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out7, scratch + (int64_t)nb * MOM_VALUES, sizeof(double) * MOM_VALUES,
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipFreeAsync(scratch, st));
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
    HIP_TRY(hipSetDevice(device));
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
    // block sums (+1 total) and the two pointer tables (up to 16 arrays each)
    return (nb + 1) * (int64_t)sizeof(int64_t) + 2 * 16 * (int64_t)sizeof(void *) + 64;
}

int32_t prt_compact(int64_t n, const uint8_t *mask, int32_t n_arrays, const double *const *src,
                    double *const *dst, const int64_t *id_src, int64_t *id_dst,
                    const uint8_t *u8_src, uint8_t *u8_dst, void *scratch, int64_t *n_kept,
                    void *stream) {
    if (n < 0 || (n > 0 && !mask) || n_arrays < 0 || n_arrays > 16 || (n_arrays && (!src || !dst)) ||
        (n > 0 && !scratch) || !n_kept || (id_src && !id_dst) || (u8_src && !u8_dst))
        return fail(PRT_ERR_INVALID_ARG, "prt_compact: bad argument");
    *n_kept = 0;
    if (n == 0) return PRT_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (n + CMP_TILE - 1) / CMP_TILE;
    int64_t *sums = (int64_t *)scratch;
    uintptr_t pbase = ((uintptr_t)(sums + nb + 1) + 15u) & ~(uintptr_t)15u;
    const double **d_src = (const double **)pbase;
    double **d_dst = (double **)(pbase + 16 * sizeof(void *));
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
