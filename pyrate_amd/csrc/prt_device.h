// prt_device.h -- per-ray device functions of the gfx950 raytrace engine.
//
// One thread owns one ray (or RPT adjacent rays): state x(3), k(3), valid lives in
// VGPRs; everything that is the same for all rays -- the surface record: frames,
// shape coefficients, aperture, index -- is read through wave-uniform addresses,
// which hipcc turns into s_load (scalar cache -> SGPRs).  FP64 VALU instructions
// take SGPR pairs as operands directly, so uniform coefficients cost no VGPRs, no
// LDS traffic and no ds_read issue slots; that is why the table is NOT staged
// through LDS (measured alternative: see DESIGN.md "surface table placement").
//
// No MFMA anywhere: the work is a per-ray chain of ~150 dependent FP64 ops with
// 3x3 mat-vecs whose matrix is wave-uniform -- there is no contraction dimension
// to feed a matrix core.
//
// Formulas follow the reference (cited per function); algebraically identical
// short-cuts are marked "== " with the identity used.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/prt.h"

#define PRT_DEV __device__ __forceinline__

// The record the kernels read: prt_surface_t (include/prt.h, what a caller fills in) repacked by
// prt_system_create -- same field names, but the coefficient arrays (3 x PRT_MAX_COEFFS entries, 2 KB)
// are replaced by pointers into one side array on the device that holds only the entries in use, and
// fields the kernels never look at (eps_im) are gone.  504 bytes per surface instead of 2.7 KB; a
// 12-surface table is 6 KB and stays in the scalar cache whatever else runs.
struct prt_dev_surface {
    int32_t shape_type, n_coeffs, ap_type, interaction, mat_type, frame_flags, newton_maxit, aniso_class;
    int32_t n_asphere, grid_nx, grid_ny;
    int32_t poly_dense;  // xypoly: 1 if the polynomial fits the dense triangle of total degree <= PRT_POLY_DENSE_DEG (dense_poly_eval)
    double curv, cc;
    const double *coeffs;  // n_coeffs doubles (biconic: 2 n_coeffs); GRIDSAG: knots tx, ty, coefficients
    const void *pows;      // xypoly / combo: the polynomial part as dense Horner rows (xypoly_eval)
    double B_shape[9], g_shape[3];
    double B_ap[9], g_ap[3];
    double ap_p0, ap_p1;
    double B_mat[9];
    double n_after;
    double eps_re[9];
    double aniso_eo, aniso_ee, aniso_axis[3];
    double curv_y, cc_y;
    double asphere_scale;
};
static_assert(sizeof(prt_dev_surface) < 512, "device surface record grew beyond 512 bytes");

// The HOT BLOCK of a surface record: the fields every step of a march through conic surfaces reads, packed into
// 128 bytes that a wave fetches with two s_load_dwordx16 -- ONE scalar-memory round trip per step of the crystal
// march (k_trace_general), issued at the top of the step, instead of one s_load -> s_waitcnt round trip per field and
// basic block (the compiler places a scalar load in the block that uses it: ~20 dependent round trips per step).
// Everything else (rotation matrices of tilted frames, the epsilon tensor of biaxial crystals, coefficient
// pointers) stays in the full record and is fetched where it is needed.
//   v[0]  two int32: bits (shape_type | ap_type << 4 | interaction << 6 | mat_type << 7 | aniso_class << 8 |
//         frame_flags << 10), newton_maxit
//   v[1..2] curv, cc   v[3..5] g_shape   v[6..7] ap_p0, ap_p1   v[8] n_after
//   v[9..10] aniso_eo, aniso_ee   v[11..13] aniso_axis   v[14..15] spare
#define PRT_HOT_DOUBLES 16
struct prt_hot_surface {
    double v[PRT_HOT_DOUBLES];
};
static_assert(sizeof(prt_hot_surface) == 128, "hot block = two s_load_dwordx16");

// the hot block unpacked into (scalar) registers; same member names as prt_dev_surface, so that the per-ray device
// functions below take either (`cold()` leads to the fields only the full record has)
struct hot_rec {
    int32_t shape_type, ap_type, interaction, mat_type, frame_flags, aniso_class, newton_maxit;
    double curv, cc, g_shape[3], ap_p0, ap_p1, n_after, aniso_eo, aniso_ee, aniso_axis[3];
    const prt_dev_surface *full;
};
__device__ __forceinline__ const prt_dev_surface *cold(const prt_dev_surface *sf) { return sf; }
__device__ __forceinline__ const prt_dev_surface *cold(const hot_rec *sf) { return sf->full; }

struct vec3 {
    double x, y, z;
};

PRT_DEV vec3 v3(double x, double y, double z) { return vec3{x, y, z}; }
PRT_DEV double dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// r = B v            (localcoordinates.py:391-396)
PRT_DEV vec3 mat_vec(const double *__restrict__ B, const vec3 &v) {
    return v3(B[0] * v.x + B[1] * v.y + B[2] * v.z, B[3] * v.x + B[4] * v.y + B[5] * v.z,
              B[6] * v.x + B[7] * v.y + B[8] * v.z);
}
// r = B^T v          (localcoordinates.py:406-413)
PRT_DEV vec3 matT_vec(const double *__restrict__ B, const vec3 &v) {
    return v3(B[0] * v.x + B[3] * v.y + B[6] * v.z, B[1] * v.x + B[4] * v.y + B[7] * v.z,
              B[2] * v.x + B[5] * v.y + B[8] * v.z);
}

// 1/a to ~1 ulp: v_rcp_f64 (2^-24 relative) + two Newton steps.  Replaces the IEEE
// division sequence (div_scale x2, rcp, 5 fma, div_fmas, div_fixup) on the per-ray
// critical path; operands here are O(1e-6 .. 1e6), far from the exponent limits the
// scaling steps of the IEEE sequence exist for.  0, inf and NaN behave like 1/a.
PRT_DEV double fast_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-a, r, 1.0);
    // the last step would turn r = inf (a = 0) into NaN; keep the hardware result there
    return isfinite(r) ? __builtin_fma(r, e, r) : r;
}

// sqrt(x) to ~1 ulp for x in the normal range: v_rsq_f64 (2^-24 relative) -> one coupled
// Goldschmidt step (error ~2^-47) -> two Newton residual corrections.  hipcc's sqrt() adds
// input scaling (ldexp / class tests) for denormal and huge arguments that per-ray optical
// quantities (O(1e-12 .. 1e12)) never reach; x = 0 is handled explicitly, x < 0 gives NaN
// like sqrt.
PRT_DEV double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0.0) ? 0.0 : g;
}

// 1/sqrt(x) to ~1 ulp: v_rsq_f64 + two Newton steps  r <- r (1.5 - 0.5 x r^2)
PRT_DEV double fast_rsqrt(double x) {
    const double r0 = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double r = r0 * __builtin_fma(-h * r0, r0, 1.5);
    r = r * __builtin_fma(-h * r, r, 1.5);
    return r;
}

// ---------------------------------------------------------------------------
// first-segment direction: RayBundle.returnKtoD, ray.py:136-152
//   S = Re(|E|^2 k - (E.k) conj(E)),  d = S/|S|      (k real)
// ---------------------------------------------------------------------------
PRT_DEV vec3 poynting_dir(const vec3 &k, const vec3 &er, const vec3 &ei) {
    const double e2 = dot(er, er) + dot(ei, ei);
    const double ekr = dot(er, k), eki = dot(ei, k);
    // Re((ekr + i eki)(er - i ei)) = ekr*er + eki*ei
    vec3 s = v3(e2 * k.x - (ekr * er.x + eki * ei.x), e2 * k.y - (ekr * er.y + eki * ei.y),
                e2 * k.z - (ekr * er.z + eki * ei.z));
    const double inv = fast_rcp(fast_sqrt(dot(s, s)));
    return v3(s.x * inv, s.y * inv, s.z * inv);
}

PRT_DEV vec3 normalized(const vec3 &k) {
    const double inv = fast_rcp(fast_sqrt(dot(k, k)));
    return v3(k.x * inv, k.y * inv, k.z * inv);
}

// ---------------------------------------------------------------------------
// shapes
// ---------------------------------------------------------------------------

// Conic.intersect, surface_shape.py:305-321.  r0, d in the shape frame.
// d need not be a unit vector: with d2 = d.d the reference's H = -c - cc c dz^2 (written for
// |d| = 1) generalises to H = -c d2 - cc c dz^2 and the returned t is measured in units of
// |d| (the hit point r0 + t d is the same).  The fused march passes d = k (|k| = n) and
// saves the normalisation.
PRT_DEV double conic_t(double c, double cc, const vec3 &r0, const vec3 &d, double d2, bool &ok) {
    double F, G, H;
    if (cc == 0.0) {  // sphere / plane: wave-uniform branch, drops the (1+cc) and cc terms
        F = d.z - c * (d.x * r0.x + d.y * r0.y + d.z * r0.z);
        G = c * (r0.x * r0.x + r0.y * r0.y + r0.z * r0.z) - 2.0 * r0.z;
        H = -c * d2;
    } else {
        F = d.z - c * (d.x * r0.x + d.y * r0.y + d.z * r0.z * (1.0 + cc));
        G = c * (r0.x * r0.x + r0.y * r0.y + r0.z * r0.z * (1.0 + cc)) - 2.0 * r0.z;
        H = -c * d2 - cc * c * d.z * d.z;
    }
    const double square = F * F + H * G;
    ok = square >= 0.0;
    return G * fast_rcp(F + fast_sqrt(square));
}

// Conic.getGrad + conic_function, surface_shape.py:208-237:
//   z = c r2/(1+sq), sq = sqrt(1-(1+cc)c^2 r2);  grad = (-c x, -c y, 1 - c z (1+cc))
//   == (-c x, -c y, sq)   because c z (1+cc) = (1-sq^2)/(1+sq) = 1 - sq.
// Outside the conic's domain (1-(1+cc)c^2 r2 <= 0) the reference's sag is NaN.
PRT_DEV vec3 conic_grad(double c, double cc, double x, double y) {
    const double r2 = x * x + y * y;
    const double st = 1.0 - (1.0 + cc) * c * c * r2;
    // (fast_sqrt: ~1 ulp without the IEEE sequence's scaling steps; this gradient feeds the surface normal of the
    // interactions that evaluate the shape at a given point -- crystal interfaces, the per-surface entry points)
    const double sq = (st > 0.0) ? fast_sqrt(st) : __builtin_nan("");
    return v3(-c * x, -c * y, sq);
}

// The same gradient for a point p that lies ON the conic (the hit point the intersection just
// produced): there c z (1+cc) with z = p.z needs no sag evaluation, and
//   (1 - c (1+cc) z)^2 = 1 - (1+cc) c^2 r2        (conic equation c r2 - 2z + c(1+cc)z^2 = 0)
// so gz = |1 - c (1+cc) p.z| equals the reference's +sqrt branch on either sheet.  The domain
// test (NaN when 1-(1+cc)c^2 r2 <= 0) is kept.  |grad|^2 = 1 - cc c^2 r2  (== 1 for spheres).
PRT_DEV vec3 conic_grad_on_surface(double c, double cc, const vec3 &p, double &norm2) {
    const double r2 = p.x * p.x + p.y * p.y;
    const double c2r2 = c * c * r2;
    const double st = 1.0 - (1.0 + cc) * c2r2;
    double gz = fabs(1.0 - c * (1.0 + cc) * p.z);
    if (!(st > 0.0)) gz = __builtin_nan("");
    norm2 = 1.0 - cc * c2r2;
    return v3(-c * p.x, -c * p.y, gz);
}

PRT_DEV double conic_sag(double c, double cc, double r2) {
    const double st = 1.0 - (1.0 + cc) * c * c * r2;
    if (!(st > 0.0)) return __builtin_nan("");
    return c * r2 / (1.0 + sqrt(st));
}

// The side array (coefficients, term powers, spline knots) is written by the host before the launch and only
// read by the kernels.  Read through the CONSTANT address space, an access whose index every lane of the wave
// shares becomes a scalar load (s_load into SGPRs through the scalar cache).  Read through the generic pointer
// stored in the record it was a flat_load per lane + s_waitcnt vmcnt(0) per coefficient and Newton step -- a
// vector-memory round trip each, and on gfx950 vmcnt counts the wave's outstanding path stores too, so each of
// those waits also drained the stores.
#define PRT_CONST_AS __attribute__((address_space(4)))
#define PRT_GLOBAL_AS __attribute__((address_space(1)))
typedef const PRT_CONST_AS double *prt_cdoubles;
PRT_DEV prt_cdoubles side_doubles(const prt_dev_surface *__restrict__ sf) {
    return (prt_cdoubles)(uint64_t)sf->coeffs;
}

// The first PRT_ASPHERE_PREFETCH even-asphere coefficients a_n and the products b_n = (n+1) a_n of one surface,
// fetched ONCE per surface (two s_load_dwordx16) and held in scalar registers across the Newton iteration.
// Side-array layout of a surface with an asphere part (prt_system_create): the coefficients as the host
// table has them (n_coeffs doubles: a_0.., for a combination followed by its polynomial terms), then the
// b_n of the asphere part.  The array is padded, so both 8-slot reads are always in bounds; slots >= nc are
// forced to zero, which makes the fixed-length Horner below exact for any nc <= 8.
#define PRT_ASPHERE_PREFETCH 8
struct asphere_coeffs {
    double a[PRT_ASPHERE_PREFETCH], b[PRT_ASPHERE_PREFETCH];
};
PRT_DEV void asphere_prefetch(const prt_dev_surface *__restrict__ sf, int nc, asphere_coeffs &ac) {
    prt_cdoubles cf = side_doubles(sf);
    prt_cdoubles cb = cf + sf->n_coeffs;
#pragma unroll
    for (int n = 0; n < PRT_ASPHERE_PREFETCH; ++n) {
        const double va = cf[n], vb = cb[n];
        ac.a[n] = (n < nc) ? va : 0.0;
        ac.b[n] = (n < nc) ? vb : 0.0;
    }
}

// Asphere.F and Asphere.gradF, surface_shape.py:529-555, in one pass:
//   F  = c r2/(1+sq) + sum_n a_n r2^(n+1)
//   Fx = x (c/sq + sum_n 2(n+1) a_n r2^n),   gradient of z-F = (-Fx, -Fy, 1)
// Horner in r2 (the reference sums the powers; same polynomial).
// PRE: the first 8 coefficient pairs come from `ac` (scalar registers, the asphere-only kernels); otherwise
// every pair is a scalar load inside the loop (the all-shapes kernels have no registers to spare for `ac`).
// The sag comes back as a FRACTION, F = Fn / Fd with Fd = 1 + sq (round 5): the Newton step needs g / g' =
// (z Fd - Fn) / (Fd g'), ONE reciprocal instead of the two of c r2 / (1 + sq) followed by g / g' -- a quarter-rate
// instruction and its refinement less per evaluation; callers that want the sag itself divide (shape_sag).
template <bool PRE>
PRT_DEV void asphere_eval(const prt_dev_surface *__restrict__ sf, int nc, const asphere_coeffs &ac, double x,
                          double y, double &Fn, double &Fd, double &dFdr2x2 /* Fx = x * this */) {
    const double c = sf->curv, cc = sf->cc;
    const double r2 = x * x + y * y;
    // sq and 1/sq from ONE reciprocal square root (v_rsq_f64 + two Newton steps) instead of a square root and a
    // reciprocal sequence: image mode of BASELINE configs[2] 0.266 -> 0.251 ms (benchmarks/ab_builds.py,
    // profiles/r03b_ab_asphere_eval.json; hit points move by at most 4e-14 mm)
    const double st = 1.0 - c * c * (1.0 + cc) * r2;
    const double isq = fast_rsqrt(st);
    const double sq = st * isq;
    double p = 0.0, dp = 0.0;  // p = sum a_n r2^n ; dp = sum (n+1) a_n r2^n
    prt_cdoubles cf = side_doubles(sf);
    prt_cdoubles cb = cf + sf->n_coeffs;
    for (int n = nc - 1; n >= (PRE ? PRT_ASPHERE_PREFETCH : 0); --n) {
        p = p * r2 + cf[n];
        dp = dp * r2 + cb[n];
    }
    if (PRE) {
        // (a_n = b_n = 0 for n >= nc: the upper half of the fixed-length Horner does nothing for the usual 2..4
        // coefficients -- a wave-uniform branch skips it: 0.266 -> 0.258 ms in image mode, both changes 0.241;
        // path mode 0.358 -> 0.348 ms)
        if (nc > PRT_ASPHERE_PREFETCH / 2) {
#pragma unroll
            for (int n = PRT_ASPHERE_PREFETCH - 1; n >= PRT_ASPHERE_PREFETCH / 2; --n) {
                p = p * r2 + ac.a[n];
                dp = dp * r2 + ac.b[n];
            }
        }
#pragma unroll
        for (int n = PRT_ASPHERE_PREFETCH / 2 - 1; n >= 0; --n) {
            p = p * r2 + ac.a[n];
            dp = dp * r2 + ac.b[n];
        }
    }
    Fd = 1.0 + sq;
    Fn = r2 * __builtin_fma(p, Fd, c);      // c r2 / (1 + sq) + p r2 = r2 (c + p (1 + sq)) / (1 + sq)
    dFdr2x2 = c * isq + 2.0 * dp;
}

// XYPolynomials.F / gradF, surface_shape.py:785-807: F = sum c_ij x^i y^j (c_ij already divided by
// normradius^(i+j) on the host), as a nested Horner scheme over the DENSE rows of the coefficient triangle:
//   F = sum_i x^i P_i(y),   P_i(y) = sum_j c_ij y^j
//   rows i = D .. 0:   (P_i, P_i') by Horner in y with derivative;   Fx <- Fx x + F;  F <- F x + P_i;  Fy <- Fy x + P_i'
// Two FMAs per coefficient and three per row, no data-dependent control flow, no powers to keep -- the
// monomial-by-monomial sum it replaces took six VALU operations per term plus the running powers
// (42 monomials up to degree 8: 1.13 -> 0.6x ms at 1e7 rays; |difference| of the two evaluations on the
// golden Zernike / polynomial cases: 3e-17).  Side-array layout (prt_system_create, poly_rows):
//   int32 n_rows, int32 n_chunks, int32 chunks_of_row[n_rows] (row D first), padded to 8 bytes;
//   then 32-byte chunks of four coefficients in Horner order (highest power of y first, a row's first
//   chunk padded with leading zeros).  One s_load_dwordx8 per chunk, issued one chunk ahead.
//   In front of all that: the DENSE TRIANGLE of total degree <= PRT_POLY_DENSE_DEG, PRT_POLY_DENSE_SLOTS doubles
//   (rows i = D .. 0, row i holding c_i,D-i .. c_i,0 -- Horner order --, absent terms zero; all zero and unused
//   when the polynomial does not fit): what dense_poly_eval keeps in scalar registers.
#define PRT_POLY_DENSE_DEG 4
#define PRT_POLY_DENSE_COEFFS ((PRT_POLY_DENSE_DEG + 1) * (PRT_POLY_DENSE_DEG + 2) / 2)
#define PRT_POLY_DENSE_SLOTS 16
typedef const PRT_CONST_AS int32_t *prt_cints;
PRT_DEV void xypoly_eval(const prt_dev_surface *__restrict__ sf, double x, double y, double &F, double &Fx,
                         double &Fy) {
    prt_cints hd = (prt_cints)((uint64_t)sf->pows + 8 * PRT_POLY_DENSE_SLOTS);
    const int nrows = hd[0];
    prt_cdoubles cd = (prt_cdoubles)((uint64_t)sf->pows + 8 * PRT_POLY_DENSE_SLOTS + 8 * (uint64_t)((nrows + 3) >> 1));
    F = 0.0;
    Fx = 0.0;
    Fy = 0.0;
    // (the side array is padded: the chunk behind the last one exists, its content is never used)
    double c0 = cd[0], c1 = cd[1], c2 = cd[2], c3 = cd[3];
    int ch = 0;
    for (int r = 0; r < nrows; ++r) {
        const int nch = hd[2 + r];
        double p = 0.0, dp = 0.0;
        for (int q = 0; q < nch; ++q) {
            ++ch;
            const double n0 = cd[4 * ch], n1 = cd[4 * ch + 1], n2 = cd[4 * ch + 2], n3 = cd[4 * ch + 3];
            dp = dp * y + p;
            p = p * y + c0;
            dp = dp * y + p;
            p = p * y + c1;
            dp = dp * y + p;
            p = p * y + c2;
            dp = dp * y + p;
            p = p * y + c3;
            c0 = n0;
            c1 = n1;
            c2 = n2;
            c3 = n3;
        }
        Fx = Fx * x + F;
        F = F * x + p;
        Fy = Fy * x + dp;
    }
}

// The same polynomial when it fits the dense triangle of total degree <= PRT_POLY_DENSE_DEG (12-term freeforms,
// low-order Zernike expansions): all coefficients fetched ONCE per surface (two s_load_dwordx16) and held in
// scalar registers across the Newton iteration, like the first asphere coefficients; the loops have
// compile-time bounds, so nothing but FMAs remains per evaluation -- no scalar-load round trip per chunk.  Same
// Horner order as xypoly_eval (whose padding zeros leave p and dp untouched): bit-identical results.
struct poly_coeffs {
    double c[PRT_POLY_DENSE_COEFFS];
};
PRT_DEV void dense_poly_prefetch(const prt_dev_surface *__restrict__ sf, poly_coeffs &pc) {
    prt_cdoubles cd = (prt_cdoubles)(uint64_t)sf->pows;
#pragma unroll
    for (int q = 0; q < PRT_POLY_DENSE_COEFFS; ++q) pc.c[q] = cd[q];
}
PRT_DEV void dense_poly_eval(const poly_coeffs &pc, double x, double y, double &F, double &Fx, double &Fy) {
    F = 0.0;
    Fx = 0.0;
    Fy = 0.0;
    int q = 0;
#pragma unroll
    for (int i = PRT_POLY_DENSE_DEG; i >= 0; --i) {
        double p = 0.0, dp = 0.0;
#pragma unroll
        for (int j = PRT_POLY_DENSE_DEG - i; j >= 0; --j) {
            dp = dp * y + p;
            p = p * y + pc.c[q++];
        }
        Fx = Fx * x + F;
        F = F * x + p;
        Fy = Fy * x + dp;
    }
}

// Biconic.F / gradF, surface_shape.py:618-647:
//   F = (cx x^2 + cy y^2)/(1+sq) + sum_n a_n w_n^(n+1),  w_n = r2 - b_n (x^2 - y^2)
//   sq = sqrt(1 - cx^2 (1+ccx) x^2 - cy^2 (1+ccy) y^2)
//   Fx = cx x (cx (1+ccx) u + 2 (sq+1) sq) / ((sq+1)^2 sq) + sum 2 a_n (n+1) x (1-b_n) w_n^n   (u = cx x^2 + cy y^2)
PRT_DEV void biconic_eval(const prt_dev_surface *__restrict__ sf, double x, double y, double &F,
                          double &Fx, double &Fy) {
    const double cx = sf->curv, ccx = sf->cc, cy = sf->curv_y, ccy = sf->cc_y;
    const double x2 = x * x, y2 = y * y;
    const double u = cx * x2 + cy * y2;
    const double st = 1.0 - cx * cx * (1.0 + ccx) * x2 - cy * cy * (1.0 + ccy) * y2;
    const double isq = fast_rsqrt(st);  // sq and 1/sq from one reciprocal square root (as in asphere_eval)
    const double sq = st * isq;
    const double den = 1.0 + sq;
    const double iden = fast_rcp(den);
    F = u * iden;
    const double common = iden * iden * isq;
    const double two_den_sq = 2.0 * den * sq;
    Fx = cx * x * (cx * (ccx + 1.0) * u + two_den_sq) * common;
    Fy = cy * y * (cy * (ccy + 1.0) * u + two_den_sq) * common;
    const int np = sf->n_coeffs;
    prt_cdoubles cf = side_doubles(sf);
    for (int n = 0; n < np; ++n) {
        const double a = cf[2 * n], b = cf[2 * n + 1];
        const double w = x2 * (1.0 - b) + y2 * (1.0 + b);
        double wn = 1.0;  // w^n
        for (int q = 0; q < n; ++q) wn *= w;
        F += a * wn * w;
        const double d = 2.0 * a * (double)(n + 1) * wn;
        Fx += d * x * (1.0 - b);
        Fy += d * y * (1.0 + b);
    }
}

// GridSag.F / gradF, surface_shape.py:865-873: scipy's RectBivariateSpline.ev(x, y[, dx | dy]) =
// FITPACK bispev / parder on the (tx, ty, c) form.  Arguments outside the knot range are clamped to
// it (bispev); the knot interval is located by bisection; the four non-zero cubic basis functions
// and their derivatives come from the Cox-de Boor recurrence (FITPACK fpbspl).
typedef const PRT_GLOBAL_AS double *prt_gdoubles;
PRT_DEV int bspline_interval(prt_gdoubles t, int n, double x) {
    int lo = 3, hi = n - 5;  // t[lo] <= x < t[lo+1], lo in [3, n-5]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (x >= t[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}

PRT_DEV void bspline_basis3(prt_gdoubles t, int l, double x, double b[4], double db[4]) {
    // degree 1
    const double t0 = t[l], t1 = t[l + 1];
    const double i1 = fast_rcp(t1 - t0);
    const double b1_0 = (t1 - x) * i1, b1_1 = (x - t0) * i1;
    // degree 2 (three functions on knots l-1 .. l+2)
    const double tm1 = t[l - 1], t2 = t[l + 2];
    const double f0 = b1_0 * fast_rcp(t1 - tm1), f1 = b1_1 * fast_rcp(t2 - t0);
    const double b2_0 = f0 * (t1 - x);
    const double b2_1 = f0 * (x - tm1) + f1 * (t2 - x);
    const double b2_2 = f1 * (x - t0);
    // degree 3 and its derivative: B'_{i,3} = 3 (B_{i,2}/(t_{i+3}-t_i) - B_{i+1,2}/(t_{i+4}-t_{i+1}))
    const double tm2 = t[l - 2], t3 = t[l + 3];
    const double g0 = b2_0 * fast_rcp(t1 - tm2), g1 = b2_1 * fast_rcp(t2 - tm1), g2 = b2_2 * fast_rcp(t3 - t0);
    b[0] = g0 * (t1 - x);
    b[1] = g0 * (x - tm2) + g1 * (t2 - x);
    b[2] = g1 * (x - tm1) + g2 * (t3 - x);
    b[3] = g2 * (x - t0);
    db[0] = -3.0 * g0;
    db[1] = 3.0 * (g0 - g1);
    db[2] = 3.0 * (g1 - g2);
    db[3] = 3.0 * g2;
}

PRT_DEV void gridsag_eval(const prt_dev_surface *__restrict__ sf, double x, double y, double &F,
                          double &Fx, double &Fy) {
    const int nx = sf->grid_nx, ny = sf->grid_ny;
    // knots and coefficients are indexed per ray: vector loads, through the GLOBAL address space (the generic
    // pointer of the record would make them flat loads); the four range limits are wave-uniform: scalar loads
    prt_gdoubles tx = (prt_gdoubles)(uint64_t)sf->coeffs;
    prt_gdoubles ty = tx + nx;
    prt_gdoubles c = ty + ny;
    prt_cdoubles ux = side_doubles(sf);
    const double xc = fmin(fmax(x, ux[3]), ux[nx - 4]);
    const double yc = fmin(fmax(y, ux[nx + 3]), ux[nx + ny - 4]);
    const int lx = bspline_interval(tx, nx, xc), ly = bspline_interval(ty, ny, yc);
    double bx[4], dbx[4], by[4], dby[4];
    bspline_basis3(tx, lx, xc, bx, dbx);
    bspline_basis3(ty, ly, yc, by, dby);
    const int ncy = ny - 4;
    F = 0.0;
    Fx = 0.0;
    Fy = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        prt_gdoubles row = c + (int64_t)(lx - 3 + i) * ncy + (ly - 3);
        double s = 0.0, sy = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double v = row[j];
            s += by[j] * v;
            sy += dby[j] * v;
        }
        F += bx[i] * s;
        Fx += dbx[i] * s;
        Fy += bx[i] * sy;
    }
}

// Which shape code a kernel instantiation carries (the host picks the level from the table's content):
// 0: conics only; 1: conics and even aspheres; 2: conics, even aspheres, XY polynomials (also the monomial
// expansions of Zernike surfaces) and biconics -- the explicit shapes BASELINE's north_star names plus SURVEY
// 8 f3 --, all of them fed by scalar loads only; 3: every shape (adds the sag grid, whose spline data is
// indexed per ray = vector loads inside the Newton loop, and the asphere + polynomial combination).
// Fewer shapes = fewer VGPRs = more waves.
#define PRT_SHAPES_CONIC 0
#define PRT_SHAPES_ASPHERE 1
#define PRT_SHAPES_POLY 2
#define PRT_SHAPES_ALL 3

// explicit z = F(x,y) shapes: value and in-plane derivatives
// what explicit_eval wants fetched once per surface: the first asphere coefficient pairs (asphere-only kernels),
// a dense polynomial (the "conics + aspheres + polynomials + biconics" kernels); nothing otherwise
struct explicit_prefetched {
    asphere_coeffs ac;
    poly_coeffs pc;
    bool dense_poly;
};
// (sag = F / Fd: Fd = 1 for every shape but the even asphere, see asphere_eval)
template <int SHAPES = PRT_SHAPES_ALL>
PRT_DEV void explicit_eval(const prt_dev_surface *__restrict__ sf, const explicit_prefetched &pre, double x, double y,
                           double &F, double &Fd, double &Fx, double &Fy) {
    const asphere_coeffs &ac = pre.ac;
    Fd = 1.0;
    if (SHAPES == PRT_SHAPES_ASPHERE || sf->shape_type == PRT_SHAPE_ASPHERE) {
        double m;
        asphere_eval<SHAPES == PRT_SHAPES_ASPHERE>(sf, sf->n_coeffs, ac, x, y, F, Fd, m);
        Fx = x * m;
        Fy = y * m;
    } else if (sf->shape_type == PRT_SHAPE_BICONIC) {
        biconic_eval(sf, x, y, F, Fx, Fy);
    } else if (SHAPES == PRT_SHAPES_ALL && sf->shape_type == PRT_SHAPE_GRIDSAG) {
        gridsag_eval(sf, x, y, F, Fx, Fy);
    } else if (SHAPES == PRT_SHAPES_ALL && sf->shape_type == PRT_SHAPE_COMBO) {
        // LinearCombination.F / gradF (surface_shape.py:713-748) of one conic / asphere part and
        // polynomial parts, merged by the host: scale * asphere + sum c_ij x^i y^j
        double Fan, Fad, m;
        asphere_eval<false>(sf, sf->n_asphere, ac, x, y, Fan, Fad, m);
        const double Fa = Fan * fast_rcp(Fad);
        xypoly_eval(sf, x, y, F, Fx, Fy);
        const double sc = sf->asphere_scale;
        F += sc * Fa;
        Fx += sc * x * m;
        Fy += sc * y * m;
    } else if (SHAPES == PRT_SHAPES_POLY && pre.dense_poly) {
        dense_poly_eval(pre.pc, x, y, F, Fx, Fy);
    } else {
        xypoly_eval(sf, x, y, F, Fx, Fy);
    }
}

template <int SHAPES = PRT_SHAPES_ALL>
PRT_DEV void explicit_prefetch(const prt_dev_surface *__restrict__ sf, explicit_prefetched &pre) {
    pre.dense_poly = false;
    if (SHAPES == PRT_SHAPES_ASPHERE) asphere_prefetch(sf, sf->n_coeffs, pre.ac);
    if (SHAPES == PRT_SHAPES_POLY && sf->shape_type == PRT_SHAPE_XYPOLY && sf->poly_dense) {
        dense_poly_prefetch(sf, pre.pc);
        pre.dense_poly = true;
    }
}

// ExplicitShape.intersect, surface_shape.py:448-465: root of
//   g(t) = r0z + t dz - F(r0x + t dx, r0y + t dy), start t = 0.
// The reference gives the N-vector to MINPACK hybrd (xtol 1e-6); here every ray runs scalar Newton to machine
// precision.  The loop is wave-uniform: a wave leaves when all its lanes have converged (or the cap is hit);
// converged lanes keep their t.  *nonconv reports lanes that hit the cap while still moving (last step > 1e-11).
//
// When is a lane done (round 5)?  Newton converges quadratically: behind a step dt the error is K dt^2 with
// K = g''/2g', about half the curvature along the ray.  Until round 4 a lane was done when a step came out <= 1e-15
// (relative to max(1, |t|)) -- the evaluation that produced that step only OBSERVED that the step before it had
// already arrived: a quarter of the work of the four evaluations a ray of BASELINE configs[2] takes (steps 1,
// 1e-3, 1e-9, 1e-17 of |t|).  Now a lane is done behind the first step <= 1e-8 that is also <= 1e-3 of the step
// before it (the signature of quadratic convergence; grazing incidence up to 84 degrees still shows it:
// tests/golden/asphere_grazing_field30_tight.npz): what is left, K (1e-8 |t|)^2, is
// below the rounding of t for K |t| <= 1 (a ray that flies a hundred curvature radii to its surface is the limit;
// beyond it the error grows like K |t| 1e-16 -- still five digits inside the parity bar).  The FIRST evaluation
// never ends the iteration that way (a start point that happens to lie 1e-9 off the surface gets its second one).
//
// gx, gy: the in-plane derivatives (Fx, Fy) AT THE ROOT, for the normal, without another evaluation of the shape:
// the derivatives of the last evaluation -- taken up to 1e-8 |t| in front of the root -- moved there along the
// secant through the evaluation before it (the two evaluation points lie dt_prev apart on the ray):
//   Fx(root) = Fx_n + (Fx_n - Fx_prev) dt_n / dt_prev      (error: third derivative x dt_n x dt_prev)
// Without it the normal would be off by curvature x 1e-8 |t| (6e-10 on configs[2]): outside the 1e-10 bar.
// The step needs 1/g' only as accurately as the step it scales: v_rcp_f64 + ONE Newton step (2^-48).
// The convergence scale is in units of |d| (d may be k, |k| = n ~ 1..2).
PRT_DEV double newton_rcp(double a) {
    const double r = __builtin_amdgcn_rcp(a);
    return __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
}

template <int SHAPES = PRT_SHAPES_ALL>
PRT_DEV double explicit_t(const prt_dev_surface *__restrict__ sf, const vec3 &r0, const vec3 &d,
                          bool &nonconv, double &gx, double &gy) {
    double t = 0.0;
    bool done = false;
    bool at_noise_floor = false;   // the last step was <= 1e-11 (relative): see below
    gx = 0.0;
    gy = 0.0;
    double fx_prev = 0.0, fy_prev = 0.0, dt_prev = 1.0;
    const int maxit = sf->newton_maxit > 0 ? sf->newton_maxit : 30;
    explicit_prefetched pre;
    explicit_prefetch<SHAPES>(sf, pre);
    for (int it = 0; it < maxit; ++it) {
        double F, Fd, Fx, Fy;
        const double px = r0.x + t * d.x, py = r0.y + t * d.y;
        explicit_eval<SHAPES>(sf, pre, px, py, F, Fd, Fx, Fy);
        // g / g' with the sag as the fraction F / Fd:  (z Fd - F) / (Fd g')
        const double gn = __builtin_fma(r0.z + t * d.z, Fd, -F);
        const double gp = d.z - Fx * d.x - Fy * d.y;
        const double dt = gn * newton_rcp(gp * Fd);
        // (first evaluation: nothing in front of it -- fx_prev := Fx makes the correction vanish; a lane that is done
        //  keeps t, gx, gy -- a ray's result must not depend on its neighbours in the wave --, the rest is scratch)
        const double ratio = dt * __builtin_amdgcn_rcp(dt_prev);
        const bool contracted = it > 0 && fabs(dt) <= 1e-3 * fabs(dt_prev);
        const double gxn = __builtin_fma(Fx - (it > 0 ? fx_prev : Fx), ratio, Fx);
        const double gyn = __builtin_fma(Fy - (it > 0 ? fy_prev : Fy), ratio, Fy);
        const double tn = t - dt;
        fx_prev = Fx;
        fy_prev = Fy;
        dt_prev = dt;
        const double adt = fabs(dt), scale = fmax(1.0, fabs(tn));
        if (!done) {
            t = tn;
            gx = gxn;
            gy = gyn;
            // a NaN step stops the lane too (t is NaN already: the ray is invalid later); an infinite one makes the next NaN
            // The 1e-8 exit is taken only behind a step that CONTRACTED like Newton does at a simple root (|dt| <= 1e-3
            // |dt_prev|: quadratic convergence that ends at 1e-8 comes from 1e-4 or better).  A lane that creeps towards
            // a (near-)double root -- steps halving, the remaining error as large as the step itself -- keeps
            // iterating to the 1e-15 rule or to the cap, where it is reported (round 6).
            // (two compares against wave-uniform thresholds, combined as lane masks: a per-lane threshold would be two
            //  more vector registers in a kernel that sits at its 96-register cap)
            done = !(adt > 1e-8 * scale) && (contracted || !(adt > 1e-15 * scale));
            at_noise_floor = !(adt > 1e-11 * scale);
        }
        if (__all(done)) break;
    }
    // A lane that reaches the cap while its steps sit at the rounding noise of g / g' HAS converged, to the 1e-11 its last
    // step shows: it keeps its t.  (With the 1e-8 rule such a lane is done at its second evaluation anyway -- the noisy
    // biconic of demos/demo_hud.py, whose steps stall at 1e-14 --; the flag still decides for a cap of ONE evaluation.)
    // Only lanes that are still moving at the cap are reported.
    nonconv = !done && !at_noise_floor;
    return t;
}

// gradient of the implicit surface function in the shape frame (not normalised)
PRT_DEV vec3 shape_grad(const prt_dev_surface *__restrict__ sf, double x, double y) {
    if (sf->shape_type == PRT_SHAPE_CONIC) return conic_grad(sf->curv, sf->cc, x, y);
    double F, Fd, Fx, Fy;
    explicit_prefetched pre;
    explicit_prefetch(sf, pre);
    explicit_eval(sf, pre, x, y, F, Fd, Fx, Fy);
    return v3(-Fx, -Fy, 1.0);
}

PRT_DEV double shape_sag(const prt_dev_surface *__restrict__ sf, double x, double y) {
    if (sf->shape_type == PRT_SHAPE_CONIC) return conic_sag(sf->curv, sf->cc, x * x + y * y);
    double F, Fd, Fx, Fy;
    explicit_prefetched pre;
    explicit_prefetch(sf, pre);
    explicit_eval(sf, pre, x, y, F, Fd, Fx, Fy);
    return F * fast_rcp(Fd);
}

// ---------------------------------------------------------------------------
// aperture.py:71-139
// ---------------------------------------------------------------------------
template <class REC>
PRT_DEV bool aperture_ok(const REC *__restrict__ sf, double x, double y) {
    if (sf->ap_type == PRT_AP_CIRCULAR) {
        const double r2 = x * x + y * y;
        return (r2 >= sf->ap_p0 * sf->ap_p0) && (r2 <= sf->ap_p1 * sf->ap_p1);
    }
    if (sf->ap_type == PRT_AP_RECTANGULAR) {
        const double hw = sf->ap_p0 * 0.5, hh = sf->ap_p1 * 0.5;
        return (x >= -hw) && (x <= hw) && (y >= -hh) && (y <= hh);
    }
    return true;
}

// ---------------------------------------------------------------------------
// Material.propagate -> Surface.intersect (surface.py:116-135)
//   in : x global start point, d global unit direction
//   out: xh global hit point, p hit point in the shape frame, valid &= hit & aperture
// ---------------------------------------------------------------------------
//        g (unnormalised surface gradient at p, shape frame) and g2 = |g|^2 as by-products
//   d may be any positive multiple of the unit direction, d2 = d.d
template <int SHAPES = PRT_SHAPES_ALL, class REC>
PRT_DEV void propagate_step(const REC *__restrict__ sf, const vec3 &x, const vec3 &d,
                            double d2, vec3 &xh, vec3 &p, vec3 &g, double &g2, bool &valid, bool &nonconv) {
    nonconv = false;
    const int ff = sf->frame_flags;
    vec3 r0 = v3(x.x - sf->g_shape[0], x.y - sf->g_shape[1], x.z - sf->g_shape[2]);
    vec3 dl = d;
    if (!(ff & PRT_FRAME_SHAPE_IDENTITY)) {
        r0 = matT_vec(cold(sf)->B_shape, r0);
        dl = matT_vec(cold(sf)->B_shape, d);
    }
    double t;
    if (SHAPES == PRT_SHAPES_CONIC || sf->shape_type == PRT_SHAPE_CONIC) {
        bool ok;
        t = conic_t(sf->curv, sf->cc, r0, dl, d2, ok);
        valid = valid && ok;
        p = v3(r0.x + dl.x * t, r0.y + dl.y * t, r0.z + dl.z * t);
        g = conic_grad_on_surface(sf->curv, sf->cc, p, g2);
    } else {
        double fx, fy;
        t = explicit_t<SHAPES>(cold(sf), r0, dl, nonconv, fx, fy);  // reference: valid all True (surface_shape.py:462)
        // A ray whose Newton iteration hit the cap has no trustworthy hit point.  The mask after
        // propagate stays reference-compatible (True); the NaN hit point makes the normal NaN, so
        // the ray is dropped by the finite-normal test of the following refraction.  `nonconv` is
        // what tells such a ray from one that left the domain of the shape.
        if (nonconv) t = __builtin_nan("");
        p = v3(r0.x + dl.x * t, r0.y + dl.y * t, r0.z + dl.z * t);
        // (fx, fy are the derivatives at the last iterate: finite even when t is not -- poison them
        // so that the gradient is what the reference would compute AT the NaN point)
        if (!isfinite(t)) fx = __builtin_nan("");
        g = v3(-fx, -fy, 1.0);
        g2 = fx * fx + fy * fy + 1.0;
    }
    if (ff & PRT_FRAME_SHAPE_IDENTITY) {
        xh = v3(p.x + sf->g_shape[0], p.y + sf->g_shape[1], p.z + sf->g_shape[2]);
    } else {
        xh = mat_vec(cold(sf)->B_shape, p);
        xh.x += sf->g_shape[0];
        xh.y += sf->g_shape[1];
        xh.z += sf->g_shape[2];
    }
    if (sf->ap_type != PRT_AP_NONE) {
        vec3 pa = p;  // aperture frame == shape frame in the common case
        if (!(ff & PRT_FRAME_AP_IS_SHAPE)) {
            pa = matT_vec(cold(sf)->B_ap, v3(xh.x - cold(sf)->g_ap[0], xh.y - cold(sf)->g_ap[1], xh.z - cold(sf)->g_ap[2]));
        }
        valid = valid && aperture_ok(sf, pa.x, pa.y);
    }
}

// shape-frame hit point from a global one (used when interact is called on its own)
PRT_DEV vec3 to_shape_frame(const prt_dev_surface *__restrict__ sf, const vec3 &xh) {
    vec3 r = v3(xh.x - sf->g_shape[0], xh.y - sf->g_shape[1], xh.z - sf->g_shape[2]);
    if (!(sf->frame_flags & PRT_FRAME_SHAPE_IDENTITY)) r = matT_vec(cold(sf)->B_shape, r);
    return r;
}

// unit normal in the frame of the medium from the shape-frame gradient g (|g|^2 = g2):
// Shape.getNormal (surface_shape.py:100-112) + RayBundle.getLocalSurfaceNormal (ray.py:156-161).
// Spheres have |g| = 1 identically (conic_grad_on_surface): no normalisation.
template <int SHAPES = PRT_SHAPES_ALL, class REC>
PRT_DEV vec3 normal_from_grad(const REC *__restrict__ sf, const vec3 &g, double g2) {
    vec3 n = g;
    if (!((SHAPES == PRT_SHAPES_CONIC || sf->shape_type == PRT_SHAPE_CONIC) && sf->cc == 0.0)) {
        const double r = fast_rsqrt(g2);
        n = v3(g.x * r, g.y * r, g.z * r);
    }
    const int ff = sf->frame_flags;
    if (!(ff & PRT_FRAME_SHAPE_IDENTITY)) n = mat_vec(cold(sf)->B_shape, n);
    if (!(ff & PRT_FRAME_MAT_IDENTITY)) n = matT_vec(cold(sf)->B_mat, n);
    return n;
}

// the same for an arbitrary point of the shape frame (per-surface API: the caller's points
// need not lie on the surface, so the sag is evaluated like the reference does)
template <int SHAPES = PRT_SHAPES_ALL, class REC>
PRT_DEV vec3 normal_in_material_frame(const REC *__restrict__ sf, const vec3 &p) {
    vec3 g = (SHAPES == PRT_SHAPES_CONIC) ? conic_grad(sf->curv, sf->cc, p.x, p.y) : shape_grad(cold(sf), p.x, p.y);
    const double inv = fast_rsqrt(dot(g, g));
    vec3 n = v3(g.x * inv, g.y * inv, g.z * inv);
    const int ff = sf->frame_flags;
    if (!(ff & PRT_FRAME_SHAPE_IDENTITY)) n = mat_vec(cold(sf)->B_shape, n);
    if (!(ff & PRT_FRAME_MAT_IDENTITY)) n = matT_vec(cold(sf)->B_mat, n);
    return n;
}

// ---------------------------------------------------------------------------
// IsotropicMaterial.refract / reflect, material_isotropic.py:137-236
//   k global in -> k global out; valid &= (n2^2 - kin.kin > 0) & finite(normal)
// ---------------------------------------------------------------------------
template <class REC>
PRT_DEV void interact_isotropic(const REC *__restrict__ sf, const vec3 &n, vec3 &k,
                                bool &valid) {
    const bool mat_id = sf->frame_flags & PRT_FRAME_MAT_IDENTITY;
    vec3 k1 = k;
    if (!mat_id) k1 = matT_vec(cold(sf)->B_mat, k);
    const double kn = dot(k1, n);
    vec3 kin = v3(k1.x - kn * n.x, k1.y - kn * n.y, k1.z - kn * n.z);
    const double n2 = sf->n_after;
    const double square = n2 * n2 - dot(kin, kin);
    const double xi = fast_sqrt(square);
    // checkfinite(normal) (material_isotropic.py:173): k is finite for a valid ray, so the
    // normal is finite iff k.n is, and square inherits every NaN of kin
    valid = valid && (square > 0.0) && isfinite(kn);
    if (sf->interaction == PRT_MIRROR) kin = v3(-kin.x, -kin.y, -kin.z);  // :224
    vec3 k2 = v3(kin.x + xi * n.x, kin.y + xi * n.y, kin.z + xi * n.z);
    k = mat_id ? k2 : mat_vec(cold(sf)->B_mat, k2);
}
