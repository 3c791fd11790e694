// prt_aniso.h -- AnisotropicMaterial.refract / reflect per ray (device).
//
// Reference: material_anisotropic.py:70-155 -> MaxwellMaterial.sortKnormEField
// (material.py:122-153) -> calcKnormEfield (:98-108) -> calcXiEigenvectorsNorm
// (:407-454, per-ray scipy.linalg.eig of a 6x6 pencil) -> calcPoytingVectorNorm
// (:214-223).  Mathematically: with k = kpa + xi n (kpa the in-plane part of the
// incoming wave vector, n the unit normal, everything in the material frame)
//     W(xi) E = 0,   W = eps - (k.k) I + k k^T  ( = xi^2 M + xi C + K, :385-392 )
// has four finite solutions xi.  The reference takes them from LAPACK; here
//   * det W(xi) = p4 xi^4 + ... + p0 with the coefficients of
//     calcXiPolynomialNorm (material.py:501-566);
//   * eps = e I            : xi = +-sqrt(e - kpa.kpa), each double (closed form);
//   * uniaxial eps = eo I + (ee-eo) c c^T : the quartic factors exactly into the
//     ordinary sphere  xi^2 = eo - kpa.kpa  and the extraordinary ellipsoid
//     k^T eps k = eo ee  (a quadratic in xi) -- each well conditioned even where
//     the two sheets touch (the polynomial's double root is not);
//   * anything else (biaxial / non-symmetric real eps): Aberth-Ehrlich iteration
//     on the quartic in complex arithmetic + Newton polish.
//   The host classifies eps once per surface (surface_table.py) -> aniso_class.
//   E = null vector of W(xi) (largest cross product of two rows; rank-1 fallback
//   for the touching sheets), scaled like LAPACK's unit-norm 6-vector [xi E; E]:
//   |E|^2 (1+|xi|^2) = 1 -- S.n is computed with that scaling and decides the
//   order of the two transmitted solutions (material.py:144-151).
// Solutions with complex xi (evanescent) come out of THIS code as NaN and are not traced further; the complex k
// the reference carries there is filled in by a post-pass (prt_kernels.h: k_evanescent_fill, prt_trace_args_t.k_out_im).
// Complex (absorbing) eps: prt_aniso_cplx.h.
#pragma once
#include "prt_device.h"

struct aniso_solution {
    vec3 k;       // wave vector, global frame
    vec3 d;       // unit Poynting direction, global frame
    vec3 er, ei;  // E field, global frame (imaginary part 0 for real eps / real xi)
    vec3 kv;      // the wave vector in the frame of the medium, before a mirror's sign: what closed_form_ray takes
    bool is_e;    // uniaxial epsilon: the extraordinary wave
};

struct cplx {
    double re, im;
};
PRT_DEV cplx cmul(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PRT_DEV cplx cadd(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
PRT_DEV cplx csub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
PRT_DEV cplx cdiv(cplx a, cplx b) {
    const double den = fast_rcp(b.re * b.re + b.im * b.im);
    return cplx{(a.re * b.re + a.im * b.im) * den, (a.im * b.re - a.re * b.im) * den};
}
PRT_DEV double cabs2(cplx a) { return a.re * a.re + a.im * a.im; }

PRT_DEV vec3 cross(const vec3 &a, const vec3 &b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// quartic coefficients, calcXiPolynomialNorm (material.py:501-566), real eps
PRT_DEV void xi_polynomial(const double *__restrict__ eps, const vec3 &n, const vec3 &kpa,
                           double p[5]) {
    const vec3 en = mat_vec(eps, n), ek = mat_vec(eps, kpa);
    const vec3 etn = matT_vec(eps, n), etk = matT_vec(eps, kpa);
    const double a1 = eps[0] + eps[4] + eps[8];
    double a2 = 0.0, a3 = 0.0;
    // a2 = tr(eps^2), a3 = tr(eps^3)
    double e2[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            e2[i * 3 + j] = eps[i * 3] * eps[j] + eps[i * 3 + 1] * eps[3 + j] + eps[i * 3 + 2] * eps[6 + j];
    a2 = e2[0] + e2[4] + e2[8];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) a3 += e2[i * 3 + j] * eps[j * 3 + i];
    const double a4 = dot(kpa, kpa);
    const double a5 = dot(kpa, ek);
    const double a6 = dot(etk, ek);   // kpa^T eps eps kpa
    const double a7 = dot(n, en);
    const double a8 = dot(n, ek);     // eps_ij kpa_j n_i
    const double a9 = dot(kpa, en);   // eps_ij kpa_i n_j
    const double a11 = dot(etn, ek);  // n^T eps eps kpa
    const double a12 = dot(etk, en);  // kpa^T eps eps n
    const double a13 = dot(etn, en);  // n^T eps eps n
    p[4] = a7;
    p[3] = a8 + a9;
    p[2] = (a5 + a4 * a7) + (a13 - a1 * a7);
    p[1] = a4 * p[3] + (a11 + a12 - a1 * p[3]);
    p[0] = a4 * a5 + (-a1 * a5 + a6) + (1.0 / 6.0) * (a1 * a1 * a1 - 3.0 * a1 * a2 + 2.0 * a3);
}

// Aberth-Ehrlich on p4 z^4 + ... + p0 (complex roots), then two Newton steps each.
PRT_DEV void quartic_roots(const double p[5], cplx z[4]) {
    const double ip4 = fast_rcp(p[4]);
    const double a3 = p[3] * ip4, a2 = p[2] * ip4, a1 = p[1] * ip4, a0 = p[0] * ip4;
    // Cauchy bound for the start radius
    const double rad = 1.0 + fmax(fmax(fabs(a3), fabs(a2)), fmax(fabs(a1), fabs(a0)));
    const double r0 = 0.5 * rad;
    z[0] = cplx{r0 * 0.9238795325112867, r0 * 0.3826834323650898};
    z[1] = cplx{-r0 * 0.3826834323650898, r0 * 0.9238795325112867};
    z[2] = cplx{-r0 * 0.9238795325112867, -r0 * 0.3826834323650898};
    z[3] = cplx{r0 * 0.3826834323650898, -r0 * 0.9238795325112867};
    bool done = false;
    for (int it = 0; it < 60; ++it) {
        double worst = 0.0;
        cplx w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const cplx zi = z[i];
            // f = z^4 + a3 z^3 + a2 z^2 + a1 z + a0 ; f' by Horner
            cplx f = cplx{1.0, 0.0}, fp = cplx{0.0, 0.0};
            fp = cadd(cmul(fp, zi), f);
            f = cadd(cmul(f, zi), cplx{a3, 0.0});
            fp = cadd(cmul(fp, zi), f);
            f = cadd(cmul(f, zi), cplx{a2, 0.0});
            fp = cadd(cmul(fp, zi), f);
            f = cadd(cmul(f, zi), cplx{a1, 0.0});
            fp = cadd(cmul(fp, zi), f);
            f = cadd(cmul(f, zi), cplx{a0, 0.0});
            cplx newton = cdiv(f, fp);
            cplx sum = cplx{0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j != i) sum = cadd(sum, cdiv(cplx{1.0, 0.0}, csub(zi, z[j])));
            const cplx den = csub(cplx{1.0, 0.0}, cmul(newton, sum));
            w[i] = cdiv(newton, den);
            if (!(cabs2(fp) > 0.0) || !isfinite(w[i].re) || !isfinite(w[i].im)) w[i] = cplx{0.0, 0.0};
            worst = fmax(worst, cabs2(w[i]) * fast_rcp(fmax(cabs2(zi), 1e-300)));
        }
        if (!done) {
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = csub(z[i], w[i]);
            done = worst < 1e-22;  // 1e-11 relative: the real-polynomial Newton polish below finishes the job
        }
        if (__all(done)) break;
    }
}

// Real-arithmetic alternative for the common case of four real roots (lossless crystal, no
// evanescent branch): Bairstow's method splits the monic quartic into the forward pair
// (x^2 - r x - s, both roots near +xi0) and the backward pair (the quotient).  The two factors
// are well separated (roots near +xi0 vs -xi0) even when the two modes inside a pair nearly
// coincide, so the Newton system stays well conditioned for weak birefringence.  ~20 flops per
// iteration instead of ~250 complex flops for an Aberth sweep.  Returns false (caller falls
// back to Aberth) when it does not converge.
PRT_DEV bool quartic_roots_bairstow(const double p[5], double xi0, double x[4]) {
    const double ip4 = fast_rcp(p[4]);
    const double a3 = p[3] * ip4, a2 = p[2] * ip4, a1 = p[1] * ip4, a0 = p[0] * ip4;
    double r = 2.0 * xi0, s = -xi0 * xi0;  // (x - xi0)^2
    double b2 = 0.0, b3 = 0.0;
    bool conv = false;
    for (int it = 0; it < 16; ++it) {
        // b_i = a_i + r b_{i+1} + s b_{i+2}  (b4 = 1); remainder b1 (x - r) + b0
        b3 = a3 + r;
        b2 = a2 + r * b3 + s;
        const double b1 = a1 + r * b2 + s * b3;
        const double b0 = a0 + r * b1 + s * b2;
        const double c3 = b3 + r;
        const double c2 = b2 + r * c3 + s;
        const double c1 = b1 + r * c2 + s * c3;
        // [c2 c3; c1 c2] [dr ds]^T = -[b1 b0]^T
        const double det = c2 * c2 - c1 * c3;
        const double idet = fast_rcp(det);
        const double dr = (-b1 * c2 + b0 * c3) * idet;
        const double ds = (-b0 * c2 + b1 * c1) * idet;
        if (!conv) {
            r += dr;
            s += ds;
            conv = (fabs(dr) + fabs(ds) <= 1e-13 * (fabs(r) + fabs(s) + 1e-300)) || !isfinite(dr + ds);
        }
        if (__all(conv)) break;
    }
    if (!isfinite(r) || !isfinite(s) || !conv) return false;
    // quotient with the converged (r, s)
    b3 = a3 + r;
    b2 = a2 + r * b3 + s;
    // factor 1: x^2 - r x - s ; factor 2: x^2 + b3 x + b2
    const double d1 = 0.25 * r * r + s;
    const double d2 = 0.25 * b3 * b3 - b2;
    const double q1 = fast_sqrt(d1), q2 = fast_sqrt(d2);  // NaN if a pair is complex
    x[0] = -0.5 * b3 - q2;
    x[1] = -0.5 * b3 + q2;
    x[2] = 0.5 * r - q1;
    x[3] = 0.5 * r + q1;
    return true;
}

// null vector(s) of the real 3x3 matrix W.  variant selects which of the two
// basis vectors is returned when W has rank <= 1 (touching dispersion sheets).
// (rows passed as three register vectors: with an array the compiler turns the row selection
// of the fallback into indexed scratch loads)
PRT_DEV vec3 null_vector(const vec3 &r0, const vec3 &r1, const vec3 &r2, int variant) {
    const vec3 c01 = cross(r0, r1), c02 = cross(r0, r2), c12 = cross(r1, r2);
    const double n01 = dot(c01, c01), n02 = dot(c02, c02), n12 = dot(c12, c12);
    // component-wise selects (a vec3 "best = cXY" under an if becomes an indexed scratch load)
    const bool p02 = (n02 > n01) && (n02 >= n12), p12 = (n12 > n01) && (n12 > n02);
    const double nb = p12 ? n12 : (p02 ? n02 : n01);
    const vec3 best = v3(p12 ? c12.x : (p02 ? c02.x : c01.x), p12 ? c12.y : (p02 ? c02.y : c01.y),
                         p12 ? c12.z : (p02 ? c02.z : c01.z));
    const double fro = dot(r0, r0) + dot(r1, r1) + dot(r2, r2);
    if (nb > 1e-20 * fro * fro) {
        const double inv = fast_rsqrt(nb);
        return v3(best.x * inv, best.y * inv, best.z * inv);
    }
    // rank <= 1: null space is the plane perpendicular to the dominant row
    const double q0 = dot(r0, r0), q1 = dot(r1, r1), q2 = dot(r2, r2);
    const bool use1 = (q1 > q0) && (q1 >= q2), use2 = (q2 > q0) && (q2 > q1);
    const vec3 r = v3(use2 ? r2.x : (use1 ? r1.x : r0.x), use2 ? r2.y : (use1 ? r1.y : r0.y),
                      use2 ? r2.z : (use1 ? r1.z : r0.z));
    const double ax = fabs(r.x), ay = fabs(r.y), az = fabs(r.z);
    vec3 a = (ax <= ay && ax <= az) ? v3(1, 0, 0) : ((ay <= az) ? v3(0, 1, 0) : v3(0, 0, 1));
    vec3 v1 = cross(r, a);
    double inv = fast_rsqrt(dot(v1, v1));
    v1 = v3(v1.x * inv, v1.y * inv, v1.z * inv);
    if (variant == 0) return v1;
    vec3 v2 = cross(r, v1);
    inv = fast_rsqrt(dot(v2, v2));
    return v3(v2.x * inv, v2.y * inv, v2.z * inv);
}

// Eigenvector E (scaled like LAPACK's unit-norm [xi E; E]) and S.n of ONE solution xi of the
// dispersion relation.  Closed forms where the class of eps has them (a third of the instructions of
// the generic null vector, which stays for biaxial crystals and for the neighbourhood of the optic
// axis, where the two sheets touch and these forms lose their digits):
//   eps = e I:                       any two orthonormal vectors perpendicular to k
//   eps = eo I + (ee - eo) c c^T:    ordinary  E = k x c   (perpendicular to k and to the axis),
//                                    extraordinary  E = (k.c) k - eo c
//     (W E = [k^T eps k - eo ee] c = 0 on the extraordinary sheet; D = eps E is perpendicular to k)
// In three pieces, so that the caller can run the straight-line parts of two solutions side by side (two
// independent dependency chains in one basic block) and keep the rare fallback in one branch:
template <class REC>
PRT_DEV vec3 closed_form_e(const REC *__restrict__ sf, int cls, const vec3 &kv, double k2, int variant,
                           bool &closed) {
    if (cls == PRT_ANISO_ISOTROPIC) {
        const double ax = fabs(kv.x), ay = fabs(kv.y), az = fabs(kv.z);
        const vec3 a = (ax <= ay && ax <= az) ? v3(1, 0, 0) : ((ay <= az) ? v3(0, 1, 0) : v3(0, 0, 1));
        vec3 v1 = cross(kv, a);
        double inv = fast_rsqrt(dot(v1, v1));
        v1 = v3(v1.x * inv, v1.y * inv, v1.z * inv);
        closed = true;
        if (variant == 0) return v1;
        const vec3 v2 = cross(kv, v1);
        inv = fast_rsqrt(dot(v2, v2));
        return v3(v2.x * inv, v2.y * inv, v2.z * inv);
    }
    if (cls == PRT_ANISO_UNIAXIAL) {
        const vec3 caxis = v3(sf->aniso_axis[0], sf->aniso_axis[1], sf->aniso_axis[2]);
        const vec3 kxc = cross(kv, caxis);
        const double q = dot(kxc, kxc);
        closed = q > 1e-8 * k2;  // sin^2 of the angle to the optic axis
        const double kc = dot(kv, caxis);
        const vec3 ex = v3(kc * kv.x - sf->aniso_eo * caxis.x, kc * kv.y - sf->aniso_eo * caxis.y,
                           kc * kv.z - sf->aniso_eo * caxis.z);
        const vec3 raw = (variant == 0) ? kxc : ex;
        const double inv = fast_rsqrt(dot(raw, raw));
        return v3(raw.x * inv, raw.y * inv, raw.z * inv);
    }
    closed = false;
    return v3(0.0, 0.0, 0.0);
}

template <class REC>
PRT_DEV vec3 generic_e(const REC *__restrict__ sf, const vec3 &kv, double k2, int variant) {
    const double *__restrict__ eps = cold(sf)->eps_re;
    const vec3 w0 = v3(eps[0] - k2 + kv.x * kv.x, eps[1] + kv.x * kv.y, eps[2] + kv.x * kv.z);
    const vec3 w1 = v3(eps[3] + kv.y * kv.x, eps[4] - k2 + kv.y * kv.y, eps[5] + kv.y * kv.z);
    const vec3 w2 = v3(eps[6] + kv.z * kv.x, eps[7] + kv.z * kv.y, eps[8] - k2 + kv.z * kv.z);
    return null_vector(w0, w1, w2, variant);
}

PRT_DEV void scaled_e_and_flux(const vec3 &E0, const vec3 &kv, const vec3 &n, double x, vec3 &E_out, double &sn_out) {
    const double sc = fast_rsqrt(1.0 + x * x);  // LAPACK unit-norm [xi E; E]
    const vec3 E = v3(E0.x * sc, E0.y * sc, E0.z * sc);
    const double e2 = dot(E, E), ke = dot(kv, E);
    const vec3 S = v3(e2 * kv.x - ke * E.x, e2 * kv.y - ke * E.y, e2 * kv.z - ke * E.z);
    E_out = E;
    sn_out = dot(S, n);
}

template <class REC>
PRT_DEV void eigen_solution(const REC *__restrict__ sf, int cls, const vec3 &kpa, const vec3 &n,
                            double x, int variant, vec3 &E_out, double &sn_out) {
    const vec3 kv = v3(kpa.x + x * n.x, kpa.y + x * n.y, kpa.z + x * n.z);
    const double k2 = dot(kv, kv);
    bool closed;
    vec3 E = closed_form_e(sf, cls, kv, k2, variant, closed);
    if (!closed) E = generic_e(sf, kv, k2, variant);
    scaled_e_and_flux(E, kv, n, x, E_out, sn_out);
}

// The solutions (xi, E, S.n) of the dispersion relation for one ray, then the reference's ordering.
// p: hit point in the shape frame; k: incoming wave vector (global).
// out[0], out[1]: refract -> sorted solutions 2, 3; mirror -> -(0), -(1).
// GENERAL = false: the host guarantees that no crystal of the table needs the quartic solver (all
// epsilon tensors isotropic or uniaxial) and that code is compiled out.
// The complete path of the general (biaxial) class: eigenvectors and S.n of all four solutions, the reference's sort
// (material.py:147), the two that leave.  It runs only where the pair test of the fast path fails (evanescent
// modes, a failed Bairstow split, an exotic slowness surface).  Inlined: as a real call (tried in round 4) the callee is
// compiled without the kernel's register budget -- 214 VGPRs, 2 waves per SIMD for the whole kernel.
template <class REC>
__device__ __forceinline__ void four_solution_path(const REC *__restrict__ sf, int cls, const vec3 &kpa,
                                                           const vec3 &n, const double pc[5], double xr[4], bool mirror,
                                                           double x_out[2], vec3 e_out[2]) {
        double xi[4];
        // ascending order (neighbours in a near-degenerate pair get different null-vector variants)
#pragma unroll
        for (int i = 1; i < 4; ++i)
#pragma unroll
            for (int j = i; j > 0; --j) {
                const bool sw = (xr[j] < xr[j - 1]) || (isnan(xr[j - 1]) && !isnan(xr[j]));
                const double lo = sw ? xr[j] : xr[j - 1], hi = sw ? xr[j - 1] : xr[j];
                xr[j - 1] = lo;
                xr[j] = hi;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double x = xr[i];
            // Newton polish on the real polynomial
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const double f = (((pc[4] * x + pc[3]) * x + pc[2]) * x + pc[1]) * x + pc[0];
                const double fp = ((4.0 * pc[4] * x + 3.0 * pc[3]) * x + 2.0 * pc[2]) * x + pc[1];
                const double dx = f * fast_rcp(fp);
                if (isfinite(dx) && fabs(dx) < 1e-6 * fmax(1.0, fabs(x))) x -= dx;
            }
            xi[i] = x;
        }

        // eigenvectors and S.n of the four solutions.  Only E (scaled) and S.n are kept per solution --
        // k = kpa + xi n and S are a handful of operations to rebuild for the two solutions that leave,
        // and holding all four (k, E, S) triples cost 24 more live doubles at the kernel's register peak
        vec3 ee_[4];
        double sn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) eigen_solution(sf, cls, kpa, n, xi[i], i & 1, ee_[i], sn[i]);
        // argsort ascending by S.n (material.py:147); NaNs last like numpy.  Compare-exchange
        // network on (key, id) pairs held in registers (indexing sn[] by a sorted index would put
        // the arrays in scratch memory).  Evanescent modes: key 0, see above.
        int id0 = 0, id1 = 1, id2 = 2, id3 = 3;
        double s0 = isnan(sn[0]) ? 0.0 : sn[0], s1 = isnan(sn[1]) ? 0.0 : sn[1];
        double s2 = isnan(sn[2]) ? 0.0 : sn[2], s3 = isnan(sn[3]) ? 0.0 : sn[3];
#define PRT_CSWAP(ka, kb, ia, ib)                                   \
    {                                                               \
        const bool sw = (kb < ka) || (isnan(ka) && !isnan(kb));     \
        const double tk = sw ? kb : ka;                             \
        kb = sw ? ka : kb;                                          \
        ka = tk;                                                    \
        const int ti = sw ? ib : ia;                                \
        ib = sw ? ia : ib;                                          \
        ia = ti;                                                    \
    }
        PRT_CSWAP(s0, s1, id0, id1)
        PRT_CSWAP(s2, s3, id2, id3)
        PRT_CSWAP(s0, s2, id0, id2)
        PRT_CSWAP(s1, s3, id1, id3)
        PRT_CSWAP(s1, s2, id1, id2)
#undef PRT_CSWAP
        const int idx[4] = {id0, id1, id2, id3};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int src = mirror ? idx[b] : idx[2 + b];
            // pick without dynamic register indexing
            vec3 E = ee_[0];
            double x = xi[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (src == q) { E = ee_[q]; x = xi[q]; }
            e_out[b] = E;
            x_out[b] = x;
        }
}

// n: unit surface normal in the frame of the medium (the fused march has it from the intersection it just did:
// normal_from_grad; the per-surface entry point evaluates the shape at the caller's point)
// Energy flux of a solution in a crystal with a SYMMETRIC real epsilon, without its eigenvector.  W = eps - (k.k) I +
// k k^T is symmetric and singular at a solution; its adjugate is then  adj W = t e e^T  (e the unit null vector,
// t = tr adj W the product of the two non-zero eigenvalues), so the part of k perpendicular to E is
//     S0 = k - (k.e) e = k - (adj W) k / tr(adj W)
// -- six cofactors, one matrix-vector product, no search for the best-conditioned pair of rows, no normalisation of
// E.  Returned: T = tr(adj W) k - (adj W) k = tr(adj W) S0 and tr(adj W); the caller divides or compares signs.
// |tr adj W| small against |W|^2 = the two sheets touch (the optic axes of a biaxial crystal): fall back to E.
PRT_DEV void flux_symmetric(const double *__restrict__ eps, const vec3 &kv, vec3 &T, double &tr, double &fro2) {
    const double k2 = dot(kv, kv);
    const double w00 = eps[0] - k2 + kv.x * kv.x, w11 = eps[4] - k2 + kv.y * kv.y, w22 = eps[8] - k2 + kv.z * kv.z;
    const double w01 = eps[1] + kv.x * kv.y, w02 = eps[2] + kv.x * kv.z, w12 = eps[5] + kv.y * kv.z;
    const double a00 = w11 * w22 - w12 * w12, a11 = w00 * w22 - w02 * w02, a22 = w00 * w11 - w01 * w01;
    const double a01 = w02 * w12 - w01 * w22, a02 = w01 * w12 - w02 * w11, a12 = w01 * w02 - w00 * w12;
    tr = a00 + a11 + a22;
    fro2 = w00 * w00 + w11 * w11 + w22 * w22 + 2.0 * (w01 * w01 + w02 * w02 + w12 * w12);
    T = v3(tr * kv.x - (a00 * kv.x + a01 * kv.y + a02 * kv.z), tr * kv.y - (a01 * kv.x + a11 * kv.y + a12 * kv.z),
           tr * kv.z - (a02 * kv.x + a12 * kv.y + a22 * kv.z));
}

// Wave vector and ray direction (global frame) of ONE solution in a medium whose epsilon is isotropic or uniaxial,
// from its wave vector kv in the frame of the medium (before a mirror's sign) and the flag "extraordinary wave":
//   u = kv (ordinary wave, eps = e I)   or   u = eps kv = eo kv + (ee - eo)(kv.c) c (extraordinary),   d = u / |u|
// (derivation: interact_anisotropic_n).  Exactly the operations interact_anisotropic_n performs for its two
// solutions, so a child of the crystal march that was parked as (kv, is_e) -- 7 values instead of the 10 of
// (k, d, alive) -- resumes with bit-identical k and d.
template <class REC>
PRT_DEV vec3 closed_form_u(const REC *__restrict__ sf, int cls, const vec3 &kv, bool is_e) {
    if (cls != PRT_ANISO_UNIAXIAL) return kv;
    const vec3 c = v3(sf->aniso_axis[0], sf->aniso_axis[1], sf->aniso_axis[2]);
    const double w = (sf->aniso_ee - sf->aniso_eo) * dot(kv, c);
    const double f = is_e ? sf->aniso_eo : 1.0, g = is_e ? w : 0.0;
    return v3(__builtin_fma(g, c.x, f * kv.x), __builtin_fma(g, c.y, f * kv.y), __builtin_fma(g, c.z, f * kv.z));
}
template <class REC>
PRT_DEV void closed_form_finish(const REC *__restrict__ sf, vec3 kv, vec3 d, vec3 &k_glob, vec3 &d_glob) {
    if (sf->interaction == PRT_MIRROR) {  // material_anisotropic.py:136-137: k negated (S is odd in k)
        kv = v3(-kv.x, -kv.y, -kv.z);
        d = v3(-d.x, -d.y, -d.z);
    }
    if (!(sf->frame_flags & PRT_FRAME_MAT_IDENTITY)) {
        kv = mat_vec(cold(sf)->B_mat, kv);
        d = mat_vec(cold(sf)->B_mat, d);
    }
    k_glob = kv;
    d_glob = d;
}
template <class REC>
PRT_DEV void closed_form_ray(const REC *__restrict__ sf, const vec3 &kv, bool is_e, vec3 &k_glob, vec3 &d_glob) {
    const vec3 u = closed_form_u(sf, sf->aniso_class, kv, is_e);
    const double inv = fast_rsqrt(dot(u, u));
    closed_form_finish(sf, kv, v3(u.x * inv, u.y * inv, u.z * inv), k_glob, d_glob);
}

// want_e (the same for every lane): the caller stores the E fields -- they are computed only then (uniaxial and
// isotropic epsilon: the ray directions come from closed forms that need no eigenvector)
template <bool GENERAL = true, class REC>
PRT_DEV void interact_anisotropic_n(const REC *__restrict__ sf, const vec3 &n,
                                    const vec3 &k_glob, aniso_solution out[2], bool want_e = true);

template <bool GENERAL = true, int SHAPES = PRT_SHAPES_ALL, class REC>
PRT_DEV void interact_anisotropic(const REC *__restrict__ sf, const vec3 &p,
                                  const vec3 &k_glob, aniso_solution out[2], bool want_e = true) {
    interact_anisotropic_n<GENERAL>(sf, normal_in_material_frame<SHAPES>(sf, p), k_glob, out, want_e);
}

// The two solutions of the pair (xa, xb) that should leave, from the adjugate flux (symmetric epsilon): S.n order,
// ray directions.  Returns whether both carry energy the right way and neither sits where the sheets touch (per lane).
// x_out / d_out are assigned in any case (the caller decides what to do with a lane that is not ok).
PRT_DEV bool flux_pair(const double *__restrict__ eps, const vec3 &kpa, const vec3 &n, double xa, double xb, bool mirror,
                       double x_out[2], vec3 d_out[2]) {
    const vec3 ka = v3(kpa.x + xa * n.x, kpa.y + xa * n.y, kpa.z + xa * n.z);
    const vec3 kb = v3(kpa.x + xb * n.x, kpa.y + xb * n.y, kpa.z + xb * n.z);
    vec3 Ta, Tb;
    double ta, tb, fa2, fb2;
    flux_symmetric(eps, ka, Ta, ta, fa2);
    flux_symmetric(eps, kb, Tb, tb, fb2);
    // S0.n = (T.n)/t;  S.n = S0.n / (1 + xi^2)
    const double sa = dot(Ta, n) * ta, sb = dot(Tb, n) * tb;  // same sign as S.n (multiplied by t^2 > 0)
    const bool ok = (fabs(ta) > 1e-9 * fa2) && (fabs(tb) > 1e-9 * fb2) &&
                    (mirror ? (sa < 0.0 && sb < 0.0) : (sa > 0.0 && sb > 0.0));
    // ascending S.n inside the pair: sb / (tb^2 (1 + xb^2)) < sa / (ta^2 (1 + xa^2))
    const bool sw = sb * (ta * ta) * (1.0 + xa * xa) < sa * (tb * tb) * (1.0 + xb * xb);
    const double ia = copysign(fast_rsqrt(dot(Ta, Ta)), ta), ib = copysign(fast_rsqrt(dot(Tb, Tb)), tb);
    const vec3 da = v3(Ta.x * ia, Ta.y * ia, Ta.z * ia), db = v3(Tb.x * ib, Tb.y * ib, Tb.z * ib);
    x_out[0] = sw ? xb : xa;
    x_out[1] = sw ? xa : xb;
    d_out[0] = v3(sw ? db.x : da.x, sw ? db.y : da.y, sw ? db.z : da.z);
    d_out[1] = v3(sw ? da.x : db.x, sw ? da.y : db.y, sw ? da.z : db.z);
    return ok;
}

// the two real roots of the pair that should leave, polished on the real polynomial (two Newton steps each)
PRT_DEV void polish_pair(const double pc[5], double &xa, double &xb) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double fa = (((pc[4] * xa + pc[3]) * xa + pc[2]) * xa + pc[1]) * xa + pc[0];
        const double fpa = ((4.0 * pc[4] * xa + 3.0 * pc[3]) * xa + 2.0 * pc[2]) * xa + pc[1];
        const double da = fa * fast_rcp(fpa);
        if (isfinite(da) && fabs(da) < 1e-6 * fmax(1.0, fabs(xa))) xa -= da;
        const double fb = (((pc[4] * xb + pc[3]) * xb + pc[2]) * xb + pc[1]) * xb + pc[0];
        const double fpb = ((4.0 * pc[4] * xb + 3.0 * pc[3]) * xb + 2.0 * pc[2]) * xb + pc[1];
        const double db = fb * fast_rcp(fpb);
        if (isfinite(db) && fabs(db) < 1e-6 * fmax(1.0, fabs(xb))) xb -= db;
    }
}

template <bool GENERAL, class REC>
PRT_DEV void interact_anisotropic_n(const REC *__restrict__ sf, const vec3 &n,
                                    const vec3 &k_glob, aniso_solution out[2], bool want_e) {
    const bool mat_id = sf->frame_flags & PRT_FRAME_MAT_IDENTITY;
    const vec3 k1 = mat_id ? k_glob : matT_vec(cold(sf)->B_mat, k_glob);
    const double kn = dot(k1, n);
    const vec3 kpa = v3(k1.x - kn * n.x, k1.y - kn * n.y, k1.z - kn * n.z);
    const double *__restrict__ eps = cold(sf)->eps_re;
    const double kap2 = dot(kpa, kpa);
    const int cls = sf->aniso_class;
    const bool mirror = sf->interaction == PRT_MIRROR;

    // the two solutions that leave, in the reference's order
    vec3 e_out[2];
    double x_out[2];
    vec3 d_out[2];        // their unit ray directions (material frame, before a mirror's sign), if have_d
    bool have_d = false;  // (compile-time after inlining: the closed-form classes set it)
    bool is_e_out[2] = {false, false};

    if (cls == PRT_ANISO_ISOTROPIC || cls == PRT_ANISO_UNIAXIAL) {
        // The four roots come as an ordinary pair -ro < +ro and an extraordinary pair x1 < x2 of a
        // quadratic A x^2 + 2 Bh x + C with A > 0.  S.n has the sign of the derivative of the dispersion
        // polynomial at the root: negative at -ro and x1, positive at +ro and x2.  The reference sorts
        // all four by S.n (material_anisotropic.py:147) and refracts into the sorted solutions 2, 3,
        // reflects into -(0), -(1): the forward pair {+ro, x2} resp. the backward pair {-ro, x1},
        // ordered among themselves by S.n.  Only that pair is computed here -- half the eigenvector
        // work of the generic path below, and no four-element sort.
        // An evanescent mode (complex xi, NaN here) carries no energy flux through the interface: the
        // reference computes S.n ~ 0 for it, which places it BETWEEN the backward (S.n < 0) and the
        // forward (S.n > 0) propagating modes, i.e. before a propagating forward partner and after a
        // propagating backward one.  Its key is 0 here for the same order ([NaN, real] for one
        // transmitted mode totally reflected).
        double x_o, x_e;
        if (cls == PRT_ANISO_ISOTROPIC) {
            const double r = fast_sqrt(sf->aniso_eo - kap2);  // NaN if evanescent
            x_o = x_e = mirror ? -r : r;
        } else {
            const double eo = sf->aniso_eo, ee = sf->aniso_ee;
            const vec3 c = v3(sf->aniso_axis[0], sf->aniso_axis[1], sf->aniso_axis[2]);
            const double ro = fast_sqrt(eo - kap2);
            const double nc = dot(n, c), kc = dot(kpa, c);
            const double A = eo + (ee - eo) * nc * nc;
            const double Bh = (ee - eo) * kc * nc;  // B/2
            const double C = eo * kap2 + (ee - eo) * kc * kc - eo * ee;
            const double disc = fast_sqrt(Bh * Bh - A * C);  // NaN if evanescent
            // stable quadratic roots
            const double q = -(Bh + copysign(disc, Bh));
            // (fast_rcp: ~1 ulp; the IEEE division sequences of these four quotients were a sixth of the
            // solver's instructions)
            const double iA = fast_rcp(A);
            double x1 = q * iA, x2 = (q != 0.0) ? C * fast_rcp(q) : -x1;
            if (Bh == 0.0) { x1 = -disc * iA; x2 = disc * iA; }
            x_o = mirror ? -ro : ro;
            x_e = mirror ? fmin(x1, x2) : fmax(x1, x2);
            if (!isfinite(disc)) x_e = __builtin_nan("");
        }
        // THE TWO RAYS WITHOUT THEIR EIGENVECTORS.  With E scaled like LAPACK's unit 6-vector, |E|^2 = 1/(1 + xi^2), the
        // reference's S = |E|^2 k - (k.E) E (material.py:214-223) is  S = S0 / (1 + xi^2),  S0 = k - (k.e) e  for the
        // unit vector e = E/|E|: the part of k perpendicular to E.  S0 is known without E:
        //   ordinary wave  (E = k x c, perpendicular to k):           S0 = k
        //   extraordinary wave (D = eps E perpendicular to k):       S0 is parallel to eps k (the normal of the sheet
        //       k^T eps k = eo ee of the slowness surface) and |S0|^2 = S0.k, so  S0 = (u.k) u,  u = eps k / |eps k|
        //   eps = e I: both waves like the ordinary one.
        // Ray directions d = S0/|S0| and the S.n order of the pair (material.py:147) follow from that -- 160 VALU
        // instructions less per interface than with the two eigenvectors (k x c, (k.c) k - eo c, their norms, LAPACK's
        // scaling, the flux), and no loss of digits next to the optic axis, where the extraordinary eigenvector form
        // cancels (the eigenvectors keep their fall-back there).  E itself is computed only for a caller that stores it.
        const vec3 kv_o = v3(kpa.x + x_o * n.x, kpa.y + x_o * n.y, kpa.z + x_o * n.z);
        const vec3 kv_e = v3(kpa.x + x_e * n.x, kpa.y + x_e * n.y, kpa.z + x_e * n.z);
        const vec3 u_e = closed_form_u(sf, cls, kv_e, true);
        const double inv_o = fast_rsqrt(dot(kv_o, kv_o)), inv_e = fast_rsqrt(dot(u_e, u_e));
        const vec3 d_o = v3(kv_o.x * inv_o, kv_o.y * inv_o, kv_o.z * inv_o);
        const vec3 d_e = v3(u_e.x * inv_e, u_e.y * inv_e, u_e.z * inv_e);
        // S.n of the two, compared without dividing:  S0o.n / (1 + xo^2)  vs  S0e.n / (1 + xe^2); an evanescent mode
        // (NaN) has the key 0, between the backward and the forward propagating modes (see above)
        const double so_n = dot(kv_o, n), se_n = dot(d_e, kv_e) * dot(d_e, n);
        const bool nan_o = isnan(so_n), nan_e = isnan(se_n);
        const double a_o = nan_o ? 0.0 : so_n, a_e = nan_e ? 0.0 : se_n;
        const double w_o = nan_o ? 1.0 : 1.0 + x_o * x_o, w_e = nan_e ? 1.0 : 1.0 + x_e * x_e;
        // (eps = e I: the two waves are the same wave, kept in the order of their two E vectors)
        const bool sw = (cls != PRT_ANISO_ISOTROPIC) && (a_e * w_o < a_o * w_e);  // stable: the ordinary solution first on a tie
        x_out[0] = sw ? x_e : x_o;
        x_out[1] = sw ? x_o : x_e;
        d_out[0] = v3(sw ? d_e.x : d_o.x, sw ? d_e.y : d_o.y, sw ? d_e.z : d_o.z);
        d_out[1] = v3(sw ? d_o.x : d_e.x, sw ? d_o.y : d_e.y, sw ? d_o.z : d_e.z);
        have_d = true;
        is_e_out[0] = sw && cls == PRT_ANISO_UNIAXIAL;
        is_e_out[1] = !sw && cls == PRT_ANISO_UNIAXIAL;
        e_out[0] = e_out[1] = v3(0.0, 0.0, 0.0);
        if (want_e) {
            // the eigenvectors (closed forms; the generic null vector next to the optic axis), scaled like LAPACK's
            const double k2_o = dot(kv_o, kv_o), k2_e = dot(kv_e, kv_e);
            bool closed_o, closed_e;
            vec3 E_o = closed_form_e(sf, cls, kv_o, k2_o, 0, closed_o);
            vec3 E_e = closed_form_e(sf, cls, kv_e, k2_e, 1, closed_e);
            if (!closed_o) E_o = generic_e(sf, kv_o, k2_o, 0);
            if (!closed_e) E_e = generic_e(sf, kv_e, k2_e, 1);
            const double sc_o = fast_rsqrt(1.0 + x_o * x_o), sc_e = fast_rsqrt(1.0 + x_e * x_e);
            E_o = v3(E_o.x * sc_o, E_o.y * sc_o, E_o.z * sc_o);
            E_e = v3(E_e.x * sc_e, E_e.y * sc_e, E_e.z * sc_e);
            e_out[0] = v3(sw ? E_e.x : E_o.x, sw ? E_e.y : E_o.y, sw ? E_e.z : E_o.z);
            e_out[1] = v3(sw ? E_o.x : E_e.x, sw ? E_o.y : E_e.y, sw ? E_o.z : E_e.z);
        }
    } else if (!GENERAL) {
        x_out[0] = x_out[1] = __builtin_nan("");
        e_out[0] = e_out[1] = v3(0.0, 0.0, 0.0);
        d_out[0] = d_out[1] = v3(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
        have_d = true;
    } else {
        double pc[5];
        xi_polynomial(eps, n, kpa, pc);
        // forward / backward pairs by Bairstow from the mean-index guess; the complex Aberth
        // sweep only where that fails (wave-uniform vote keeps the expensive path out of the
        // common case)
        double xr[4];
        const double xi0sq = (eps[0] + eps[4] + eps[8]) * (1.0 / 3.0) - kap2;
        bool have = false;
        if (xi0sq > 0.0) have = quartic_roots_bairstow(pc, fast_sqrt(xi0sq), xr);
        const bool all_real = have && isfinite(xr[0]) && isfinite(xr[1]) && isfinite(xr[2]) && isfinite(xr[3]);
        if (__builtin_expect(!__all(all_real), 0)) {
            cplx z[4];
            quartic_roots(pc, z);
            // sort by real part (insertion), keep real ones
#pragma unroll
            for (int i = 1; i < 4; ++i)
#pragma unroll
                for (int j = i; j > 0; --j)
                    if (z[j].re < z[j - 1].re) { cplx t = z[j]; z[j] = z[j - 1]; z[j - 1] = t; }
            if (!all_real) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool real_root = fabs(z[i].im) <= 1e-9 * fmax(1.0, fabs(z[i].re));
                    xr[i] = real_root ? z[i].re : __builtin_nan("");
                }
            }
        }
        // FAST PATH (wave-uniform vote): Bairstow delivered four real roots in every lane, as the backward pair
        // xr[0] <= xr[1] (the quotient) and the forward pair xr[2] <= xr[3] (the factor near +xi0).  A lossless crystal
        // has exactly two modes with S.n > 0 among four real ones (every sheet of the slowness surface is entered and
        // left once along the normal), so if BOTH roots of the pair that should leave carry energy the right way they
        // ARE the reference's sorted solutions 2, 3 (refraction) resp. 0, 1 (reflection) -- no need for the
        // eigenvectors of the other pair nor for the four-element sort.  Any lane that fails the check sends the
        // wave through the complete path (four_solution_path).  Biaxial doublet, same arrays, builds interleaved: image
        // mode 0.180 -> 0.141 ms, path mode 0.219 -> 0.18 ms; 400-stack crystal stress campaign: 1 578 473 ray-surfaces, no deviation.
        bool fast = __all(all_real);
        if (__builtin_expect(fast, 1)) {
            double xa = mirror ? xr[0] : xr[2], xb = mirror ? xr[1] : xr[3];
            polish_pair(pc, xa, xb);  // Newton polish on the real polynomial, like in four_solution_path
            // symmetric epsilon, no E wanted: flux and ray direction of the pair from the adjugate of W (flux_symmetric)
            const bool sym = eps[1] == eps[3] && eps[2] == eps[6] && eps[5] == eps[7];
            bool flux_ok = false;
            if (!want_e && sym) {
                double xo[2];
                vec3 dd[2];
                flux_ok = __all(flux_pair(eps, kpa, n, xa, xb, mirror, xo, dd));
                if (flux_ok) {
                    x_out[0] = xo[0];
                    x_out[1] = xo[1];
                    d_out[0] = dd[0];
                    d_out[1] = dd[1];
                    e_out[0] = e_out[1] = v3(0.0, 0.0, 0.0);
                    have_d = true;
                }
            }
            if (__builtin_expect(!flux_ok, 0)) {
                vec3 Ea, Eb;
                double sa, sb;
                eigen_solution(sf, cls, kpa, n, xa, 0, Ea, sa);
                eigen_solution(sf, cls, kpa, n, xb, 1, Eb, sb);
                const bool ok = mirror ? (sa < 0.0 && sb < 0.0) : (sa > 0.0 && sb > 0.0);
                fast = __all(ok);
                if (fast) {  // ascending S.n inside the pair (material.py:147)
                    const bool sw = sb < sa;
                    x_out[0] = sw ? xb : xa;
                    x_out[1] = sw ? xa : xb;
                    e_out[0] = v3(sw ? Eb.x : Ea.x, sw ? Eb.y : Ea.y, sw ? Eb.z : Ea.z);
                    e_out[1] = v3(sw ? Ea.x : Eb.x, sw ? Ea.y : Eb.y, sw ? Ea.z : Eb.z);
                }
            }
        }
        if (__builtin_expect(!fast, 0)) four_solution_path(sf, cls, kpa, n, pc, xr, mirror, x_out, e_out);
    }

#pragma unroll
    for (int b = 0; b < 2; ++b) {
        vec3 E = e_out[b];
        const double x = x_out[b];
        vec3 kv = v3(kpa.x + x * n.x, kpa.y + x * n.y, kpa.z + x * n.z);
        out[b].kv = kv;
        out[b].is_e = is_e_out[b];
        vec3 d;
        if (have_d) {
            d = d_out[b];
        } else {
            const double e2 = dot(E, E), ke = dot(kv, E);
            const vec3 S = v3(e2 * kv.x - ke * E.x, e2 * kv.y - ke * E.y, e2 * kv.z - ke * E.z);
            const double inv = fast_rsqrt(dot(S, S));
            d = v3(S.x * inv, S.y * inv, S.z * inv);
        }
        if (mirror) {  // material_anisotropic.py:136-137: k, E negated (S is even in E, odd in k)
            kv = v3(-kv.x, -kv.y, -kv.z);
            E = v3(-E.x, -E.y, -E.z);
            d = v3(-d.x, -d.y, -d.z);
        }
        if (!mat_id) {
            kv = mat_vec(cold(sf)->B_mat, kv);
            E = mat_vec(cold(sf)->B_mat, E);
            d = mat_vec(cold(sf)->B_mat, d);
        }
        out[b].k = kv;
        out[b].d = d;
        out[b].er = E;
        out[b].ei = v3(0, 0, 0);
    }
}
