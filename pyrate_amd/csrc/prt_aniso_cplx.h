// prt_aniso_cplx.h -- AnisotropicMaterial.refract / reflect for a COMPLEX (absorbing) epsilon tensor, per ray.
//
// Reference: material_anisotropic.py:52-56 (get_epsilon_tensor returns the constant 3x3 tensor, which may be
// complex), :70-155 (refract / reflect), material.py:122-153 (sortKnormEField), :353-454 (the 6x6 pencil solved by
// scipy.linalg.eig per ray), :214-223 (calcPoytingVectorNorm).  With an absorbing crystal every wave vector behind
// the first interface is complex, k = kpa + xi n with complex kpa (the in-plane part of a complex incoming k; all
// products bilinear, no conjugate: material_anisotropic.py:79) and complex xi -- the four roots of
//     det W(xi) = 0,   W = eps - (k.k) I + k k^T = xi^2 M + xi C + K          (material.py:385-392)
// whose coefficients are the ones of calcXiPolynomialNorm (material.py:501-566) evaluated in complex arithmetic.
// The reference takes (xi, E) from LAPACK; here: Aberth-Ehrlich on the complex quartic + Newton polish, E = null
// vector of W(xi) (largest bilinear cross product of two rows) scaled like LAPACK's unit-norm 6-vector [xi E; E],
// S.n = Re(|E|^2 (k.n) - (k.E)(E*.n)) decides the order (ascending; refract -> solutions 2, 3; mirror -> -(0), -(1)).
// The ray then travels along the unit Poynting vector Re(|E|^2 k - (E.k) E*) (ray.py:136-152), which is real.
//
// This is the path of tables that contain a complex epsilon: they run through the per-surface march
// (prt.hip: trace_general), one launch per interface -- no register budget to respect, no fast paths; parity with
// the reference's own bundles (tests/golden/aniso_absorbing_*.npz) is what it is for.
#pragma once
#include "prt_aniso.h"

struct cx {
    double re, im;
};
PRT_DEV cx operator+(cx a, cx b) { return cx{a.re + b.re, a.im + b.im}; }
PRT_DEV cx operator-(cx a, cx b) { return cx{a.re - b.re, a.im - b.im}; }
PRT_DEV cx operator-(cx a) { return cx{-a.re, -a.im}; }
PRT_DEV cx operator*(cx a, cx b) { return cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PRT_DEV cx operator*(double s, cx a) { return cx{s * a.re, s * a.im}; }
PRT_DEV cx cx_conj(cx a) { return cx{a.re, -a.im}; }
PRT_DEV double cx_abs2(cx a) { return a.re * a.re + a.im * a.im; }
PRT_DEV cx cx_div(cx a, cx b) {
    const double den = 1.0 / (b.re * b.re + b.im * b.im);
    return cx{(a.re * b.re + a.im * b.im) * den, (a.im * b.re - a.re * b.im) * den};
}
PRT_DEV cx cx_real(double x) { return cx{x, 0.0}; }

struct cvec3 {
    cx x, y, z;
};
PRT_DEV cvec3 cv3(const vec3 &re, const vec3 &im) { return cvec3{cx{re.x, im.x}, cx{re.y, im.y}, cx{re.z, im.z}}; }
PRT_DEV vec3 cv_re(const cvec3 &a) { return v3(a.x.re, a.y.re, a.z.re); }
PRT_DEV vec3 cv_im(const cvec3 &a) { return v3(a.x.im, a.y.im, a.z.im); }
// bilinear products (no conjugate), like NumPy's sum(a * b)
PRT_DEV cx cv_dot(const cvec3 &a, const cvec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PRT_DEV cx cv_dot(const cvec3 &a, const vec3 &b) { return b.x * a.x + b.y * a.y + b.z * a.z; }
PRT_DEV cvec3 cv_cross(const cvec3 &a, const cvec3 &b) {
    return cvec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PRT_DEV double cv_norm2(const cvec3 &a) { return cx_abs2(a.x) + cx_abs2(a.y) + cx_abs2(a.z); }
PRT_DEV cvec3 cv_scale(cx s, const cvec3 &a) { return cvec3{s * a.x, s * a.y, s * a.z}; }
PRT_DEV cvec3 cv_mat_vec(const cx *__restrict__ e, const cvec3 &v) {
    return cvec3{e[0] * v.x + e[1] * v.y + e[2] * v.z, e[3] * v.x + e[4] * v.y + e[5] * v.z,
                 e[6] * v.x + e[7] * v.y + e[8] * v.z};
}
PRT_DEV cvec3 cv_matT_vec(const cx *__restrict__ e, const cvec3 &v) {
    return cvec3{e[0] * v.x + e[3] * v.y + e[6] * v.z, e[1] * v.x + e[4] * v.y + e[7] * v.z,
                 e[2] * v.x + e[5] * v.y + e[8] * v.z};
}

// quartic coefficients of det W(xi), calcXiPolynomialNorm (material.py:501-566) with complex eps and kpa
PRT_DEV void xi_polynomial_cplx(const cx *__restrict__ eps, const vec3 &n, const cvec3 &kpa, cx p[5]) {
    const cvec3 nc = cv3(n, v3(0.0, 0.0, 0.0));
    const cvec3 en = cv_mat_vec(eps, nc), ek = cv_mat_vec(eps, kpa);
    const cvec3 etn = cv_matT_vec(eps, nc), etk = cv_matT_vec(eps, kpa);
    const cx a1 = eps[0] + eps[4] + eps[8];
    cx e2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            e2[i * 3 + j] = eps[i * 3] * eps[j] + eps[i * 3 + 1] * eps[3 + j] + eps[i * 3 + 2] * eps[6 + j];
    const cx a2 = e2[0] + e2[4] + e2[8];
    cx a3 = cx{0.0, 0.0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a3 = a3 + e2[i * 3 + j] * eps[j * 3 + i];
    const cx a4 = cv_dot(kpa, kpa);
    const cx a5 = cv_dot(kpa, ek);
    const cx a6 = cv_dot(etk, ek);
    const cx a7 = cv_dot(nc, en);
    const cx a8 = cv_dot(nc, ek);
    const cx a9 = cv_dot(kpa, en);
    const cx a11 = cv_dot(etn, ek);
    const cx a12 = cv_dot(etk, en);
    const cx a13 = cv_dot(etn, en);
    p[4] = a7;
    p[3] = a8 + a9;
    p[2] = (a5 + a4 * a7) + (a13 - a1 * a7);
    p[1] = a4 * p[3] + (a11 + a12 - a1 * p[3]);
    p[0] = a4 * a5 + (a6 - a1 * a5) + (1.0 / 6.0) * (a1 * a1 * a1 - 3.0 * (a1 * a2) + 2.0 * a3);
}

// value and derivative of the monic quartic z^4 + a[3] z^3 + a[2] z^2 + a[1] z + a[0]
PRT_DEV void quartic_eval_cplx(const cx a[4], cx z, cx &f, cx &fp) {
    f = cx{1.0, 0.0};
    fp = cx{0.0, 0.0};
    for (int q = 3; q >= 0; --q) {
        fp = fp * z + f;
        f = f * z + a[q];
    }
}

// Aberth-Ehrlich on p[4] z^4 + ... + p[0] with complex coefficients, then Newton steps on each root
PRT_DEV void quartic_roots_cplx(const cx p[5], cx z[4]) {
    cx a[4];
    for (int q = 0; q < 4; ++q) a[q] = cx_div(p[q], p[4]);
    double big = 0.0;
    for (int q = 0; q < 4; ++q) big = fmax(big, sqrt(cx_abs2(a[q])));
    const double r0 = 0.5 * (1.0 + big);
    z[0] = cx{r0 * 0.9238795325112867, r0 * 0.3826834323650898};
    z[1] = cx{-r0 * 0.3826834323650898, r0 * 0.9238795325112867};
    z[2] = cx{-r0 * 0.9238795325112867, -r0 * 0.3826834323650898};
    z[3] = cx{r0 * 0.3826834323650898, -r0 * 0.9238795325112867};
    for (int it = 0; it < 80; ++it) {
        double worst = 0.0;
        cx w[4];
        for (int i = 0; i < 4; ++i) {
            cx f, fp;
            quartic_eval_cplx(a, z[i], f, fp);
            const cx newton = cx_div(f, fp);
            cx sum = cx{0.0, 0.0};
            for (int j = 0; j < 4; ++j)
                if (j != i) sum = sum + cx_div(cx{1.0, 0.0}, z[i] - z[j]);
            w[i] = cx_div(newton, cx{1.0, 0.0} - newton * sum);
            if (!(cx_abs2(fp) > 0.0) || !isfinite(w[i].re) || !isfinite(w[i].im)) w[i] = cx{0.0, 0.0};
            worst = fmax(worst, cx_abs2(w[i]) / fmax(cx_abs2(z[i]), 1e-300));
        }
        for (int i = 0; i < 4; ++i) z[i] = z[i] - w[i];
        if (worst < 1e-26) break;
    }
    for (int i = 0; i < 4; ++i)
        for (int rep = 0; rep < 2; ++rep) {
            cx f, fp;
            quartic_eval_cplx(a, z[i], f, fp);
            const cx dz = cx_div(f, fp);
            if (isfinite(dz.re) && isfinite(dz.im)) z[i] = z[i] - dz;
        }
}

// E of one solution: null vector of W(xi) = eps - (k.k) I + k k^T, unit length (Hermitian norm)
PRT_DEV cvec3 null_vector_cplx(const cx *__restrict__ eps, const cvec3 &kv) {
    const cx k2 = cv_dot(kv, kv);
    const cvec3 w0 = cvec3{eps[0] - k2 + kv.x * kv.x, eps[1] + kv.x * kv.y, eps[2] + kv.x * kv.z};
    const cvec3 w1 = cvec3{eps[3] + kv.y * kv.x, eps[4] - k2 + kv.y * kv.y, eps[5] + kv.y * kv.z};
    const cvec3 w2 = cvec3{eps[6] + kv.z * kv.x, eps[7] + kv.z * kv.y, eps[8] - k2 + kv.z * kv.z};
    const cvec3 c01 = cv_cross(w0, w1), c12 = cv_cross(w1, w2), c20 = cv_cross(w2, w0);
    const double n01 = cv_norm2(c01), n12 = cv_norm2(c12), n20 = cv_norm2(c20);
    cvec3 e = c01;
    double nn = n01;
    if (n12 > nn) {
        e = c12;
        nn = n12;
    }
    if (n20 > nn) {
        e = c20;
        nn = n20;
    }
    const double scale = cv_norm2(w0) + cv_norm2(w1) + cv_norm2(w2);
    if (!(nn > 1e-24 * scale * scale)) {
        // W has rank <= 1 (two sheets of the dispersion surface touch): any vector orthogonal (bilinear) to its
        // one row -- the reference's pick there is LAPACK's, no parity target
        cvec3 r = w0;
        double nr = cv_norm2(w0);
        if (cv_norm2(w1) > nr) {
            r = w1;
            nr = cv_norm2(w1);
        }
        if (cv_norm2(w2) > nr) r = w2;
        const double ax = cx_abs2(r.x), ay = cx_abs2(r.y), az = cx_abs2(r.z);
        const cvec3 u = (ax <= ay && ax <= az) ? cvec3{cx{1, 0}, cx{0, 0}, cx{0, 0}}
                                                : ((ay <= az) ? cvec3{cx{0, 0}, cx{1, 0}, cx{0, 0}}
                                                              : cvec3{cx{0, 0}, cx{0, 0}, cx{1, 0}});
        e = cv_cross(r, u);
        nn = cv_norm2(e);
    }
    const double inv = 1.0 / sqrt(nn);
    return cvec3{inv * e.x, inv * e.y, inv * e.z};
}

struct aniso_solution_cplx {
    vec3 k_re, k_im;  // wave vector, global frame
    vec3 d;           // unit Poynting direction, global frame
    vec3 e_re, e_im;  // E field, global frame
};

// unit Poynting direction Re(|E|^2 k - (E.k) E*) / |.| (ray.py:136-152; calcPoytingVectorNorm)
PRT_DEV vec3 poynting_dir_cplx(const cvec3 &k, const cvec3 &E) {
    const double e2 = cv_norm2(E);
    const cx ke = cv_dot(k, E);
    const cx sx = e2 * k.x - ke * cx_conj(E.x), sy = e2 * k.y - ke * cx_conj(E.y), sz = e2 * k.z - ke * cx_conj(E.z);
    const vec3 S = v3(sx.re, sy.re, sz.re);
    const double inv = 1.0 / sqrt(dot(S, S));
    return v3(S.x * inv, S.y * inv, S.z * inv);
}

// n: unit normal in the frame of the medium; (k_re, k_im): incoming wave vector, global frame
PRT_DEV void interact_anisotropic_cplx(const prt_dev_surface *__restrict__ sf, const double *__restrict__ eps_im,
                                       const vec3 &n, const vec3 &k_re, const vec3 &k_im,
                                       aniso_solution_cplx out[2]) {
    cx eps[9];
    for (int q = 0; q < 9; ++q) eps[q] = cx{sf->eps_re[q], eps_im ? eps_im[q] : 0.0};     // (NULL: a lossless crystal)
    const cvec3 k1 = cv3(matT_vec(sf->B_mat, k_re), matT_vec(sf->B_mat, k_im));
    const cx kn = cv_dot(k1, n);
    const cvec3 kpa = cvec3{k1.x - n.x * kn, k1.y - n.y * kn, k1.z - n.z * kn};
    cx p[5], xi[4];
    xi_polynomial_cplx(eps, n, kpa, p);
    quartic_roots_cplx(p, xi);
    cvec3 kv[4], E[4];
    double sn[4];
    for (int i = 0; i < 4; ++i) {
        kv[i] = cvec3{kpa.x + n.x * xi[i], kpa.y + n.y * xi[i], kpa.z + n.z * xi[i]};
        const cvec3 E0 = null_vector_cplx(eps, kv[i]);
        const double sc = 1.0 / sqrt(1.0 + cx_abs2(xi[i]));      // LAPACK's unit-norm [xi E; E]
        E[i] = cvec3{sc * E0.x, sc * E0.y, sc * E0.z};
        const double e2 = cv_norm2(E[i]);
        const cx ke = cv_dot(kv[i], E[i]);
        const cx en = cx_conj(cv_dot(E[i], n));                   // E* . n
        sn[i] = (e2 * cv_dot(kv[i], n) - ke * en).re;
    }
    // ascending S.n (material.py:144-151); NaNs last
    int order[4] = {0, 1, 2, 3};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3 - a; ++b) {
            const double u = sn[order[b]], v = sn[order[b + 1]];
            if (u > v || (u != u && v == v)) {
                const int t = order[b];
                order[b] = order[b + 1];
                order[b + 1] = t;
            }
        }
    const bool mirror = sf->interaction == PRT_MIRROR;
    for (int b = 0; b < 2; ++b) {
        const int i = mirror ? order[b] : order[2 + b];
        cvec3 kk = kv[i], ee = E[i];
        if (mirror) {  // -(solutions 0, 1), material_anisotropic.py:131-133
            kk = cvec3{-kk.x, -kk.y, -kk.z};
            ee = cvec3{-ee.x, -ee.y, -ee.z};
        }
        out[b].k_re = mat_vec(sf->B_mat, cv_re(kk));
        out[b].k_im = mat_vec(sf->B_mat, cv_im(kk));
        out[b].e_re = mat_vec(sf->B_mat, cv_re(ee));
        out[b].e_im = mat_vec(sf->B_mat, cv_im(ee));
        out[b].d = poynting_dir_cplx(cv3(out[b].k_re, out[b].k_im), cv3(out[b].e_re, out[b].e_im));
    }
}
