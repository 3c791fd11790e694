/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (not part of the product path).
 *
 * C / OpenMP restatement of the reference's sequential trace (mess42/pyrate, package
 * pyrateoptics 0.4.0), formula by formula, used
 *   (1) as the multi-core CPU baseline that bench.py times beside the GPU
 *       ("cpu_baseline": kind "port"), for every single-GPU BASELINE configuration, and
 *   (2) as a second, independently written checker in tests/test_oracle_c.py.
 * Pinned against the golden vectors generated from the real reference
 * (tests/golden/*.npz) exactly like oracle/seqtrace_np.py.
 *
 * Shapes: Conic (closed form), Asphere, XYPolynomials, Biconic (explicit z = F(x, y), root of
 * g(t) = r0z + t dz - F(r0x + t dx, r0y + t dy) from t = 0).  Media: isotropic (Snell / mirror) and
 * anisotropic (constant, possibly complex epsilon tensor: four (xi, E) solutions of the 6x6 pencil,
 * sorted by S.n, ray doubling).
 *
 * Per surface and ray (dense ray set, cumulative masks instead of compaction):
 *   r0 = Bs^T (x - gs), dl = Bs^T d               raytracer/localcoordinates.py:398-413
 *   conic: F, G, H, t = G/(F + sqrt(F^2 + H G))   raytracer/surface_shape.py:305-321
 *   explicit: ExplicitShape.intersect             raytracer/surface_shape.py:448-465 with
 *             Asphere.F / gradF :529-555, XYPolynomials.F / gradF :785-807, Biconic.F / gradF :618-647.
 *             The reference hands the N-vector to scipy.optimize.fsolve (xtol 1e-6); like
 *             oracle/seqtrace_np.py this runs the same root search per ray by Newton's method to machine
 *             precision (cap 30 iterations; a ray that has not settled gets t = NaN)
 *   x_hit = Bs (r0 + t dl) + gs                   :323, localcoordinates.py:383-389
 *   aperture on Ba^T (x_hit - ga)                 raytracer/surface.py:126-135, aperture.py:97-139
 *   normal: xs = Bs^T (x_hit - gs); grad of z - F(x, y) at xs; n = Bm^T Bs grad/|grad|
 *                                                 ray.py:156-161, surface_shape.py:100-112, 208-237
 *   isotropic: Snell / mirror in the material frame          material/material_isotropic.py:137-236
 *              next direction d = k/|k| (E perpendicular k)  ray.py:136-152
 *   anisotropic: MaxwellMaterial.calcXiEigenvectorsNorm (material/material.py:407-454): pencil
 *              A = [[C, K], [-I, 0]], B = -[[M, 0], [0, I]] with M = -I + n n^T, C = kpa n^T + n kpa^T,
 *              K = eps - (kpa.kpa) I + kpa kpa^T (calcXiQEVMatricesNorm, :353-403), solved by LAPACK zggev --
 *              the routine behind the reference's scipy.linalg.eig; its entry point is handed over by the
 *              Python front end (scipy.linalg.cython_lapack), every 6-vector scaled to unit 2-norm like
 *              scipy.linalg.eig does, non-finite eigenvalues dropped, the four of smallest modulus kept;
 *              sortKnormEField (:122-153): ascending S.n, S = Re(|E|^2 k - (k.E) E*) (:214-223);
 *              refract: solutions 2, 3, reflect: -(0, 1), stacked [a, b] (material_anisotropic.py:70-155);
 *              next direction: the unit Poynting direction of (k, E)     ray.py:136-152
 *
 * Table: S records of PRT_C_REC doubles:
 *   [0] curv [1] cc [2..10] Bs [11..13] gs [14] ap_type (0 none, 1 circular, 2 rectangular)
 *   [15] ap_p0 [16] ap_p1 [17..25] Ba [26..28] ga [29] mirror (0/1) [30] n_after [31..39] Bm
 *   [40] shape (0 conic, 1 asphere, 2 xypoly, 3 biconic) [41] number of coefficients / terms / pairs
 *   [42] offset of the surface's data in the coefficient array [43] medium (0 isotropic, 1 anisotropic)
 *   [44..52] eps_re [53..61] eps_im (row major) [62] biconic curvy [63] biconic ccy [64] xypoly normradius
 * Coefficient array: asphere a_0, a_1, ..; xypoly triples (i, j, c); biconic pairs (a, b).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PRT_C_REC 72
#define NEWTON_MAXIT 30
#define NEWTON_TOL 1e-15

typedef double complex zc;
typedef void (*zggev_fn)(char *jobvl, char *jobvr, int *n, zc *a, int *lda, zc *b, int *ldb, zc *alpha, zc *beta,
                         zc *vl, int *ldvl, zc *vr, int *ldvr, zc *work, int *lwork, double *rwork, int *info);
static zggev_fn g_zggev = 0;
void seqtrace_c_set_zggev(void *fn) { g_zggev = (zggev_fn)fn; }
int seqtrace_c_has_zggev(void) { return g_zggev != 0; }

static void matT_vec(const double *B, const double *v, double *o) {
    o[0] = B[0] * v[0] + B[3] * v[1] + B[6] * v[2];
    o[1] = B[1] * v[0] + B[4] * v[1] + B[7] * v[2];
    o[2] = B[2] * v[0] + B[5] * v[1] + B[8] * v[2];
}
static void mat_vec(const double *B, const double *v, double *o) {
    o[0] = B[0] * v[0] + B[1] * v[1] + B[2] * v[2];
    o[1] = B[3] * v[0] + B[4] * v[1] + B[5] * v[2];
    o[2] = B[6] * v[0] + B[7] * v[1] + B[8] * v[2];
}

int seqtrace_c_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static double ipow(double b, int e) {  /* b ** e for small non-negative integers, like numpy's power */
    double r = 1.0;
    for (int q = 0; q < e; ++q) r *= b;
    return r;
}

/* sag F(x, y) and gradient (gx, gy, 1) of z - F(x, y) of a surface's shape, shape frame */
static void shape_eval(const double *r, const double *cf, double x, double y, double *F, double *gx, double *gy,
                       double *gz) {
    const int shape = (int)r[40], nc = (int)r[41];
    const double *c = cf + (int64_t)r[42];
    const double curv = r[0], cc = r[1];
    *gz = 1.0;
    if (shape == 0) { /* Conic.conic_function + getGrad, surface_shape.py:208-237 */
        const double r2 = x * x + y * y;
        const double st = 1 - (1 + cc) * curv * curv * r2;
        const double z = (st > 0) ? curv * r2 / (1 + sqrt(st)) : NAN;
        *F = z;
        *gx = -curv * x;
        *gy = -curv * y;
        *gz = 1. - curv * z * (1 + cc);
    } else if (shape == 1) { /* Asphere.F / gradF, :529-555 */
        const double r2 = x * x + y * y;
        const double sq = sqrt(1 - curv * curv * (1 + cc) * r2);
        double f = curv * r2 / (1 + sq);
        double ax = -curv * x / sq, ay = -curv * y / sq;
        for (int n = 0; n < nc; ++n) {
            f += c[n] * ipow(r2, n + 1);
            ax -= 2. * x * (n + 1) * c[n] * ipow(r2, n);
            ay -= 2. * y * (n + 1) * c[n] * ipow(r2, n);
        }
        *F = f;
        *gx = ax;
        *gy = ay;
    } else if (shape == 2) { /* XYPolynomials.F / gradF, :785-807 */
        const double nr = r[64];
        double f = 0, ax = 0, ay = 0;
        for (int t = 0; t < nc; ++t) {
            const int i = (int)c[3 * t], j = (int)c[3 * t + 1];
            const double co = c[3 * t + 2], nrm = 1. / ipow(nr, i + j);
            f += ipow(x, i) * ipow(y, j) * co * nrm;
            ax -= i * (i >= 1 ? ipow(x, i - 1) : 0.0) * ipow(y, j) * co * nrm;
            ay -= j * ipow(x, i) * (j >= 1 ? ipow(y, j - 1) : 0.0) * co * nrm;
        }
        *F = f;
        *gx = ax;
        *gy = ay;
    } else { /* Biconic.F / gradF, :618-647 */
        const double cx = curv, ccx = cc, cy = r[62], ccy = r[63];
        const double x2 = x * x, y2 = y * y, r2 = x2 + y2, ast2 = x2 - y2;
        const double sq = sqrt(1 - cx * cx * (1 + ccx) * x2 - cy * cy * (1 + ccy) * y2);
        const double u = cx * x2 + cy * y2;
        double f = u / (1 + sq);
        double ax = -cx * x * (cx * (ccx + 1) * u + 2 * (sq + 1) * sq) / ((sq + 1) * (sq + 1) * sq);
        double ay = -cy * y * (cy * (ccy + 1) * u + 2 * (sq + 1) * sq) / ((sq + 1) * (sq + 1) * sq);
        for (int n = 0; n < nc; ++n) {
            const double an = c[2 * n], bn = c[2 * n + 1], w = r2 - bn * ast2;
            f += an * ipow(w, n + 1);
            ax += 2 * an * (n + 1) * x * (bn - 1) * ipow(w, n);
            ay += -2 * an * (n + 1) * y * (bn + 1) * ipow(w, n);
        }
        *F = f;
        *gx = ax;
        *gy = ay;
    }
}

/* Material.propagate -> Surface.intersect: hit point, validity after intersect + aperture, and the unit
 * surface normal in the frame of the medium behind the surface */
static inline void intersect_part(const double *r, const double *cf, const double *x, const double *d, int *ok,
                                  double *xh, double *n) {
    const double c = r[0], cc = r[1];
    const double *Bs = r + 2, *gs = r + 11, *Ba = r + 17, *ga = r + 26, *Bm = r + 31;
    double tmp[3], r0[3], dl[3], p[3];
    tmp[0] = x[0] - gs[0]; tmp[1] = x[1] - gs[1]; tmp[2] = x[2] - gs[2];
    matT_vec(Bs, tmp, r0);
    matT_vec(Bs, d, dl);
    double t;
    if ((int)r[40] == 0) {
        const double F = dl[2] - c * (dl[0] * r0[0] + dl[1] * r0[1] + dl[2] * r0[2] * (1 + cc));
        const double G = c * (r0[0] * r0[0] + r0[1] * r0[1] + r0[2] * r0[2] * (1 + cc)) - 2 * r0[2];
        const double H = -c - cc * c * dl[2] * dl[2];
        const double square = F * F + H * G;
        t = G / (F + sqrt(square));
        if (!(square >= 0)) *ok = 0;
    } else {
        /* per-ray Newton from t = 0 (see the header); valid stays True like the reference's (:462) */
        t = 0.0;
        int settled = 0, noise_floor = 0;
        for (int it = 0; it < NEWTON_MAXIT && !settled; ++it) {
            double F, gx, gy, gz;
            shape_eval(r, cf, r0[0] + t * dl[0], r0[1] + t * dl[1], &F, &gx, &gy, &gz);
            const double g = r0[2] + t * dl[2] - F;
            const double gp = gx * dl[0] + gy * dl[1] + gz * dl[2];
            const double dt = g / gp;
            t -= dt;
            const double scale = fabs(t) > 1.0 ? fabs(t) : 1.0;
            settled = !isfinite(dt) || (fabs(dt) <= NEWTON_TOL * scale);
            noise_floor = fabs(dt) <= 1e-11 * scale;
        }
        /* steps at the rounding noise of g / g' when the cap is reached: converged (csrc/prt_device.h explicit_t) */
        if (!settled && !noise_floor) t = NAN;
    }
    p[0] = r0[0] + dl[0] * t; p[1] = r0[1] + dl[1] * t; p[2] = r0[2] + dl[2] * t;
    mat_vec(Bs, p, xh);
    xh[0] += gs[0]; xh[1] += gs[1]; xh[2] += gs[2];
    const int ap = (int)r[14];
    if (ap != 0) {
        double pa[3];
        tmp[0] = xh[0] - ga[0]; tmp[1] = xh[1] - ga[1]; tmp[2] = xh[2] - ga[2];
        matT_vec(Ba, tmp, pa);
        if (ap == 1) {
            const double rr = pa[0] * pa[0] + pa[1] * pa[1];
            if (!(rr >= r[15] * r[15] && rr <= r[16] * r[16])) *ok = 0;
        } else {
            if (!(pa[0] >= -0.5 * r[15] && pa[0] <= 0.5 * r[15] && pa[1] >= -0.5 * r[16] &&
                  pa[1] <= 0.5 * r[16]))
                *ok = 0;
        }
    }
    /* normal */
    double xs[3], g[3], ng[3], F;
    tmp[0] = xh[0] - gs[0]; tmp[1] = xh[1] - gs[1]; tmp[2] = xh[2] - gs[2];
    matT_vec(Bs, tmp, xs);
    shape_eval(r, cf, xs[0], xs[1], &F, &g[0], &g[1], &g[2]);
    const double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    g[0] /= gn; g[1] /= gn; g[2] /= gn;
    mat_vec(Bs, g, ng);
    matT_vec(Bm, ng, n);
}

/* IsotropicMaterial.refract / reflect: k global in -> k global out, new direction, validity */
static inline void isotropic_part(const double *r, const double *n, double *k, double *d, int *ok) {
    const double *Bm = r + 31;
    double k1[3], kin[3], k2[3];
    matT_vec(Bm, k, k1);
    const double kn = k1[0] * n[0] + k1[1] * n[1] + k1[2] * n[2];
    kin[0] = k1[0] - kn * n[0]; kin[1] = k1[1] - kn * n[1]; kin[2] = k1[2] - kn * n[2];
    const double n2 = r[30];
    const double sq2 = n2 * n2 - (kin[0] * kin[0] + kin[1] * kin[1] + kin[2] * kin[2]);
    const double xi = sqrt(sq2);
    if (!(sq2 > 0) || !isfinite(n[0]) || !isfinite(n[1]) || !isfinite(n[2])) *ok = 0;
    const double sgn = (r[29] != 0.0) ? -1.0 : 1.0;
    k2[0] = sgn * kin[0] + xi * n[0]; k2[1] = sgn * kin[1] + xi * n[1]; k2[2] = sgn * kin[2] + xi * n[2];
    mat_vec(Bm, k2, k);
    const double kl = sqrt(k[0] * k[0] + k[1] * k[1] + k[2] * k[2]);
    d[0] = k[0] / kl; d[1] = k[1] / kl; d[2] = k[2] / kl;
}

/* one surface step through an isotropic interface for one ray */
static inline void surface_step(const double *r, const double *cf, double *x, double *k, double *d, int *ok,
                                int *ok_hit, double *xh_out) {
    double xh[3], n[3];
    intersect_part(r, cf, x, d, ok, xh, n);
    *ok_hit = *ok;
    isotropic_part(r, n, k, d, ok);
    x[0] = xh[0]; x[1] = xh[1]; x[2] = xh[2];
    xh_out[0] = xh[0]; xh_out[1] = xh[1]; xh_out[2] = xh[2];
}

#define RAY_BLOCK 512

/* All-isotropic tables.  Rays are processed in blocks of RAY_BLOCK through all surfaces (state in a small
 * per-thread buffer), so that every output row is written in contiguous runs.  Returns the threads used. */
int seqtrace_c(const double *tab, const double *cf, int S, int64_t N, const double *x0, const double *k0,
               const double *d0, double *x_hit, double *k_out, uint8_t *valid, uint8_t *valid_out,
               int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int used = omp_get_max_threads();
#else
    int used = 1;
#endif
    const int64_t nblk = (N + RAY_BLOCK - 1) / RAY_BLOCK;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t lo = b * RAY_BLOCK;
        const int m = (int)((lo + RAY_BLOCK <= N) ? RAY_BLOCK : (N - lo));
        double x[RAY_BLOCK][3], k[RAY_BLOCK][3], d[RAY_BLOCK][3];
        int ok[RAY_BLOCK];
        for (int j = 0; j < m; ++j) {
            const int64_t i = lo + j;
            x[j][0] = x0[i]; x[j][1] = x0[N + i]; x[j][2] = x0[2 * N + i];
            k[j][0] = k0[i]; k[j][1] = k0[N + i]; k[j][2] = k0[2 * N + i];
            d[j][0] = d0[i]; d[j][1] = d0[N + i]; d[j][2] = d0[2 * N + i];
            ok[j] = 1;
        }
        for (int s = 0; s < S; ++s) {
            const double *r = tab + (int64_t)s * PRT_C_REC;
            double *xr = x_hit + ((int64_t)s * 3) * N + lo;
            double *kr = k_out + ((int64_t)s * 3) * N + lo;
            uint8_t *vr = valid + (int64_t)s * N + lo, *wr = valid_out + (int64_t)s * N + lo;
            for (int j = 0; j < m; ++j) {
                int ok_hit;
                double xh[3];
                surface_step(r, cf, x[j], k[j], d[j], &ok[j], &ok_hit, xh);
                xr[j] = xh[0]; xr[N + j] = xh[1]; xr[2 * N + j] = xh[2];
                kr[j] = k[j][0]; kr[N + j] = k[j][1]; kr[2 * N + j] = k[j][2];
                vr[j] = (uint8_t)ok_hit;
                wr[j] = (uint8_t)ok[j];
            }
        }
    }
    return used;
}

/* ---- anisotropic interface for one ray ----------------------------------------------------------------
 * n: unit normal, k1: incoming wave vector (real part), both in the material frame; eps (row major, complex).
 * out: the two leaving solutions (k, E complex, material frame) in the reference's order. */
static void aniso_solutions(const double *n, const zc *k1, const zc *eps, int mirror, zc kout[2][3], zc eout[2][3]) {
    /* k1 is complex behind an absorbing crystal / an evanescent mode; bilinear products, no conjugate
     * (material_anisotropic.py:72-79) */
    zc kpa[3];
    const zc kn = k1[0] * n[0] + k1[1] * n[1] + k1[2] * n[2];
    for (int q = 0; q < 3; ++q) kpa[q] = k1[q] - kn * n[q];
    int finite = 1;
    for (int q = 0; q < 3; ++q)
        if (!isfinite(n[q]) || !isfinite(creal(kpa[q])) || !isfinite(cimag(kpa[q]))) finite = 0;
    if (!finite || !g_zggev) {
        for (int b = 0; b < 2; ++b)
            for (int q = 0; q < 3; ++q) kout[b][q] = eout[b][q] = NAN + NAN * I;
        return;
    }
    /* calcXiQEVMatricesNorm, material.py:353-403 */
    zc M[3][3], C[3][3], K[3][3];
    const zc kk = kpa[0] * kpa[0] + kpa[1] * kpa[1] + kpa[2] * kpa[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            M[i][j] = (i == j ? -1.0 : 0.0) + n[i] * n[j];
            C[i][j] = kpa[i] * n[j] + n[i] * kpa[j];
            K[i][j] = eps[3 * i + j] - (i == j ? kk : 0.0) + kpa[i] * kpa[j];
        }
    /* A = [[C, K], [-I, 0]], B = -[[M, 0], [0, I]], column major for LAPACK (material.py:426-433) */
    zc A[36], B[36], alpha[6], beta[6], vr[36], vl[1], work[64];
    double rwork[48];
    memset(A, 0, sizeof A);
    memset(B, 0, sizeof B);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i + 6 * j] = C[i][j];
            A[i + 6 * (j + 3)] = K[i][j];
            B[i + 6 * j] = -M[i][j];
        }
    for (int i = 0; i < 3; ++i) {
        A[(i + 3) + 6 * i] = -1.0;
        B[(i + 3) + 6 * (i + 3)] = -1.0;
    }
    int six = 6, one = 1, lwork = 64, info = 0;
    char jn = 'N', jv = 'V';
    g_zggev(&jn, &jv, &six, A, &six, B, &six, alpha, beta, vl, &one, vr, &six, work, &lwork, rwork, &info);
    /* scipy.linalg.eig: w = alpha / beta; every eigenvector scaled to unit 2-norm (_geneig) */
    zc w[6];
    int idx[6], nfin = 0;
    for (int e = 0; e < 6; ++e) {
        w[e] = alpha[e] / beta[e];
        double nrm = 0;
        for (int q = 0; q < 6; ++q) nrm += creal(vr[q + 6 * e]) * creal(vr[q + 6 * e]) + cimag(vr[q + 6 * e]) * cimag(vr[q + 6 * e]);
        nrm = sqrt(nrm);
        for (int q = 0; q < 6; ++q) vr[q + 6 * e] /= nrm;
        if (isfinite(creal(w[e])) && isfinite(cimag(w[e]))) idx[nfin++] = e;
    }
    /* keep the four of smallest modulus (material.py:441-446) */
    for (int a = 1; a < nfin; ++a) { /* insertion sort by |w| */
        const int e = idx[a];
        int b = a - 1;
        while (b >= 0 && cabs(w[idx[b]]) > cabs(w[e])) {
            idx[b + 1] = idx[b];
            --b;
        }
        idx[b + 1] = e;
    }
    if (nfin < 4 || info != 0) {
        for (int b = 0; b < 2; ++b)
            for (int q = 0; q < 3; ++q) kout[b][q] = eout[b][q] = NAN + NAN * I;
        return;
    }
    /* sortKnormEField, material.py:122-153 */
    zc k4[4][3], e4[4][3];
    double sn[4];
    int order[4] = {0, 1, 2, 3};
    for (int s = 0; s < 4; ++s) {
        const int e = idx[s];
        zc ee = 0, ke = 0;
        for (int q = 0; q < 3; ++q) {
            k4[s][q] = kpa[q] + w[e] * n[q];
            e4[s][q] = vr[(q + 3) + 6 * e];
            ee += conj(e4[s][q]) * e4[s][q];
        }
        for (int q = 0; q < 3; ++q) ke += k4[s][q] * e4[s][q];
        sn[s] = 0;
        for (int q = 0; q < 3; ++q) sn[s] += creal(ee * k4[s][q] - ke * conj(e4[s][q])) * n[q];
    }
    for (int a = 1; a < 4; ++a) {
        const int e = order[a];
        int b = a - 1;
        while (b >= 0 && sn[order[b]] > sn[e]) {
            order[b + 1] = order[b];
            --b;
        }
        order[b + 1] = e;
    }
    for (int b = 0; b < 2; ++b) {
        const int s = mirror ? order[b] : order[2 + b];
        const double sg = mirror ? -1.0 : 1.0;
        for (int q = 0; q < 3; ++q) {
            kout[b][q] = sg * k4[s][q];
            eout[b][q] = sg * e4[s][q];
        }
    }
}

/*
 * Tables with anisotropic media: surface by surface over the current (doubling) ray set, concatenated
 * outputs like the engine's (include/prt.h): x_hit = concat_s (3, n_in[s]), valid = concat_s (n_in[s]),
 * k_out_re / k_out_im = concat_s (3, n_out[s]), valid_out = concat_s (n_out[s]); e_out_re / e_out_im
 * (may be NULL) in the layout of k_out, written behind anisotropic interfaces only.  Returns the threads used,
 * -1 if memory ran out.
 */
int seqtrace_c_general(const double *tab, const double *cf, int S, int64_t N, const double *x0, const double *k0,
                       const double *d0, double *x_hit, double *k_out_re, double *k_out_im, double *e_out_re,
                       double *e_out_im, uint8_t *valid, uint8_t *valid_out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int used = omp_get_max_threads();
#else
    int used = 1;
#endif
    int64_t n_final = N;
    for (int s = 0; s < S; ++s)
        if ((int)tab[(int64_t)s * PRT_C_REC + 43] == 1) n_final *= 2;
    /* state of the current bundle: x, Re k, d, Im k, alive -- ping-pong */
    double *st[2];
    uint8_t *al[2];
    for (int b = 0; b < 2; ++b) {
        st[b] = (double *)malloc(sizeof(double) * 12 * (size_t)(n_final > 0 ? n_final : 1));
        al[b] = (uint8_t *)malloc((size_t)(n_final > 0 ? n_final : 1));
        if (!st[b] || !al[b]) return -1;
    }
    int64_t n = N;
    /* a table with absorbing media (complex eps / complex index): complex wave vectors are carried into the isotropic
     * refraction behind the last surface; in lossless tables an evanescent leftover enters with its real part, like
     * in the NumPy oracle */
    int absorbing = 0;
    for (int s = 0; s < S; ++s)
        for (int q = 53; q < 62; ++q)
            if (tab[(int64_t)s * PRT_C_REC + q] != 0.0) absorbing = 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        for (int q = 0; q < 3; ++q) {
            st[0][(0 + q) * n_final + i] = x0[q * N + i];
            st[0][(3 + q) * n_final + i] = k0[q * N + i];
            st[0][(6 + q) * n_final + i] = d0[q * N + i];
            st[0][(9 + q) * n_final + i] = 0.0;
        }
        al[0][i] = 1;
    }
    int64_t off_in = 0, off_out = 0;
    int cur = 0;
    for (int s = 0; s < S; ++s) {
        const double *r = tab + (int64_t)s * PRT_C_REC;
        const int aniso = (int)r[43] == 1;
        const int64_t n_o = aniso ? 2 * n : n;
        const double *si = st[cur];
        double *so = st[1 - cur];
        const uint8_t *ai = al[cur];
        uint8_t *ao = al[1 - cur];
        double *xr = x_hit + 3 * off_in, *kr = k_out_re + 3 * off_out;
        double *ki = k_out_im ? k_out_im + 3 * off_out : 0;
        double *er = e_out_re ? e_out_re + 3 * off_out : 0, *ei = e_out_im ? e_out_im + 3 * off_out : 0;
        uint8_t *vr = valid + off_in, *wr = valid_out + off_out;
        zc eps[9];
        for (int q = 0; q < 9; ++q) eps[q] = r[44 + q] + r[53 + q] * I;
        const double *Bm = r + 31;
        const int mirror = r[29] != 0.0;
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            double x[3], k[3], d[3], xh[3], nrm[3], kim[3];
            for (int q = 0; q < 3; ++q) {
                x[q] = si[(0 + q) * n_final + i];
                k[q] = si[(3 + q) * n_final + i];
                d[q] = si[(6 + q) * n_final + i];
                kim[q] = si[(9 + q) * n_final + i];
            }
            const int alive = ai[i];
            int ok = alive;
            intersect_part(r, cf, x, d, &ok, xh, nrm);
            for (int q = 0; q < 3; ++q) xr[q * n + i] = xh[q];
            vr[i] = (uint8_t)ok;
            const int cplx_iso = !aniso && absorbing && (r[53] != 0.0 || kim[0] != 0.0 || kim[1] != 0.0 || kim[2] != 0.0);
            if (cplx_iso) {
                /* IsotropicMaterial.refract / reflect with a complex incoming k and / or a complex index
                 * (material_isotropic.py:137-236): the LAST surface of a table with absorbing media */
                double k1r[3], k1i[3];
                zc k1[3], kin[3], k2[3];
                matT_vec(Bm, k, k1r);
                matT_vec(Bm, kim, k1i);
                zc kn = 0, kk = 0;
                for (int q = 0; q < 3; ++q) {
                    k1[q] = k1r[q] + k1i[q] * I;
                    kn += k1[q] * nrm[q];
                }
                for (int q = 0; q < 3; ++q) {
                    kin[q] = k1[q] - kn * nrm[q];
                    kk += kin[q] * kin[q];
                }
                const zc nn = r[30] + r[53] * I;
                const zc sq = nn * nn - kk;
                /* NumPy's order of complex numbers: by real part, then imaginary part */
                if (!(creal(sq) > 0 || (creal(sq) == 0 && cimag(sq) > 0)) || !isfinite(nrm[0]) || !isfinite(nrm[1]) ||
                    !isfinite(nrm[2]))
                    ok = 0;
                const zc xi = csqrt(sq);
                const double sgn = mirror ? -1.0 : 1.0;
                for (int q = 0; q < 3; ++q) k2[q] = sgn * kin[q] + xi * nrm[q];
                for (int q = 0; q < 3; ++q) {
                    const zc kg = Bm[3 * q] * k2[0] + Bm[3 * q + 1] * k2[1] + Bm[3 * q + 2] * k2[2];
                    kr[q * n + i] = creal(kg);
                    if (ki) ki[q * n + i] = cimag(kg);
                    so[(0 + q) * n_final + i] = xh[q];
                    so[(3 + q) * n_final + i] = creal(kg);
                    so[(6 + q) * n_final + i] = d[q];
                    so[(9 + q) * n_final + i] = cimag(kg);
                }
                wr[i] = (uint8_t)ok;
                ao[i] = (uint8_t)ok;
            } else if (!aniso) {
                isotropic_part(r, nrm, k, d, &ok);
                for (int q = 0; q < 3; ++q) {
                    kr[q * n + i] = k[q];
                    if (ki) ki[q * n + i] = 0.0;
                    so[(0 + q) * n_final + i] = xh[q];
                    so[(3 + q) * n_final + i] = k[q];
                    so[(6 + q) * n_final + i] = d[q];
                    so[(9 + q) * n_final + i] = 0.0;   /* (an isotropic medium: real wave vectors) */
                }
                wr[i] = (uint8_t)ok;
                ao[i] = (uint8_t)ok;
            } else {
                /* no validity filtering at a crystal interface: every ray still in the bundle gets two
                 * children in a fresh all-valid bundle (material_anisotropic.py:87-100, ray.py:68) */
                double k1r[3], k1i[3];
                zc k1[3], ko[2][3], eo[2][3];
                matT_vec(Bm, k, k1r);
                matT_vec(Bm, kim, k1i);
                for (int q = 0; q < 3; ++q) k1[q] = k1r[q] + k1i[q] * I;
                aniso_solutions(nrm, k1, eps, mirror, ko, eo);
                for (int b = 0; b < 2; ++b) {
                    const int64_t o = i + b * n;
                    zc kg[3], eg[3];
                    for (int q = 0; q < 3; ++q) { /* l2g_dirs(Bm, .) on complex vectors */
                        kg[q] = Bm[3 * q] * ko[b][0] + Bm[3 * q + 1] * ko[b][1] + Bm[3 * q + 2] * ko[b][2];
                        eg[q] = Bm[3 * q] * eo[b][0] + Bm[3 * q + 1] * eo[b][1] + Bm[3 * q + 2] * eo[b][2];
                    }
                    /* RayBundle.returnKtoD, ray.py:136-152 */
                    zc ee = 0, ek = 0;
                    for (int q = 0; q < 3; ++q) {
                        ee += conj(eg[q]) * eg[q];
                        ek += eg[q] * kg[q];
                    }
                    double sv[3], sl = 0;
                    for (int q = 0; q < 3; ++q) {
                        sv[q] = creal(ee * kg[q] - ek * conj(eg[q]));
                        sl += sv[q] * sv[q];
                    }
                    sl = sqrt(sl);
                    for (int q = 0; q < 3; ++q) {
                        kr[q * n_o + o] = creal(kg[q]);
                        if (ki) ki[q * n_o + o] = cimag(kg[q]);
                        if (er) er[q * n_o + o] = creal(eg[q]);
                        if (ei) ei[q * n_o + o] = cimag(eg[q]);
                        so[(0 + q) * n_final + o] = xh[q];
                        so[(3 + q) * n_final + o] = creal(kg[q]);
                        so[(6 + q) * n_final + o] = sv[q] / sl;
                        so[(9 + q) * n_final + o] = cimag(kg[q]);
                    }
                    wr[o] = (uint8_t)alive;
                    ao[o] = (uint8_t)alive;
                }
            }
        }
        off_in += n;
        off_out += n_o;
        n = n_o;
        cur = 1 - cur;
    }
    for (int b = 0; b < 2; ++b) {
        free(st[b]);
        free(al[b]);
    }
    return used;
}
