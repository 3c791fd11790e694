/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (not part of the product path).
 *
 * C / OpenMP restatement of the Conic + isotropic part of the reference's sequential
 * trace (mess42/pyrate, package pyrateoptics 0.4.0), formula by formula, used
 *   (1) as the multi-core CPU baseline that bench.py times beside the GPU
 *       ("cpu_baseline": kind "port"), and
 *   (2) as a second, independently written checker in tests/test_oracle_c.py.
 * Pinned against the golden vectors generated from the real reference
 * (tests/golden/*.npz) exactly like oracle/seqtrace_np.py.
 *
 * Per surface and ray (dense ray set, cumulative masks instead of compaction):
 *   r0 = Bs^T (x - gs), dl = Bs^T d               raytracer/localcoordinates.py:398-413
 *   F, G, H, t = G/(F + sqrt(F^2 + H G))          raytracer/surface_shape.py:305-321
 *   x_hit = Bs (r0 + t dl) + gs                   :323, localcoordinates.py:383-389
 *   aperture on Ba^T (x_hit - ga)                 raytracer/surface.py:126-135, aperture.py:97-139
 *   normal: xs = Bs^T (x_hit - gs); z = sag(xs); grad = (-c x, -c y, 1 - c z (1+cc));
 *           n = Bm^T Bs grad/|grad|               ray.py:156-161, surface_shape.py:100-112, 208-237
 *   Snell / mirror in the material frame          material/material_isotropic.py:137-236
 *   next direction d = k/|k| (E perpendicular k)  ray.py:136-152
 *
 * Table: S records of PRT_C_REC doubles:
 *   [0] curv [1] cc [2..10] Bs [11..13] gs [14] ap_type (0 none, 1 circular, 2 rectangular)
 *   [15] ap_p0 [16] ap_p1 [17..25] Ba [26..28] ga [29] mirror (0/1) [30] n_after [31..39] Bm
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PRT_C_REC 40

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

/* one surface step for one ray; returns validity after intersect+aperture in *ok_hit and the
 * cumulative validity in *ok */
static inline void surface_step(const double *r, double *x, double *k, double *d, int *ok,
                                int *ok_hit, double *xh_out) {
    const double c = r[0], cc = r[1];
    const double *Bs = r + 2, *gs = r + 11, *Ba = r + 17, *ga = r + 26, *Bm = r + 31;
    double tmp[3], r0[3], dl[3], p[3], xh[3];
    tmp[0] = x[0] - gs[0]; tmp[1] = x[1] - gs[1]; tmp[2] = x[2] - gs[2];
    matT_vec(Bs, tmp, r0);
    matT_vec(Bs, d, dl);
    const double F = dl[2] - c * (dl[0] * r0[0] + dl[1] * r0[1] + dl[2] * r0[2] * (1 + cc));
    const double G = c * (r0[0] * r0[0] + r0[1] * r0[1] + r0[2] * r0[2] * (1 + cc)) - 2 * r0[2];
    const double H = -c - cc * c * dl[2] * dl[2];
    const double square = F * F + H * G;
    const double t = G / (F + sqrt(square));
    if (!(square >= 0)) *ok = 0;
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
    *ok_hit = *ok;
    /* normal */
    double xs[3], g[3], ng[3], n[3];
    tmp[0] = xh[0] - gs[0]; tmp[1] = xh[1] - gs[1]; tmp[2] = xh[2] - gs[2];
    matT_vec(Bs, tmp, xs);
    const double r2 = xs[0] * xs[0] + xs[1] * xs[1];
    const double st = 1 - (1 + cc) * c * c * r2;
    const double z = (st > 0) ? c * r2 / (1 + sqrt(st)) : NAN;
    g[0] = -c * xs[0]; g[1] = -c * xs[1]; g[2] = 1. - c * z * (1 + cc);
    const double gn = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    g[0] /= gn; g[1] /= gn; g[2] /= gn;
    mat_vec(Bs, g, ng);
    matT_vec(Bm, ng, n);
    /* refract / reflect */
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
    x[0] = xh[0]; x[1] = xh[1]; x[2] = xh[2];
    xh_out[0] = xh[0]; xh_out[1] = xh[1]; xh_out[2] = xh[2];
}

#define RAY_BLOCK 512

/* Rays are processed in blocks of RAY_BLOCK through all surfaces (state in a small per-thread
 * buffer), so that every output row is written in contiguous runs.  Returns the threads used. */
int seqtrace_c(const double *tab, int S, int64_t N, const double *x0, const double *k0,
               const double *d0, double *x_hit, double *k_out, uint8_t *valid, uint8_t *valid_out,
               int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    int used = omp_get_max_threads();
#else
    int used = 1;
#endif
    const int64_t nblk = (N + RAY_BLOCK - 1) / RAY_BLOCK;
#pragma omp parallel for schedule(static)
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
                surface_step(r, x[j], k[j], d[j], &ok[j], &ok_hit, xh);
                xr[j] = xh[0]; xr[N + j] = xh[1]; xr[2 * N + j] = xh[2];
                kr[j] = k[j][0]; kr[N + j] = k[j][1]; kr[2 * N + j] = k[j][2];
                vr[j] = (uint8_t)ok_hit;
                wr[j] = (uint8_t)ok[j];
            }
        }
    }
    return used;
}
