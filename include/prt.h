/*
 * prt.h -- C ABI of libprt.so, the MI355X (gfx950) sequential-raytrace engine
 * that sits behind mess42/pyrate's RayBundle / Surface.intersect /
 * Material.refract API.
 *
 * Everything here is plain C: pointers, sizes, PODs.  No torch / C++ types.
 * All ray buffers are DEVICE pointers owned by the caller (the Python host
 * uses torch tensors purely as an allocator); the surface table is a HOST
 * array that prt_system_create() copies to the device once.  Every entry point
 * returns 0 (PRT_OK) or a negative error code and never throws; per-ray
 * failure is a mask, like in the reference (surface_shape.py:321,
 * material_isotropic.py:183), never an error.
 *
 * Ray arrays use the reference's layout: (3, N) float64, C-contiguous, i.e.
 * component-major SoA (ray.py:40-44): x[c*N + i].
 *
 * Reference interface each entry point replaces (paths relative to the
 * reference checkout, package pyrateoptics 0.4.0):
 *
 *   prt_trace            OpticalSystem.seqtrace      raytracer/optical_system.py:73-94
 *   prt_trace_ex         the same, every option in one struct; adds the UNIFORM first segment of the
 *                        collimated bundles (OpticalSystemAnalysis.collimated_bundle,
 *                        raytracer/analysis/optical_system_analysis.py:83-122: k and E are one vector
 *                        for the whole bundle)
 *   prt_trace_seq        OpticalElement.seqtrace     raytracer/optical_element.py:324-379
 *   prt_propagate        Material.propagate          raytracer/material/material_isotropic.py:238-247
 *                        Surface.intersect           raytracer/surface.py:116-135
 *                        Conic.intersect             raytracer/surface_shape.py:289-325
 *                        ExplicitShape.intersect     raytracer/surface_shape.py:448-465
 *                          (Asphere :520-606, Biconic :609-706, XYPolynomials :780-858)
 *                        RayBundle.returnKtoD        raytracer/ray.py:136-152
 *   prt_interact         IsotropicMaterial.refract   raytracer/material/material_isotropic.py:163-199
 *                        IsotropicMaterial.reflect   raytracer/material/material_isotropic.py:201-236
 *                        AnisotropicMaterial.refract raytracer/material/material_anisotropic.py:70-113
 *                        AnisotropicMaterial.reflect raytracer/material/material_anisotropic.py:115-155
 *                        RayBundle.getLocalSurfaceNormal  raytracer/ray.py:156-161
 *   prt_interact_cplx    the same calls with complex wave vectors (absorbing media)
 *   prt_shape_eval       Shape.getSag / getGrad / getNormal   raytracer/surface_shape.py:71-112
 *   prt_compact          boolean fancy indexing [:, valid]    raytracer/material/material_isotropic.py:194-199
 *
 * Threads.  Calls on DIFFERENT systems, the handle-free calls (prt_trace_seq and its table cache, the bundle generators,
 * moments, compaction) and prt_system_create / destroy may run concurrently from any number of host threads;
 * prt_last_error() is per thread; an arena has a lock of its own.  ONE system is used by one thread at a time:
 * prt_system_update rewrites the host copy of the table and the staging ring without a lock (the reference's objects are
 * not thread-safe either).  tests/test_hostemu.py runs both statements on the host build of these sources under
 * ThreadSanitizer.
 *
 * In-reference precedent for a C-ABI plugin behind Shape.intersect:
 * raytracer/surface_shape_zmxdll.py:228-404 (ctypes.CDLL, int return, caller-
 * allocated structs).
 */
#ifndef PRT_H
#define PRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_ABI_VERSION 7
#define PRT_COMPACT_MAX_ROWS 24 /* rows prt_compact moves per call */
#define PRT_MAX_COEFFS 128 /* asphere A2.. coefficients and / or XY-polynomial terms */

/* ---- error codes ---------------------------------------------------- */
#define PRT_OK 0
#define PRT_ERR_INVALID_ARG (-1)
#define PRT_ERR_UNSUPPORTED (-2) /* shape / material outside the engine's scope */
#define PRT_ERR_DEVICE (-3)      /* HIP runtime error; see prt_last_error() */
#define PRT_ERR_NO_DEVICE (-4)
#define PRT_ERR_NOMEM (-5)

/* ---- enums (int32 in the POD) --------------------------------------- */
enum {
    PRT_SHAPE_CONIC = 0,
    PRT_SHAPE_ASPHERE = 1,
    PRT_SHAPE_XYPOLY = 2,  /* also the Zernike shapes: the host expands them into monomials */
    PRT_SHAPE_BICONIC = 3,
    PRT_SHAPE_COMBO = 4,   /* asphere_scale * asphere(x, y) + xy polynomial: LinearCombination of
                              explicit shapes whose frames differ by translations
                              (surface_shape.py:709-775) */
    PRT_SHAPE_GRIDSAG = 5  /* bicubic tensor-product B-spline through a sag grid (GridSag,
                              surface_shape.py:861-925: scipy RectBivariateSpline = FITPACK) */
};
enum { PRT_AP_NONE = 0, PRT_AP_CIRCULAR = 1, PRT_AP_RECTANGULAR = 2 };
enum { PRT_REFRACT = 0, PRT_MIRROR = 1 };
enum { PRT_MAT_ISOTROPIC = 0, PRT_MAT_ANISOTROPIC = 1 };
enum { PRT_ANISO_GENERAL = 0, PRT_ANISO_ISOTROPIC = 1, PRT_ANISO_UNIAXIAL = 2 };

/* frame_flags bits: let the kernels skip 3x3 mat-vecs that are identities */
#define PRT_FRAME_SHAPE_IDENTITY 1 /* B_shape == I (pure translation)          */
#define PRT_FRAME_AP_IS_SHAPE 2    /* aperture frame == shape frame            */
#define PRT_FRAME_MAT_IDENTITY 4   /* B_mat == I                               */

/* trace modes */
#define PRT_MODE_PATH 0  /* write hit point / outgoing k / valid at every surface */
#define PRT_MODE_IMAGE 1 /* write only the last surface's                          */
/* OR-ed into the mode of prt_trace / prt_trace_timed / prt_trace_moments (all-isotropic tables):
 * the masks of a ray-surface record share one byte, valid[i] = (valid after intersect +
 * aperture) | (valid after the interaction) << 1 | (Newton iteration cap hit, see nonconv) << 2,
 * and valid_out is not written (may be NULL): 49 instead of 50 bytes and one store stream less
 * per surface */
#define PRT_MODE_FLAGS 2

/*
 * One record per traced surface, in sequence order (the flattened
 * [(elementkey, [(surfkey, options), ...]), ...] sequence).  Matrices are
 * row-major 3x3 = LocalCoordinates.localbasis; g = .globalcoordinates
 * (localcoordinates.py:264-295).
 */
typedef struct prt_surface {
    int32_t shape_type;  /* PRT_SHAPE_*                                              */
    int32_t n_coeffs;    /* asphere: #A coefficients; xypoly: #terms; biconic: #(A,B) pairs */
    int32_t ap_type;     /* PRT_AP_*                                                 */
    int32_t interaction; /* PRT_REFRACT / PRT_MIRROR (sequence option "is_mirror")   */
    int32_t mat_type;    /* medium the ray is in AFTER the interaction               */
    int32_t frame_flags; /* PRT_FRAME_* (computed by the host, only an optimisation) */
    int32_t newton_maxit; /* explicit shapes: iteration cap (0 -> 30)                */
    int32_t aniso_class; /* PRT_ANISO_* (host classification of eps, see below)      */
    double curv;        /* conic/asphere curvature;  biconic: curvx;  xypoly: unused  */
    double cc;          /* conic constant;  biconic: ccx                             */
    double coeffs[PRT_MAX_COEFFS]; /* asphere: A2, A4, ...  xypoly: c / normradius^(i+j);
                                      biconic: A2, B2, A4, B4, ... (pairs)            */
    int32_t xpow[PRT_MAX_COEFFS];  /* xypoly term powers                             */
    int32_t ypow[PRT_MAX_COEFFS];
    double B_shape[9], g_shape[3]; /* shape.lc                                       */
    double B_ap[9], g_ap[3];       /* aperture.lc                                    */
    double ap_p0, ap_p1; /* circular: minradius, maxradius; rectangular: width, height */
    double B_mat[9];     /* lc of the medium after the interaction                   */
    double n_after;      /* isotropic: refractive index of that medium               */
    double eps_re[9], eps_im[9]; /* anisotropic: its (constant) epsilon tensor       */
    /* A real tensor whose antisymmetric part is <= 1e-14 of its largest entry (a symmetric tensor rotated into place,
     * R diag R^T) is used as its symmetric part: the crystal solver's cheapest route needs eps[i][j] == eps[j][i]. */
    /* Absorbing media.  eps_im != 0: an absorbing crystal (the reference accepts a complex tensor,
     * material_anisotropic.py:52-56).  ISOTROPIC records: eps_im[0] = Im(n), an absorbing isotropic medium (complex
     * refractive index, material_isotropic.py:59-63, 137-161); the other entries of eps_re / eps_im are unused
     * there.  Every wave vector behind the first absorbing medium is complex.  Supported wherever the reference's
     * result is defined: inside crystals (every later medium anisotropic), and an isotropic medium -- absorbing or
     * not -- behind the LAST surface of the table (its complex k = k_inplane + xi n is unique).  An isotropic medium
     * behind an absorbing one BEFORE the last surface is refused by prt_system_create (PRT_ERR_UNSUPPORTED): the
     * reference takes E there from an SVD whose null space is two-dimensional for a complex k
     * (material_isotropic.py:72-128), so the direction of the ray, and every later hit point, is LAPACK's arbitrary
     * pick.  Such tables are traced by prt_trace_ex with k_out_im (required: the imaginary parts of the wave
     * vectors), tight arrays (pitch 0), the concatenated layout, one launch pair per surface
     * (csrc/prt_aniso_cplx.h); one surface at a time: prt_interact_cplx (prt_interact has no complex k and refuses). */
    /* host-side classification of a real symmetric eps (an optimisation AND what keeps the
     * touching-sheet directions well conditioned, see csrc/prt_aniso.h):
     * ISOTROPIC: eps = aniso_eo I;  UNIAXIAL: eps = aniso_eo I + (aniso_ee-aniso_eo) c c^T,
     * c = aniso_axis (unit, material frame);  GENERAL: anything else. */
    double aniso_eo, aniso_ee, aniso_axis[3];
    double curv_y, cc_y; /* biconic: curvature and conic constant of the y section         */
    /* COMBO: coeffs[0 .. n_asphere) are the asphere's A2, A4, ...; coeffs[n_asphere .. n_coeffs)
     * with xpow / ypow are the XY terms; the conic + asphere part is multiplied by asphere_scale */
    int32_t n_asphere, pad_;
    double asphere_scale;
    /* GRIDSAG: the spline in FITPACK's (tx, ty, c) form, degree 3 in both directions.  aux points to
     * grid_nx knots tx, then grid_ny knots ty, then (grid_nx-4)*(grid_ny-4) coefficients (row-major,
     * x index first).  prt_system_create takes a HOST pointer here and keeps its own device copy;
     * the caller's array is not referenced after the call. */
    int32_t grid_nx, grid_ny;
    const double *aux;
} prt_surface_t;

/* opaque: a surface table on the device.  prt_system_create repacks the records: what the kernels
 * read is 504 bytes per surface (the coefficient arrays above become pointers into one side array
 * holding only the entries in use; csrc/prt_device.h prt_dev_surface). */
typedef struct prt_system prt_system_t;

/* ---- library / device ----------------------------------------------- */
int32_t prt_abi_version(void);
int32_t prt_device_count(void);               /* < 0: error code                  */
const char *prt_strerror(int32_t code);
const char *prt_last_error(void);             /* thread-local detail of last error */
int32_t prt_sizeof_surface(void);             /* sizeof(prt_surface_t): ABI check */

/* ---- system (surface table) ----------------------------------------- */
int32_t prt_system_create(const prt_surface_t *table, int32_t n_surfaces, int32_t device,
                          prt_system_t **out);
int32_t prt_system_destroy(prt_system_t *sys);
/* The table of an existing system replaced IN PLACE, ordered on `stream`: what a loop of "change a parameter, trace"
 * calls per step (the reference's optimiser, optimize/optimize.py:73-91, re-reads the object graph on every
 * seqtrace; here the changed table costs one small asynchronous copy from page-locked staging memory -- no allocation,
 * no blocking copy).  The new table must fit the system's device arrays: the same number of surfaces, a coefficient
 * side array no longer than the one allocated at creation, the same absorbing-media status and, with crystals, a
 * walk of the same length; otherwise PRT_ERR_UNSUPPORTED and nothing is touched (create a new system).  Launches
 * enqueued on `stream` before the call see the old table, launches enqueued afterwards the new one; traces of this
 * system on other streams must have completed.  A device error AFTER the first of the call's copies was enqueued
 * leaves the device with pieces of two tables: PRT_ERR_DEVICE, and every later call on the system returns
 * PRT_ERR_DEVICE too -- destroy it. */
int32_t prt_system_update(prt_system_t *sys, const prt_surface_t *table, int32_t n_surfaces, void *stream);
int32_t prt_system_num_surfaces(const prt_system_t *sys);
/* Which layout of the path arrays prt_trace_ex takes for this table (the ONE place that decides; callers that
 * allocate the arrays ask instead of re-deriving it from the table):
 *   PRT_LAYOUT_ROW_PITCHED            all-isotropic table: (S, 3, out_pitch) rows, any out_pitch >= n0
 *   PRT_LAYOUT_CONCATENATED_PITCHED   crystals, fused walk (k_trace_general): concatenated layout with a ray pitch
 *                                     P >= n0 (prt_crystal_pitch), or tight (pitch 0)
 *   PRT_LAYOUT_CONCATENATED_TIGHT     crystals, per-surface march (more than 8 crystal interfaces, absorbing media,
 *                                     PRT_GENERAL_PER_SURFACE set): tight arrays only (pitch 0)
 * Negative: an error code. */
#define PRT_LAYOUT_ROW_PITCHED 0
#define PRT_LAYOUT_CONCATENATED_PITCHED 1
#define PRT_LAYOUT_CONCATENATED_TIGHT 2
int32_t prt_system_layout(const prt_system_t *sys);
/* rays entering / leaving every surface for n0 input rays (anisotropic
 * interfaces double the count, material_anisotropic.py:87-100).  n_in, n_out:
 * host arrays of n_surfaces int64. */
int32_t prt_system_ray_counts(const prt_system_t *sys, int64_t n0, int64_t *n_in, int64_t *n_out);

/*
 * Whole sequence: OpticalSystem.seqtrace for splitup=False.
 *   x0, k0        (3,n0) start points / wave vectors (global frame, |k| = n), rows in_pitch
 *                 elements apart (0 = n0, i.e. C-contiguous (3,n0)).
 *   e0_re, e0_im  (3,n0) E field of the first segment (same pitch) or NULL.  Used only for the
 *                 first segment's direction d = S/|S| (ray.py:136-152); NULL e0_re
 *                 means E = (0,1,0) (ray.py:71-73), NULL e0_im means real E.
 *   mode PATH, all-isotropic table:  row-pitched arrays, element (row, ray i) at row*out_pitch+i:
 *                 x_hit, k_out  (S,3,out_pitch) doubles;  valid, valid_out (S,out_pitch) bytes;
 *                 out_pitch 0 = n0 (tight).  Use prt_recommended_pitch(): rows that do not start
 *                 on a 128-B line cost ~35 % of the HBM write bandwidth.
 *   mode PATH, table with anisotropic media: concatenated
 *                 x_hit = concat_s (3,n_in[s]),  valid = concat_s (n_in[s]),
 *                 k_out = concat_s (3,n_out[s]), valid_out = concat_s (n_out[s]),
 *                 n_in / n_out = prt_system_ray_counts(sys, P): the layout of a bundle of P = out_pitch rays
 *                 (0 = n0, tight) of which only the first n0 of every branch exist -- ray i of branch b of surface s
 *                 at b*P + i of its row; the engine writes nothing into the P - n0 padding slots of a branch
 *                 (masks there stay what the caller put: zero them once).  With P a multiple of 16 every row starts
 *                 on a 128-B line: 0.124 instead of 0.151 ms on BASELINE configs[3] (998012 rays), so a caller that
 *                 allocates the arrays itself should pass P = n0 rounded up to 128 (prt_crystal_pitch).
 *                 in_pitch: row pitch of the inputs as for isotropic tables.  More than 8 crystal interfaces (the
 *                 per-surface march): tight arrays only.
 *   mode IMAGE:   the same four arrays for the last surface only.
 *   valid_out may be NULL.
 *   nonconv (may be NULL; layout of valid): 1 where the Newton iteration of an explicit shape
 *   (Asphere, Biconic, XYPolynomials, ...) ended at its iteration cap instead of converging, 0
 *   elsewhere.  Converged: a step <= 1e-15 relative to max(1, |t|), or -- at the cap -- a last step <= 1e-11
 *   (a shape whose evaluation is noisy far from its axis stalls at 1e-14; that ray HAS converged and keeps its point).  The reference has no such mask -- ExplicitShape.intersect reports valid = True for
 *   every ray, converged or not (surface_shape.py:462) -- and `valid` stays reference-compatible: a
 *   capped ray keeps valid = 1, gets a NaN hit point and is dropped by the next refraction.  nonconv is
 *   what lets a caller tell such a ray from one that left the domain of the shape.
 *   With PRT_MODE_FLAGS the same bit is also bit 2 of the flags byte (no extra array needed).
 *   valid is the reference's cumulative mask after intersect + aperture
 *   (ray.py:100, surface.py:135); valid_out additionally ANDs the refraction
 *   checks (material_isotropic.py:183) -- it is what [:, valid] compaction uses.
 *   stream: hipStream_t (NULL = default stream).  Asynchronous.
 */
int64_t prt_recommended_pitch(int64_t n); /* n rounded up to 512 elements (4 KiB of doubles) */
int64_t prt_crystal_pitch(int64_t n);     /* n rounded up to 128: ray pitch of the concatenated layout */
int32_t prt_trace(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                  const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                  int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                  uint8_t *valid_out, uint8_t *nonconv, void *stream);

/*
 * ---- the general form: every option of the whole-sequence trace in one struct -------------------
 * prt_trace, prt_trace_seq, prt_trace_fields, prt_trace_moments and prt_trace_timed are thin wrappers
 * that fill this struct; prt_trace_ex additionally offers
 *
 *   the UNIFORM first segment (ABI v5).  The bundles of the reference's analysis layer are collimated
 *   (OpticalSystemAnalysis.collimated_bundle, analysis/optical_system_analysis.py:83-122): k0 and E0 are
 *   ONE vector for all rays, yet as arrays they cost 48 B/ray of loads (72 of the 268 B/ray of BASELINE
 *   configs[2]).  With k0 == NULL every ray has k_uniform, the first segment's direction comes from the
 *   uniform E (or is k/|k|, or is given), only x0 is read -- and the results are bit-identical to the
 *   array form (same per-ray arithmetic on the same values).
 *
 * first_dir: where the unit direction d of the first segment comes from (RayBundle.returnKtoD,
 * ray.py:136-152: the Poynting direction of (k, E))
 *   PRT_FIRST_E          from k and E: e0_re / e0_im arrays, e0_re == NULL means E = (0,1,0)
 *                        (ray.py:71-73); needs k0 when arrays are given
 *   PRT_FIRST_K          d = k/|k| (E perpendicular to k: any bundle that left an isotropic interface)
 *   PRT_FIRST_DIR        e0_re holds the unit directions themselves, (3,n0) (prt_trace_seq's d0)
 *   PRT_FIRST_E_UNIFORM  from k and the uniform E = e_uniform_re + i e_uniform_im
 *   PRT_FIRST_DIR_UNIFORM  d = e_uniform_re for every ray
 * k0 == NULL (uniform k) goes with PRT_FIRST_E (E = ey only), PRT_FIRST_K, PRT_FIRST_E_UNIFORM and
 * PRT_FIRST_DIR_UNIFORM; the uniform E / direction kinds go with k0 == NULL only.
 *
 * Outputs and modes exactly as described at prt_trace (e_out_*: prt_trace_fields; moments_*:
 * prt_trace_moments; timed_iters > 0: prt_trace_timed, *ms_avg receives the average milliseconds per
 * launch -- of the march kernel only when moments are requested too).  struct_bytes must be
 * sizeof(prt_trace_args_t): a caller built against another layout is refused (PRT_ERR_INVALID_ARG).
 */
enum { PRT_FIRST_E = 0, PRT_FIRST_K = 1, PRT_FIRST_DIR = 2, PRT_FIRST_E_UNIFORM = 3, PRT_FIRST_DIR_UNIFORM = 4 };
typedef struct prt_trace_args {
    int32_t struct_bytes; /* sizeof(prt_trace_args_t)                                     */
    int32_t mode;         /* PRT_MODE_PATH / PRT_MODE_IMAGE, optionally | PRT_MODE_FLAGS   */
    int64_t n0;           /* rays entering the first surface                              */
    /* first segment */
    int64_t in_pitch;     /* row pitch of x0 / k0 / e0 in elements, 0 = n0               */
    const double *x0;     /* (3,n0) device                                                */
    const double *k0;     /* (3,n0) device, or NULL: every ray has k_uniform              */
    const double *e0_re, *e0_im; /* (3,n0) device or NULL (see first_dir)                 */
    int32_t first_dir;    /* PRT_FIRST_*                                                  */
    int32_t pad0_;
    double k_uniform[3];
    double e_uniform_re[3], e_uniform_im[3];
    /* outputs (layouts: prt_trace) */
    int64_t out_pitch;
    double *x_hit, *k_out;
    uint8_t *valid, *valid_out, *nonconv;
    double *e_out_re, *e_out_im;  /* E behind crystal interfaces (prt_trace_fields) or NULL */
    /* Evanescent modes at crystal interfaces (PRT_MODE_PATH, fused march): the reference carries them as COMPLEX
     * wave vectors (material/material.py:407-454); the engine does not trace them (their slot of k_out is NaN and
     * everything behind it invalid).  With k_out_im != NULL (layout of k_out) such a slot receives the complex k
     * instead -- real part in k_out, imaginary part here, 0 for propagating modes -- from a post-pass over the
     * crystal surfaces (of a conjugate pair the root with Im(xi) > 0; the reference's pick is its sort's).
     * Tables with absorbing media (prt_surface_t.eps_im): REQUIRED, in both modes -- the wave vectors are
     * complex, k_out + i k_out_im (PRT_MODE_IMAGE: (3, n_out[S-1]) like k_out). */
    double *k_out_im;
    /* image-plane redirect (all-isotropic tables, PRT_MODE_PATH, 16-B aligned rows): when x_img != NULL the record
     * of the LAST surface goes to x_img, k_img (3 rows of img_pitch elements each) and valid_img (the mask byte
     * row: `valid`, or the flags byte with PRT_MODE_FLAGS; valid_out_img: the second mask row without
     * PRT_MODE_FLAGS, may be NULL) instead of to its rows of x_hit / k_out / valid / valid_out -- which are then
     * not written.  This is how a ray-sharded trace deposits its image plane straight into its slot of the
     * all-gather's receive buffer (the collective then runs in place, no copy of the shard's own rows). */
    double *x_img, *k_img;
    uint8_t *valid_img, *valid_out_img;
    int64_t img_pitch;
    /* fused image-plane moments (prt_trace_moments): requested by moments_out7_dev != NULL */
    const double *moments_ref3;   /* HOST, 3 doubles, or NULL: vertex of the last surface   */
    double *moments_out7_dev, *moments_scratch_dev;
    /* timing (prt_trace_timed): timed_iters > 0 */
    int32_t timed_iters, pad1_;
    double *ms_avg;               /* HOST                                                  */
    void *stream;                 /* hipStream_t, NULL = default stream                    */
} prt_trace_args_t;
int32_t prt_sizeof_trace_args(void);
int32_t prt_trace_ex(const prt_system_t *sys, const prt_trace_args_t *args);

/*
 * The same in one call, without a handle (the form SURVEY.md section 8b proposes): the table is uploaded
 * on first use and kept by content (the last 8 tables of the process).
 *   x0, k0        (3,n) tight arrays on the device
 *   d0            (3,n) unit directions of the first segment (RayBundle.returnKtoD of the initial
 *                 bundle, ray.py:136-152) or NULL: d = k/|k| (E perpendicular to k)
 *   ray_id        accepted for the caller's bookkeeping and ignored: outputs are dense, column i
 *                 belongs to input ray i
 *   mode          PRT_MODE_PATH: x_hit, k_out (S,3,n), valid (S,n) tight -- tables with crystals: the
 *                 concatenated layout of prt_trace; PRT_MODE_IMAGE: the last surface only
 *   valid         cumulative mask after intersect + aperture (the reference's RayBundle.valid);
 *   nonconv       optional, as in prt_trace
 * Asynchronous on `stream` (tables with grid-sag surfaces are not cached and synchronise it).
 */
int32_t prt_trace_seq(const prt_surface_t *table, int32_t n_surfaces, int64_t n, const double *x0,
                      const double *k0, const double *d0, const int64_t *ray_id, int32_t mode, double *x_hit,
                      double *k_out, uint8_t *valid, uint8_t *nonconv, int32_t device, void *stream);

/*
 * prt_trace for all-isotropic tables that also reduces the image-plane moments of the traced
 * bundle in the same launch: out7_dev = {count, sum v (3), sum v*v (3)}, v = last hit point - ref,
 * over the rays still valid after the last interaction (= valid_out of the last surface).
 * ref3 (host, 3 doubles) may be NULL: the vertex of the last surface.  From these sums the caller
 * gets centroid = ref + S1/n and RMS spot radius^2 = (sum S2 - |S1|^2/n)/(n-1), the quantities of
 * RayBundleAnalysis.get_centroid_position / get_rms_spot_size (analysis/ray_analysis.py:44-86);
 * being plain sums they combine over ray shards with ONE 7-double all-reduce.  The summation
 * order is fixed (bit-reproducible).  scratch_dev: prt_trace_moments_scratch_doubles(n0) doubles.
 * Asynchronous on `stream`.
 */
int64_t prt_trace_moments_scratch_doubles(int64_t n0);
int32_t prt_trace_moments(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                          const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                          int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                          uint8_t *valid_out, const double *ref3, double *out7_dev,
                          double *scratch_dev, void *stream);

/*
 * prt_trace in the concatenated layout (tight inputs) that also returns the E field of the rays
 * leaving every crystal interface: e_out_re / e_out_im (may be NULL) have the layout of k_out;
 * blocks of isotropic surfaces are left untouched (there E is not computed, see prt_efield_perp).  This is what
 * AnisotropicMaterial.refract stores in the new RayBundle (material_anisotropic.py:91-99).
 */
int32_t prt_trace_fields(const prt_system_t *sys, int64_t n0, const double *x0, const double *k0,
                         const double *e0_re, const double *e0_im, int32_t mode, double *x_hit,
                         double *k_out, double *e_out_re, double *e_out_im, uint8_t *valid,
                         uint8_t *valid_out, void *stream);

/*
 * Material.propagate(raybundle, surface): intersect + aperture for one surface.
 *   x, k (3,n); dir (3,n) unit ray directions or NULL (then e_re/e_im as in
 *   prt_trace, or k/|k| if e_re is NULL and use_default_e == 0).
 *   valid_in (n) or NULL (= all valid).  Writes x_hit (3,n), valid (n) and, if nonconv is
 *   not NULL, nonconv (n) as described at prt_trace.
 */
int32_t prt_propagate(const prt_system_t *sys, int32_t surface, int64_t n, const double *x,
                      const double *k, const double *dir, const double *e_re, const double *e_im,
                      int32_t use_default_e, const uint8_t *valid_in, double *x_hit,
                      uint8_t *valid, uint8_t *nonconv, void *stream);

/*
 * Material.refract / reflect at one surface (which one: table[surface].interaction).
 *   x_hit, k (3,n), valid_in (n) or NULL.
 *   isotropic:   k_out (3,n), valid_out (n); dir_out may be NULL.
 *   anisotropic: k_out (3,2n) in [sol2, sol3] stacking (material_anisotropic.py:87-95),
 *                dir_out (3,2n) unit Poynting directions (needed by the next propagate),
 *                e_out_re / e_out_im (3,2n) or NULL, valid_out (2n) all 1 (ray.py:68;
 *                valid_in is ignored: the reference does no validity filtering there).
 */
int32_t prt_interact(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit,
                     const double *k, const uint8_t *valid_in, double *k_out, double *dir_out,
                     double *e_out_re, double *e_out_im, uint8_t *valid_out, void *stream);

/*
 * prt_propagate / prt_interact for BIG isotropic bundles: every array has a ROW PITCH (elements between the starts of
 * two component rows; 0 = n, tight), so that rows can start on 128-B lines (prt_recommended_pitch) -- a thread owns two
 * adjacent rays and moves them with one 16-B access per row when all pointers are 16-B aligned and all pitches even
 * (any other layout is taken too, 8 B at a time).  Same reference interface as above (Material.propagate ->
 * Surface.intersect, raytracer/material/material_isotropic.py:238-247, raytracer/surface.py:116-135;
 * IsotropicMaterial.refract / reflect, material_isotropic.py:163-236), same results to rounding, same masks.
 *   prt_propagate_rows: x (3, n | x_pitch); k, dir, e_re, e_im (3, n | k_pitch), selected as in prt_propagate; with
 *     dir == NULL, e_re == NULL and use_default_e == 0 the ray direction is k itself (parallel to the Poynting vector
 *     behind an isotropic interface, ray.py:136-152; the intersection is homogeneous in the direction).
 *     Writes x_hit (3, n | out_pitch), valid (n), nonconv (n) or NULL.
 *   prt_interact_rows: records with an ISOTROPIC, lossless medium behind the surface only (PRT_ERR_UNSUPPORTED
 *     otherwise: crystals double the rays, prt_interact).  x_hit (3, n | x_pitch), k (3, n | k_pitch), valid_in (n) or
 *     NULL -> k_out (3, n | out_pitch), valid_out (n) or NULL, dir_out (3, n | out_pitch) or NULL (unit k; nobody
 *     needs it between two of these calls).
 * 148 B of HBM traffic per ray and surface for the pair (each call reads the 49-B state it works on and writes 25 B).
 *   prt_surface_step_rows: BOTH calls of one surface in one launch -- the loop body of OpticalElement.seqtrace
 *     (raytracer/optical_element.py:336-375: propagate to the surface, then refract | reflect there) for records with an
 *     isotropic, lossless medium behind the surface: x, k (+ dir / E as in prt_propagate_rows), valid_in -> x_hit, k_out
 *     (3, n | out_pitch), valid (n: hit the surface inside the aperture), valid_out (n: and left it), nonconv (n) or NULL.
 *     98 B per ray and surface (SURVEY.md 8d's figure: the 49-B state read once, the 49-B record written once).
 */
int32_t prt_propagate_rows(const prt_system_t *sys, int32_t surface, int64_t n, const double *x, int64_t x_pitch,
                           const double *k, int64_t k_pitch, const double *dir, const double *e_re,
                           const double *e_im, int32_t use_default_e, const uint8_t *valid_in, double *x_hit,
                           int64_t out_pitch, uint8_t *valid, uint8_t *nonconv, void *stream);
int32_t prt_interact_rows(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit, int64_t x_pitch,
                          const double *k, int64_t k_pitch, const uint8_t *valid_in, double *k_out,
                          int64_t out_pitch, double *dir_out, uint8_t *valid_out, void *stream);
int32_t prt_surface_step_rows(const prt_system_t *sys, int32_t surface, int64_t n, const double *x, int64_t x_pitch,
                              const double *k, int64_t k_pitch, const double *dir, const double *e_re,
                              const double *e_im, int32_t use_default_e, const uint8_t *valid_in, double *x_hit,
                              double *k_out, int64_t out_pitch, uint8_t *valid, uint8_t *valid_out, uint8_t *nonconv,
                              void *stream);

/*
 * The same plugin call with COMPLEX wave vectors (absorbing media: a complex epsilon tensor,
 * material_anisotropic.py:52-56, 70-113; a complex refractive index, material_isotropic.py:137-199; and whatever
 * comes behind them): k = k_re + i k_im in (k_im NULL: real), k_out_re / k_out_im out, both required.
 *   anisotropic: (3,2n) outputs in [sol2, sol3] stacking, dir_out (3,2n) required, e_out_* (3,2n) or NULL,
 *                valid_out (2n) all 1; lossless crystals are taken too (a complex k may enter one).
 *   isotropic:   k = k_inplane + xi n with the complex xi (principal square root), valid_out = valid_in and
 *                Re(xi^2) > 0 (NumPy's order of complex numbers); dir_out is not written: the reference's ray
 *                direction behind such an interface is an arbitrary null vector of an SVD (see eps_im above), so
 *                a sequence ends there.  A mirror inside an absorbing isotropic medium: PRT_ERR_UNSUPPORTED.
 */
int32_t prt_interact_cplx(const prt_system_t *sys, int32_t surface, int64_t n, const double *x_hit,
                          const double *k_re, const double *k_im, const uint8_t *valid_in, double *k_out_re,
                          double *k_out_im, double *dir_out, double *e_out_re, double *e_out_im, uint8_t *valid_out,
                          void *stream);

/* Shape.getSag / getGrad: x, y (n) in the shape frame -> sag (n) and/or grad (3,n)
 * (either output may be NULL). */
int32_t prt_shape_eval(const prt_system_t *sys, int32_t surface, int64_t n, const double *x,
                       const double *y, double *sag, double *grad, void *stream);

/* A unit E field perpendicular to k, (3,n) -> (3,n), for bundles that leave an
 * isotropic interface (IsotropicMaterial.calc_e_field, material_isotropic.py:72-128,
 * returns an arbitrary unit vector of that 2-d space; not on the parity contract). */
int32_t prt_efield_perp(int32_t device, int64_t n, const double *k, double *e_out, void *stream);

/*
 * Device-side bundle generation: RectGrid.getGrid (sampling2d/raster.py:40-60, square raster
 * clipped to the unit disk, row-major order, reproduced bit-exactly) +
 * OpticalSystemAnalysis.collimated_bundle (analysis/optical_system_analysis.py:83-122) for an
 * isotropic background: origin = radius*p + start, k and E constant over the bundle.
 *   prt_rect_grid_count: number of samples per dimension and inside the disk for a requested
 *                        ray count (the raster returns "approximately nray" points).
 *   prt_collimated_bundle: writes rays [lo, hi) of that raster (a rank's shard, or 0..n_in_disk)
 *                        into (3, pitch) arrays x_out and (optionally) k_out, e_out.  k and E are one
 *                        vector for the whole bundle: a caller that traces with the uniform first
 *                        segment of prt_trace_ex passes k_out = e_out = NULL and stores nothing per ray.
 */
typedef struct prt_collimated {
    double radius, startx, starty, startz;
    double k[3]; /* n * unit vector (optical_system_analysis.py:110-118) */
    double e[3]; /* E field of every ray                                  */
} prt_collimated_t;
int32_t prt_rect_grid_count(int32_t device, int64_t nray, int64_t *n_per_dim, int64_t *n_in_disk,
                            void *stream);
int32_t prt_collimated_bundle(int32_t device, int64_t nray, int64_t lo, int64_t hi,
                              const prt_collimated_t *prm, int64_t pitch, double *x_out,
                              double *k_out, double *e_out, void *stream);

/*
 * The other deterministic pupil rasters (sampling2d/raster.py:62-164) and the divergent bundle
 * (OpticalSystemAnalysis.divergent_bundle, analysis/optical_system_analysis.py:124-165).  A raster is
 * given as an outer product of 1-d HOST tables: point (i, j), i = 0..ni-1 slow, j = 0..nj-1 fast
 * (the order of np.meshgrid(..).reshape / .flatten), is px = xa[j]*xb[i], py = ya[j]*yb[i]; with
 * `clip` only points with px*px + py*py <= 1 are kept (order kept).  The host computes the tables
 * exactly as the reference does, so the samples are the reference's bit for bit:
 *   RectGrid        xa = x1d, xb = 1, ya = 1, yb = x1d, clip              (raster.py:40-60)
 *   HexGrid         two such lattices (base and shifted), clipped, one after the other  (:62-92)
 *   MeridionalFan   ni = nray, nj = 1: xa = {-sin a}, xb = t, ya = {cos a}, yb = t      (:125-131)
 *   CircularGrid    xa = ya = radii, xb = cos(phi_i), yb = sin(phi_i)                    (:148-164)
 * prt_raster_bundle writes rays [lo, hi) of the (clipped) raster into (3, pitch) arrays:
 *   kind 0, collimated: origin = radius*p + start; k, e constant (prt_collimated_bundle's rule)
 *   kind 1, divergent:  origin = start; unit vector (sin(angley + radius*px) cos(anglex + radius*py),
 *           sin(anglex + radius*py), cos(angley + radius*px) cos(anglex + radius*py)), k = index * unit,
 *           E = the unit vector perpendicular to k that prt_efield_perp picks
 *   p_out (optional, (2, pitch)): the pupil samples themselves.  k_out may be NULL for kind 0 (see
 *   prt_collimated_bundle).
 */
typedef struct prt_raster {
    int64_t ni, nj;
    const double *xa, *xb, *ya, *yb; /* HOST arrays: xa, ya nj entries; xb, yb ni entries */
    int32_t clip, pad_;
} prt_raster_t;
typedef struct prt_bundle {
    int32_t kind, pad_;              /* 0 collimated, 1 divergent */
    double radius, start[3], anglex, angley;
    double index;                    /* divergent: refractive index of the background medium */
    double k[3], e[3];               /* collimated: wave vector and E field of every ray */
} prt_bundle_t;
int32_t prt_raster_count(int32_t device, const prt_raster_t *raster, int64_t *n_points, void *stream);
int32_t prt_raster_bundle(int32_t device, const prt_raster_t *raster, int64_t lo, int64_t hi,
                          const prt_bundle_t *prm, int64_t pitch, double *x_out, double *k_out,
                          double *e_out, double *p_out, void *stream);

/* RayBundle.returnKtoD (raytracer/ray.py:136-152) for one stored point: unit Poynting
 * directions d_out (3,n) from k (3,n) and E (e_re / e_im (3,n) or NULL; NULL e_re means E = ey
 * when use_default_e, else d = k/|k|).  Tight arrays. */
int32_t prt_poynting_dir(int32_t device, int64_t n, const double *k, const double *e_re,
                         const double *e_im, int32_t use_default_e, double *d_out, void *stream);

/* Per-ray sums over the n_points stored points of a bundle: mode 0 arc length
 * sum |x_{p+1}-x_p| (RayBundleAnalysis.get_arc_length, analysis/ray_analysis.py:136-147),
 * mode 1 phase difference sum (x_{p+1}.k_{p+1} - x_p.k_p) (get_phase_difference, :149-163).
 * xs / ks: HOST tables of n_points device pointers to tight (3,n) arrays; out (n) device. */
int32_t prt_path_sums(int32_t device, int32_t n_points, int64_t n, const double *const *xs,
                      const double *const *ks, int32_t mode, double *out, void *stream);

/*
 * Moments of a (3,n) point array (row pitch `pitch`, 0 = n) over the rays whose mask byte is
 * non-zero (mask NULL = all):  out7 (HOST) = { count, sum(v) [3], sum(v^2) [3] } with
 *   mode 0: v = x - ref          (ref: HOST, 3 doubles, NULL = origin)   centroid / RMS spot
 *   mode 1: v = x/|x|            unit directions of wave vectors          centroid direction
 *   mode 2: v = (x/|x|) x ref    cross product with a reference direction RMS angular size
 * Deterministic two-stage reduction; synchronises the stream.  Replaces the NumPy reductions of
 * RayBundleAnalysis.get_centroid_position / get_rms_spot_size / get_centroid_direction /
 * get_rms_angluar_size (raytracer/analysis/ray_analysis.py:44-134).
 */
int32_t prt_bundle_moments(int32_t device, int64_t n, int64_t pitch, const double *x,
                           const uint8_t *mask, int32_t mode, const double *ref, double *out7,
                           void *stream);

/* The same reduction without host round trips (for stream pipelines and multi-GPU all-reduce
 * chains): the result stays on the device (out7_dev), the reference point comes from the device
 * (ref_kind 0: origin, 1: ref_dev = 3 doubles, 2: ref_dev = a moments vector, e.g. the all-reduced
 * first pass -> reference = its centroid sum/(count + 1e-17)), scratch_dev holds at least
 * prt_moments_scratch_doubles(n) doubles.  Asynchronous. */
int64_t prt_moments_scratch_doubles(int64_t n);
int32_t prt_bundle_moments_async(int32_t device, int64_t n, int64_t pitch, const double *x,
                                 const uint8_t *mask, int32_t mode, const double *ref_dev,
                                 int32_t ref_kind, double *out7_dev, double *scratch_dev,
                                 void *stream);

/*
 * Order-preserving compaction by mask (the reference's [:, valid] indexing):
 * n_arrays (<= PRT_COMPACT_MAX_ROWS) row pointers of n doubles each (src[r] -> dst[r]; the pointer
 * tables themselves are HOST arrays of device pointers), plus an optional int64
 * id row and an optional uint8 row.  *n_kept (host) receives the survivor count.  Synchronises the
 * stream (the count is returned to the host).  scratch: device buffer of at
 * least prt_compact_scratch_bytes(n) bytes.  Runs on the device that owns `mask`
 * (all arrays and the stream must belong to it); the caller's current device is left as it was.
 */
int64_t prt_compact_scratch_bytes(int64_t n);
int32_t prt_compact(int64_t n, const uint8_t *mask, int32_t n_arrays, const double *const *src,
                    double *const *dst, const int64_t *id_src, int64_t *id_dst,
                    const uint8_t *u8_src, uint8_t *u8_dst, void *scratch, int64_t *n_kept,
                    void *stream);

/* Timing helper: runs prt_trace `iters` times on `stream` between two HIP events recorded on that
 * stream and returns the average milliseconds per launch in *ms_avg.  Used by bench.py for
 * roofline.kernel_ms and by callers that place their output arrays by measurement (the write bandwidth
 * of the path-mode march depends on where x_hit and k_out sit in HBM relative to each other --
 * DESIGN.md section 5 "Placement"). */
int32_t prt_trace_timed(const prt_system_t *sys, int64_t n0, int64_t in_pitch, const double *x0,
                        const double *k0, const double *e0_re, const double *e0_im, int32_t mode,
                        int64_t out_pitch, double *x_hit, double *k_out, uint8_t *valid,
                        uint8_t *valid_out, void *stream, int32_t iters, double *ms_avg);

/*
 * ---- placement-aware device memory for path arrays -------------------------------------------
 * Not a counterpart of a reference interface (the reference's arrays are NumPy's); it is the
 * allocator behind the arrays that OpticalSystem.seqtrace's replacement (prt_trace, PRT_MODE_PATH)
 * writes, and what the host layer uses for them by default.
 *
 * The physical HBM of an MI355X consists of three "kinds" of memory (each a third of the capacity,
 * in long physically contiguous runs; benchmarks/vmm_placement_probe.hip).  The
 * 72 write streams of the path-mode march run at 5.6-5.7 TB/s when x_hit and k_out lie in the same
 * kind and at 7.0 TB/s when they lie in two different ones; hipMalloc gives no control over that.
 * An arena takes physical memory in 1-GiB slabs (hipMemCreate), determines each slab's kind with a
 * short write probe against one representative slab per kind (about 2 ms per slab; the
 * representatives, at most PRT_ARENA_MAX_KINDS GiB, stay with the arena), and maps slabs of ONE kind
 * to contiguous virtual addresses for every buffer.
 *
 *   prt_arena_alloc   n_parts (<= 8) buffers of bytes[i] bytes (rounded up to whole slabs).  The first
 *                     n_distinct parts (-1: 2) are placed in pairwise DIFFERENT kinds (pass x_hit
 *                     and k_out there, and the input arrays as a third: reads that share a kind with
 *                     the write streams cost 3 % of the march); further parts get a kind no other part
 *                     of the call uses when slabs of one are at hand.  To find the kinds the arena
 *                     may take up to max_hunt_slabs extra slabs from the driver for the duration
 *                     of the call (-1: default = 32, and never more than half of the memory that is free at the
 *                     time of the call; about 1.5 ms per slab; a hunt also stops after 50 ms of probing.  A request
 *                     that does not find its kinds within that settles for fewer -- kinds[] tells -- and the next
 *                     request continues the hunt from the slabs this one left behind.  PRT_ARENA_HUNT=full lifts
 *                     the bounds to 256 slabs / 2 s / nine tenths of the free memory (and the arena's default budget
 *                     to nine tenths of the device): a kind is 96 GiB, so the third can be 192 slabs away;
 *                     PRT_ARENA_HUNT_SLABS / PRT_ARENA_HUNT_MS set them one by one; PRT_ARENA_TRACE prints every
 *                     probe's rate to stderr); if the device cannot offer
 *                     that many kinds the call still succeeds and kinds[] tells.  avoid_mask (bit q = kind q):
 *                     kinds this request leaves to others if it can -- a caller that allocates its
 *                     input arrays separately passes 3, which keeps them out of kinds 0 and 1, the
 *                     ones a two-part output request takes first.
 *                     ptrs[i] are ordinary device pointers, 2-MiB aligned.  The probe runs on
 *                     `stream`; the call synchronises it.  A call that has to MAP memory (a buffer size not in
 *                     the cache) waits for the whole device first and touches the new mapping with a kernel
 *                     before it returns (ABI v6, round 5: mappings are made and unmade on an idle device, like
 *                     hipMalloc / hipFree do it; PRT_ARENA_SYNC_MAPS=0 switches that off); a call served from
 *                     the cache does neither.
 *   prt_arena_free    returns a buffer (pointer as given by prt_arena_alloc).  Does not block: an event
 *                     recorded on `stream` -- the stream of the last work that uses the buffer --
 *                     marks the release; the next prt_arena_alloc that hands the buffer out makes
 *                     its stream wait for that event.  The buffer stays mapped and serves the next
 *                     request of the same size and kind without any driver call.
 *   prt_arena_trim    hands all cached (unused) memory back to the driver.
 *   prt_arena_set_budget  caps the physical memory the arena may hold at any time, in 1-GiB slabs (in use +
 *                     cached + free + the representatives; negative = no cap; default: three quarters of the
 *                     device's memory, PRT_ARENA_BUDGET_GIB).  Independently, cached (unused) buffers beyond
 *                     64 GiB (PRT_ARENA_CACHE_GIB) are handed back to the driver when a buffer is freed.  At the cap a
 *                     request behaves as if the driver had nothing left: cached buffers of other sizes are
 *                     taken apart, fewer kinds are accepted, or PRT_ERR_NOMEM.
 *   prt_arena_kind_of kind index of a pointer inside one of the arena's buffers.
 *   prt_arena_stats   out[0..12): kinds seen, probes run, slabs created, slabs released, free slabs,
 *                     slabs in use, slabs cached, slab size in bytes, slabs per kind (4 entries);
 *                     rates[0..8): last same-kind probe rate, last cross-kind probe rate (GB/s),
 *                     total probe time (ms), bytes of address space handed out so far, base of the first address
 *                     window, number of windows, 1 if every window landed at its hinted address, window size.
 * Address space: the arena maps every virtual address at most once and never returns a range to the
 * runtime -- with ROCm 7.0 / 7.2 a range that is unmapped and mapped again keeps translating to the old
 * physical pages (csrc/prt_placed.h).  Only address space leaks (1 GiB per slab tested), not memory.
 * All of its addresses come from windows it reserves for itself at 0x2000'0000'0000 + 8 TiB x device (ABI v7; 1 TiB
 * each, PRT_ARENA_VA_WINDOW_GIB / PRT_ARENA_VA_BASE): tens of TiB away from the heap and from the mmap area, so an
 * arena mapping can never land on addresses the host allocator has just given back -- e.g. the destination of a
 * pageable device-to-host copy, which the runtime keeps registered with the GPU driver for a while
 * (benchmarks/va_reuse_probe.hip; until ABI v6 half of the reservations behind such a copy covered its range).
 * A prt_arena_alloc that maps memory synchronises the whole device: never call it during stream capture.
 * Thread-safe (one lock per arena).
 */
#define PRT_ARENA_MAX_KINDS 4
typedef struct prt_arena prt_arena_t;
int32_t prt_arena_create(int32_t device, prt_arena_t **out);
int32_t prt_arena_destroy(prt_arena_t *arena);
int32_t prt_arena_alloc(prt_arena_t *arena, int32_t n_parts, const int64_t *bytes, void **ptrs,
                        int32_t *kinds, int32_t n_distinct, int32_t avoid_mask, int32_t max_hunt_slabs,
                        void *stream);
int32_t prt_arena_free(prt_arena_t *arena, void *ptr, void *stream);
int32_t prt_arena_trim(prt_arena_t *arena);
int32_t prt_arena_set_budget(prt_arena_t *arena, int64_t max_live_slabs);
int32_t prt_arena_kind_of(prt_arena_t *arena, const void *ptr, int32_t *kind);
int32_t prt_arena_stats(prt_arena_t *arena, int64_t *out, int32_t n_out, double *rates, int32_t n_rates);
/* "compute partition/memory partition[; note]": the partition modes the arena found in sysfs (amdgpu's
 * current_compute_partition / current_memory_partition; "unknown" where the files are missing), and -- outside
 * SPX / NPS1, where the three kinds of HBM were characterised -- the note that slabs are not classified.  The string
 * lives in a per-thread buffer of the library: valid until the calling thread's next prt_arena_note call. */
const char *prt_arena_note(prt_arena_t *arena);

#ifdef __cplusplus
}
#endif
#endif /* PRT_H */
