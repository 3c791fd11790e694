"""
Helpers shared by the parity tests: load a golden fixture (tests/golden/*.npz,
written by oracle/make_golden.py from the real reference) and compare a DENSE
trace (oracle or HIP engine) against the reference's compacted RayBundles.

The reference removes invalid rays after every isotropic refraction
(material_isotropic.py:194-199) and duplicates all rays at every anisotropic
interface (material_anisotropic.py:87-100); ``RefWalker`` replays exactly that
bookkeeping on index arrays, so every ray of every reference bundle is mapped to
its slot in the dense arrays ("join on rayID", done positionally so that split
rays with equal rayID stay distinguishable).
"""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ISO_CASES = ["doublet", "doublet_clipped", "double_gauss_axis", "double_gauss_field5",
             "double_gauss_wide", "double_gauss_Fline", "double_gauss_defaultE",
             "tilted_frames", "mirrors", "two_elements", "catalog_doublet",
             # demos/demo_mirrors.py: three tilted spherical mirrors + an off-axis paraboloid far from its vertex
             "tma_paraboloid_field0p5",
             "spd_double_gauss_Fline", "prism_red", "prism_blue",
             # the reference's own benchmark workload (demos/demo_benchmark.py:47-78): divergent bundle, per-ray k0 / E0
             "benchmark_divergent"]
EXPLICIT_CASES = ["asphere_mild_axis", "asphere_mild_field5", "asphere_strong_axis",
                  "asphere_strong_field5", "xypoly_axis", "xypoly_field5", "xypoly_bench_field5", "biconic_axis",
                  "biconic_field5", "hud_biconic_mirrors",
                  # the patent HUD prism of demos/demo_hud.py: biconic faces with large b_n, hit points far out
                  "hud_patent_axis", "hud_patent_field-15",
                  "zmx_lenssystem",
                  "zernike_fringe_field3", "zernike_ansi_field2", "zernike_combination_mirror",
                  # m = 0 Zernike terms only: the reference's normals are right there, the whole path is compared
                  "zernike_fringe_symmetric_field3",
                  # LinearCombination whose polynomial part is decentred and rotated about the surface's axis
                  "rotated_combination_lens",
                  "gridsag_field2"]
ANISO_CASES = ["aniso_doublet_isoeps", "aniso_doublet_uniaxial", "aniso_doublet_biaxial",
               "aniso_doublet_uniaxial_clipped", "aniso_doublet_uniaxial_stopped",
               "aniso_mirror_uniaxial", "aniso_mirror_biaxial"]
ALL_CASES = ISO_CASES + EXPLICIT_CASES + ANISO_CASES
# every explicit-shape case once more with the REFERENCE converged (oracle/make_golden.py: annotations["tol"] = 1e-14 on
# every explicit shape instead of the default xtol = 1e-6, surface_shape.py:396, 457-458): compared with the flat
# 1e-10 of north_star, no allowance; the fixture carries the reference's own residual |z - F(x, y)| per surface
# cases that exist ONLY with the reference converged: a near-hemisphere hit at up to 84.4 degrees of incidence (round 6:
# Newton's g' = d . grad is 0.1 there) -- with its default xtol the reference's fsolve gives up on this bundle
# ("not making good progress"), so there is no out-of-the-box result to compare with
EXPLICIT_TIGHT_ONLY = ["asphere_grazing_field30"]
EXPLICIT_TIGHT_CASES = [c + "_tight" for c in EXPLICIT_CASES + EXPLICIT_TIGHT_ONLY]
# What the engine's RAW deviation from the converged reference is asserted to stay below on a tight fixture (the bar
# itself is 1e-10, flat).  The 19 twins measure <= 2.6e-15.  At grazing incidence the wave vectors carry the error of the
# Newton loop's gradient, which is moved to the root along the secant of the last two evaluations instead of being
# evaluated there (prt_device.h explicit_t): 1/2 g'' dt (dt + dt_prev) -- with the 1e-8 exit up to ~1e-12 where the
# surface's gradient bends fastest along the ray (the host build of the kernels measures 1.6e-12 on the dome).
TIGHT_RAW_CAP = {"asphere_grazing_field30_tight": 2e-11}
TIGHT_RAW_CAP_DEFAULT = 1e-12
REF_RESIDUAL_MAX = 1e-12        # [mm] what "the reference is converged on every ray" means for a tight fixture
# complex (absorbing) epsilon tensors, sequences that stay inside crystals: complex wave vectors, compared as such
ABSORBING_CASES = ["aniso_absorbing_mirror", "aniso_absorbing_two_crystals",
                   # ... and sequences that END in an isotropic medium (complex k behind the last surface only):
                   "aniso_absorbing_exit", "absorbing_detector"]

# caps of the first-order allowance compare_dense_to_reference grants behind an explicit surface (wave-vector units /
# millimetres): what the reference's fsolve error (xtol = 1e-6 of t) can amount to after a few surfaces
# (largest allowances actually granted, by zmx_lenssystem's thirteen fsolve surfaces in a row: 4.8e-5 mm / 3.4e-7)
ALLOWANCE_CAP_K = 2e-6
ALLOWANCE_CAP_X = 2e-4
# ... and whatever the allowance, the RAW deviation from the loose reference (no allowance subtracted) stays below this
RAW_CAP = 1e-9

# The reference's Zernike gradient (surface_shape.py:1073-1084) is not the derivative of its own sag
# for terms with m != 0 (angular part divided by rho instead of rho**2; pinned by
# test_zernike_and_combination_shapes_equal_reference), so its surface normals -- and everything
# behind the refraction / reflection at such a surface -- are not a parity target.  Value: index of
# the first surface of the case with such a shape; the comparison against the reference covers the
# hit points up to and including that surface, the rest is covered by HIP-vs-oracle.
REFERENCE_NORMAL_DEFECT = {"zernike_fringe_field3": 2, "zernike_ansi_field2": 2,
                           "zernike_combination_mirror": 0}


class Case(object):
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.x0 = z["x0"]
        self.k0 = z["k0"]
        self.E0 = z["E0"]
        self.wave = float(z["wave"])
        self.table = json.loads(str(z["table_json"]))
        self.npaths = int(z["npaths"])
        self.paths = []
        for pi in range(self.npaths if ("p1_nb" in z.files) else 1):
            pre = "" if pi == 0 else "p%d_" % pi
            nb = int(z[pre + "nb"])
            self.paths.append([dict(x=z[pre + "b%d_x" % i], k=z[pre + "b%d_k" % i],
                                    valid=z[pre + "b%d_valid" % i], id=z[pre + "b%d_id" % i])
                               for i in range(nb)])
        self.elem_lengths = [int(v) for v in z["elem_lengths"]]
        self.ref_resid = {int(k[len("ref_resid_s"):]): z[k] for k in z.files if k.startswith("ref_resid_s")}
        self.raw_bundles = self.paths[0]
        # canonical list [b0, b0, b1, ..., bS]: drop the duplicates the reference inserts at
        # every further element boundary (optical_element.py:330 + ray.py:218-219)
        self.paths = [self._drop_boundary_duplicates(p) for p in self.paths]
        self.bundles = self.paths[0]

    def _drop_boundary_duplicates(self, bundles):
        keep = []
        pos = 0
        for (e, L) in enumerate(self.elem_lengths):
            # layout per element: [duplicate of current bundle] + L new bundles
            if e == 0:
                keep += [pos, pos + 1]
            pos += 1
            keep += list(range(pos + 1, pos + 1 + L))
            pos += L
        assert pos + 1 == len(bundles), (pos, len(bundles))
        return [bundles[i] for i in keep]

    @property
    def n_surfaces(self):
        return len(self.table)


def load_case(name):
    return Case(name)


def relative_scale(x):
    """per-ray |x| for the relative error, floored at 1% of the largest |x| on the surface
    (a chief ray through a vertex at the origin has |x| = 0)"""
    nrm = np.sqrt(np.sum(x ** 2, axis=0))
    return np.maximum(nrm, max(1e-2 * float(np.max(nrm, initial=0.0)), 1e-30))


def _subsequence_positions(old_x, old_id, new_x, new_id):
    """positions i_0 < i_1 < ... in the old bundle of the rays kept in the new one
    (the new bundle is old[:, mask]; hit points are copied bit-exactly)."""
    pos = np.empty(new_x.shape[1], dtype=np.int64)
    i = 0
    n_old = old_x.shape[1]
    for j in range(new_x.shape[1]):
        while True:
            assert i < n_old, "reference bundle is not a filtered copy of its predecessor"
            same = (old_id[i] == new_id[j]) and np.array_equal(old_x[:, i], new_x[:, j], equal_nan=True)
            i += 1
            if same:
                pos[j] = i - 1
                break
    return pos


def compare_dense_to_reference(case, dense, rtol_x=1e-10, atol_k=1e-10, explicit_tol=None,
                               check_valid=True):
    """
    dense: list over surfaces of dicts with numpy arrays
        x_hit (3, n_in), valid (n_in,), k_out (3, n_out), valid_out (n_out,)
    Returns dict(max_rel_x, max_abs_k, n_compared).  Raises AssertionError on a
    validity / bookkeeping mismatch or when a tolerance is exceeded.

    explicit_tol: optional callable (surface_index, bundle, record) -> per-ray absolute tolerance [mm] on the hit
    points of a surface whose REFERENCE hit points are not converged (fsolve stops at xtol 1e-6, SURVEY.md
    headline 4), None for closed-form surfaces.  What such a position error does further down is followed to first
    order, per ray, instead of re-using millimetres for everything:
      position allowance  dx_s = [explicit_tol at s] + dx_(s-1) + L_s * dk_(s-1) / |k|    (L_s: path length to s)
      direction allowance dk_s = (n_before + n_after) * kappa_s * dx_s + 2 * dk_(s-1)
    kappa_s = Frobenius norm of the Hessian of the sag at the hit point [1/mm] (``surface_curvature``): a hit
    point displaced by dx sees a normal turned by at most kappa dx, and a refraction / reflection moves k by at
    most (n_before + n_after) times that angle; an incoming direction error is at most doubled (mirror).
    """
    b = case.bundles
    S = case.n_surfaces
    assert len(b) == S + 2, "unexpected number of reference bundles"
    # b[0] initial, b[1] the same bundle object again (optical_element.py:330), then one per surface
    pos = np.array(b[1]["id"], dtype=np.int64)      # reference slot -> dense slot (initial: rayID)
    max_rel_x = 0.0
    max_abs_k = 0.0
    ncmp = 0
    extra_x = None          # per reference-slot position allowance [mm] behind unconverged reference hit points
    extra_k = None          # per reference-slot direction allowance (dimensionless, |k| ~ n)
    n_before = 1.0
    last_trusted = REFERENCE_NORMAL_DEFECT.get(case.name[:-len("_tight")] if case.name.endswith("_tight") else case.name)
    (raw_rel_x, raw_abs_k, max_extra_x, max_extra_k) = (0.0, 0.0, 0.0, 0.0)
    for s in range(S):
        B = b[s + 1]
        d = dense[s]
        n_dense_in = d["x_hit"].shape[1]
        xr = B["x"][-1]
        vr = B["valid"][-1].astype(bool)
        xd = d["x_hit"][:, pos]
        vd = d["valid"][pos].astype(bool)
        if explicit_tol is not None:
            tol_s = explicit_tol(s, B, case.table[s])
            if extra_x is not None:
                # the inherited direction error displaces the hit point by its angle times the path length
                L = np.sqrt(np.sum((B["x"][-1] - B["x"][0]) ** 2, axis=0))
                extra_x = extra_x + np.where(np.isfinite(L), L, 0.0) * extra_k / n_before
            if tol_s is not None:
                extra_x = tol_s if extra_x is None else extra_x + tol_s
            if extra_x is not None:
                mat = case.table[s]["material"]
                n_after = mat["n"] if mat["type"] == "isotropic" else float(np.sqrt(np.max(np.abs(mat["eps_re"]))))
                kappa = surface_curvature(case.table[s], B["x"][-1])
                dk_new = (n_before + n_after) * np.where(np.isfinite(kappa), kappa, 0.0) * extra_x
                # an inherited direction error passes a refraction amplified by at most n_after / n_before (Snell: the
                # tangential part of k is kept, the normal part changes by less), a mirror by at most 2
                gain = 2.0 if case.table[s]["interaction"] == "mirror" else max(1.0, n_after / n_before)
                extra_k = dk_new if extra_k is None else dk_new + gain * extra_k
                # the allowance exists for the reference's own xtol = 1e-6 convergence error: it must stay far below
                # anything a real defect would produce -- a runaway allowance fails the test instead of hiding one
                assert float(np.max(extra_k)) < ALLOWANCE_CAP_K and float(np.max(extra_x)) < ALLOWANCE_CAP_X, \
                    "%s surface %d: tolerance allowance ran away (k %.2e, x %.2e)" % (case.name, s, np.max(extra_k), np.max(extra_x))
        if case.table[s]["interaction"] != "mirror":
            mat = case.table[s]["material"]
            n_next = mat["n"] if mat["type"] == "isotropic" else float(np.sqrt(np.max(np.abs(mat["eps_re"]))))
        else:
            n_next = n_before
        if check_valid:
            assert np.array_equal(vd, vr), \
                "%s surface %d: valid mask differs for %d rays" % (case.name, s, np.sum(vd != vr))
        cmp_mask = vr & vd
        if np.any(cmp_mask):
            scale = relative_scale(xr[:, cmp_mask])
            err = np.abs(xd[:, cmp_mask] - xr[:, cmp_mask])
            raw_rel_x = max(raw_rel_x, float(np.max(err / scale)))
            if extra_x is not None:
                max_extra_x = max(max_extra_x, float(np.max(extra_x[cmp_mask])))
                err = np.maximum(err - extra_x[cmp_mask], 0.0)
            rel = np.max(err / scale)
            max_rel_x = max(max_rel_x, float(rel))
            ncmp += int(np.sum(cmp_mask))
        if last_trusted is not None and s == last_trusted:
            break
        # ---- the bundle created by refract / reflect at this surface
        Bn = b[s + 2]
        aniso = case.table[s]["material"]["type"] == "anisotropic"
        if aniso:
            n_ref = B["x"].shape[2]
            assert Bn["x"].shape[2] == 2 * n_ref
            new_pos = np.concatenate((pos, pos + n_dense_in))
            if extra_x is not None:
                extra_x = np.concatenate((extra_x, extra_x))
                extra_k = np.concatenate((extra_k, extra_k))
        else:
            sub = _subsequence_positions(B["x"][-1], B["id"], Bn["x"][0], Bn["id"])
            new_pos = pos[sub]
            if check_valid:
                kept_dense = d["valid_out"][pos].astype(bool)
                kept_ref = np.zeros(len(pos), dtype=bool)
                kept_ref[sub] = True
                assert np.array_equal(kept_dense, kept_ref), \
                    "%s surface %d: refraction validity differs" % (case.name, s)
            if extra_x is not None:
                extra_x = extra_x[sub]
                extra_k = extra_k[sub]
        n_before = n_next
        kr = Bn["k"][0]
        kd = d["k_out"][:, new_pos]
        if np.iscomplexobj(kd):
            kr = np.asarray(kr, dtype=complex)       # absorbing crystals: the wave vectors ARE complex, compared as such
        else:
            assert np.max(np.abs(np.imag(kr))) < 1e-9 if np.iscomplexobj(kr) else True
            kr = np.real(kr)
        fin = np.all(np.isfinite(kr), axis=0)
        if np.any(fin):
            errk = np.abs(kd[:, fin] - kr[:, fin])
            raw_abs_k = max(raw_abs_k, float(np.max(errk)))
            if extra_k is not None:
                max_extra_k = max(max_extra_k, float(np.max(extra_k[fin])))
                # direction errors inherited from unconverged reference hit points (allowance derived above)
                errk = np.maximum(errk - extra_k[fin], 0.0)
            max_abs_k = max(max_abs_k, float(np.max(errk)))
        pos = new_pos
    assert max_rel_x <= rtol_x, "%s: hit points differ by %.3e relative" % (case.name, max_rel_x)
    assert max_abs_k <= atol_k, "%s: wave vectors differ by %.3e" % (case.name, max_abs_k)
    return dict(max_rel_x=max_rel_x, max_abs_k=max_abs_k, n_compared=ncmp, raw_rel_x=raw_rel_x, raw_abs_k=raw_abs_k,
                max_allowance_x=max_extra_x, max_allowance_k=max_extra_k)


def surface_curvature(rec, x_glob, h=1e-4):
    """Frobenius norm of the Hessian of the sag z = F(x, y) at the given global points (central differences of
    the oracle's gradient of z - F, shape frame) [1/mm]: bounds how fast the surface normal turns per millimetre"""
    from oracle import seqtrace_np as oracle
    p = oracle.g2l_points(np.asarray(rec["B_shape"]), np.asarray(rec["g_shape"]), x_glob)
    sh = rec["shape"]
    with np.errstate(all="ignore"):
        gxp = oracle.shape_grad(sh, p[0] + h, p[1])
        gxm = oracle.shape_grad(sh, p[0] - h, p[1])
        gyp = oracle.shape_grad(sh, p[0], p[1] + h)
        gym = oracle.shape_grad(sh, p[0], p[1] - h)
        hxx = (gxp[0] - gxm[0]) / (2 * h)
        hxy = (gxp[1] - gxm[1]) / (2 * h)
        hyx = (gyp[0] - gym[0]) / (2 * h)
        hyy = (gyp[1] - gym[1]) / (2 * h)
        return np.sqrt(hxx ** 2 + hxy ** 2 + hyx ** 2 + hyy ** 2)


def dense_from_oracle(out, complex_k=False):
    return [dict(x_hit=o["x_hit"], valid=o["valid"], k_out=(np.asarray(o["k_out"], dtype=complex) if complex_k
                                                             else np.real(o["k_out"])),
                 valid_out=o["valid_out"]) for o in out]


def dense_from_engine(res, complex_k=False):
    """TraceResult (device tensors) -> list of numpy dicts."""
    dense = []
    for s in range(len(res.x_hit)):
        k = res.k_out[s].cpu().numpy()
        if complex_k:
            k = k + 1j * res.k_out_im[s].cpu().numpy()
        dense.append(dict(x_hit=res.x_hit[s].cpu().numpy(), valid=res.valid[s].cpu().numpy(),
                          k_out=k, valid_out=res.valid_out[s].cpu().numpy()))
    return dense
