"""The sources of libprt compiled for the HOST (tests/hostemu: a stand-in HIP runtime, the threads of a block as fibres)
against the reference's golden vectors and the oracle -- CPU only, no GPU needed.

This is a check of the kernels' SOURCE: the same C++ expressions, the same launch-site logic of prt.hip (which
instantiation, which layout, which grid), every index computation under AddressSanitizer / UBSan.  It is not the product
(pyrate_amd loads the gfx950 build or raises) and not a measurement; the `-m gpu` suite runs the same comparisons on the
device build.  Tolerances are the `-m gpu` suite's (BASELINE.json: 1e-10 relative on hit points and direction cosines).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _golden
from oracle import seqtrace_np as oracle
from test_oracle_golden import explicit_tolerance

hostemu = pytest.importorskip("hostemu")



@pytest.fixture(scope="module", autouse=True)
def _host_library():
    """built on first use (not at import: `pytest -m gpu` on the GPU box collects this module and deselects all of it)"""
    try:
        hostemu.build_all()
    except Exception as exc:                                    # no clang++ on this box
        pytest.skip("no host build of libprt: %s" % exc)


def host_trace(case, **kw):
    hs = hostemu.HostSystem(case.table)
    e = np.asarray(case.E0)
    return hs, hs.trace(case.x0, np.real(case.k0), e.real, e.imag if np.iscomplexobj(e) else None, **kw)


@pytest.mark.parametrize("name", _golden.ISO_CASES)
def test_host_build_vs_reference_isotropic(name):
    case = _golden.load_case(name)
    (hs, dense) = host_trace(case)
    r = _golden.compare_dense_to_reference(case, dense, rtol_x=1e-10, atol_k=1e-10)
    assert r["n_compared"] > 0 and r["max_rel_x"] < 1e-13 and r["max_abs_k"] < 1e-13
    assert hs.padding_untouched            # nothing written beyond ray n0 - 1 of a pitched row


@pytest.mark.parametrize("name", _golden.EXPLICIT_CASES)
def test_host_build_vs_reference_explicit(name):
    case = _golden.load_case(name)
    (_, dense) = host_trace(case)
    r = _golden.compare_dense_to_reference(case, dense, rtol_x=1e-10, atol_k=1e-10, explicit_tol=explicit_tolerance)
    assert r["raw_rel_x"] <= _golden.RAW_CAP and r["raw_abs_k"] <= _golden.RAW_CAP, r
    for (s, rec) in enumerate(case.table):
        if rec["shape"]["type"] == "conic":
            continue
        p = oracle.g2l_points(np.asarray(rec["B_shape"]), np.asarray(rec["g_shape"]), dense[s]["x_hit"])
        assert np.nanmax(np.abs(p[2] - oracle.shape_sag(rec["shape"], p[0], p[1]))) < 1e-13


@pytest.mark.parametrize("name", _golden.EXPLICIT_TIGHT_CASES)
def test_host_build_vs_converged_reference_explicit(name):
    """the tight twins (reference converged to 1e-14) incl. the grazing-incidence case of round 6: flat 1e-10, no
    allowance -- the Newton stop rule (1e-8 exit behind an observed contraction, prt_device.h explicit_t) as compiled"""
    case = _golden.load_case(name)
    (_, dense) = host_trace(case, want_nonconv=True)
    r = _golden.compare_dense_to_reference(case, dense, rtol_x=1e-10, atol_k=1e-10, explicit_tol=None)
    assert r["n_compared"] > 0 and r["max_allowance_x"] == 0.0 and r["max_allowance_k"] == 0.0
    cap = _golden.TIGHT_RAW_CAP.get(name, _golden.TIGHT_RAW_CAP_DEFAULT)
    assert r["raw_rel_x"] < cap and r["raw_abs_k"] < cap, r
    for d in dense:
        assert not np.any(d["nonconv"][d["valid"].astype(bool)])


@pytest.mark.parametrize("name", _golden.ANISO_CASES)
def test_host_build_vs_reference_anisotropic(name):
    case = _golden.load_case(name)
    (_, dense) = host_trace(case)
    _golden.compare_dense_to_reference(case, dense, rtol_x=1e-10, atol_k=1e-10)


@pytest.mark.parametrize("name", _golden.ABSORBING_CASES)
def test_host_build_vs_reference_absorbing(name):
    case = _golden.load_case(name)
    (hs, dense) = host_trace(case, want_fields=True)
    assert hs.complex_eps
    out = _golden.compare_dense_to_reference(case, dense, rtol_x=1e-10, atol_k=1e-10)
    assert out["n_compared"] > 0 and out["max_abs_k"] < 1e-12


@pytest.mark.parametrize("name", _golden.ISO_CASES + _golden.EXPLICIT_CASES + _golden.ANISO_CASES)
def test_host_build_vs_oracle_dense(name):
    """every ray, valid or not: masks identical, values within 1e-11 where both are finite and valid"""
    case = _golden.load_case(name)
    (_, dense) = host_trace(case)
    out = oracle.trace(case.table, case.x0, case.k0, case.E0)
    for s in range(case.n_surfaces):
        assert np.array_equal(dense[s]["valid"].astype(bool), out[s]["valid"]), (name, s)
        assert np.array_equal(dense[s]["valid_out"].astype(bool), out[s]["valid_out"]), (name, s)
        v = out[s]["valid"]
        xo = out[s]["x_hit"][:, v]
        assert np.max(np.abs(dense[s]["x_hit"][:, v] - xo) / _golden.relative_scale(xo), initial=0.0) < 1e-11
        vo = out[s]["valid_out"] & np.all(np.isfinite(np.real(out[s]["k_out"])), axis=0)
        assert np.max(np.abs(dense[s]["k_out"][:, vo] - np.real(out[s]["k_out"])[:, vo]), initial=0.0) < 1e-11


@pytest.mark.parametrize("name", ["double_gauss_wide", "tilted_frames", "asphere_strong_field5", "aniso_doublet_uniaxial",
                                  "aniso_doublet_biaxial"])
def test_host_build_layouts_and_modes_agree(name):
    """image mode, the flags byte, tight / recommended / odd output pitches, pitched inputs: the same records"""
    from pyrate_amd import _lib as P
    case = _golden.load_case(name)
    (hs, ref) = host_trace(case)
    n0 = case.x0.shape[1]

    def same(a, b, tol=1e-13):
        assert np.array_equal(a["valid"], b["valid"]) and np.array_equal(a["valid_out"], b["valid_out"])
        m = a["valid_out"].astype(bool)
        assert np.allclose(a["x_hit"][:, a["valid"].astype(bool)], b["x_hit"][:, a["valid"].astype(bool)], rtol=tol, atol=tol)
        assert np.allclose(a["k_out"][:, m], b["k_out"][:, m], rtol=0, atol=tol)
    (_, img) = host_trace(case, mode=P.MODE_IMAGE)
    assert len(img) == 1
    same(img[0], ref[-1])
    variants = [dict(pitch=0), dict(in_pitch=n0 + 6)]
    if hs.all_isotropic:
        # (misalign: every array starts 8 B behind a 16-B boundary, mask rows at odd addresses -- even pitches or not, the
        #  library must take its 8-B / 1-B access forms; UBSan's alignment check watches in the sanitizer run)
        variants += [dict(pitch=n0 + 1), dict(pitch=n0 + 2, in_pitch=n0 + 3), dict(flags=True),
                     dict(misalign=True), dict(misalign=True, pitch=n0 + 2 + n0 % 2, in_pitch=n0 + 4 + n0 % 2, want_nonconv=True)]
    else:
        variants += [dict(pitch=n0 + 3)]
    for kw in variants:
        (_, got) = host_trace(case, **kw)
        for s in range(case.n_surfaces):
            same(got[s], ref[s])


def _walk_surface_by_surface(case, pitch, n=None, misalign=False):
    hs = hostemu.HostSystem(case.table)
    e = np.asarray(case.E0)
    sl = slice(0, n)
    (x, k) = (case.x0[:, sl], np.real(case.k0)[:, sl])
    (e_re, e_im) = (e.real[:, sl], e.imag[:, sl] if np.iscomplexobj(e) else None)
    out = oracle.trace(case.table, case.x0[:, sl], case.k0[:, sl], case.E0[:, sl])
    valid = None
    (xs, ks, valid_s) = (x, k, None)
    for s in range(case.n_surfaces):
        first = dict(e_re=e_re, e_im=e_im) if s == 0 else dict(default_e=False)
        # the two calls (Material.propagate, Material.refract) ...
        (xh, v, nc) = hs.propagate_rows(s, x, k, valid_in=valid, pitch=pitch, want_nonconv=True, misalign=misalign, **first)
        (k2, vo) = hs.interact_rows(s, xh, k, valid_in=v, pitch=pitch, misalign=misalign)
        # ... and the fused step of the same surface
        (xh_f, k2_f, v_f, vo_f, nc_f) = hs.surface_step_rows(s, xs, ks, valid_in=valid_s, pitch=pitch, want_nonconv=True,
                                                             misalign=misalign, **first)
        for (got_x, got_k, got_v, got_vo) in ((xh, k2, v, vo), (xh_f, k2_f, v_f, vo_f)):
            assert np.array_equal(got_v.astype(bool), out[s]["valid"]), (s, pitch)
            assert np.array_equal(got_vo.astype(bool), out[s]["valid_out"]), (s, pitch)
            m = out[s]["valid_out"]
            xo = out[s]["x_hit"][:, m]
            assert np.max(np.abs(got_x[:, m] - xo) / _golden.relative_scale(xo), initial=0.0) < 1e-11
            assert np.max(np.abs(got_k[:, m] - np.real(out[s]["k_out"])[:, m]), initial=0.0) < 1e-11
        assert np.array_equal(nc, nc_f)
        (x, k, valid) = (xh, k2, vo)
        (xs, ks, valid_s) = (xh_f, k2_f, vo_f)


@pytest.mark.parametrize("name", ["double_gauss_wide", "tilted_frames", "xypoly_field5", "mirrors", "asphere_strong_field5",
                                  "biconic_field5", "doublet_clipped", "asphere_grazing_field30_tight"])
def test_host_build_per_surface_calls_and_the_fused_surface_step(name):
    """prt_propagate_rows + prt_interact_rows (Material.propagate / refract, two calls per surface) and
    prt_surface_step_rows (both in one launch; never on a GPU before round 6's last session) against the oracle, surface
    by surface -- with the aligned two-rays-per-thread form (even pitch) and the 8-B fall-back (tight arrays of odd
    length), and odd ray counts in both"""
    case = _golden.load_case(name)
    n0 = case.x0.shape[1]
    for (pitch, n) in ((n0 + (n0 % 2) + 2, n0), (None, n0), (n0 + 1 - (n0 % 2), n0 - 1), (None, n0 - 1 if (n0 - 1) % 2 else n0 - 2),
                       (4, 3), (None, 1)):
        _walk_surface_by_surface(case, pitch, n)
    _walk_surface_by_surface(case, n0 + (n0 % 2) + 2, n0, misalign=True)       # even pitch, bases 8 B off a 16-B boundary


def test_host_build_refuses_crystals_in_the_row_calls():
    case = _golden.load_case("aniso_doublet_uniaxial")
    hs = hostemu.HostSystem(case.table)
    s_c = [s for (s, r) in enumerate(case.table) if r["material"]["type"] == "anisotropic"][0]
    (x, k) = (case.x0, np.real(case.k0))
    with pytest.raises(hostemu.HostemuError) as ei:
        hs.surface_step_rows(s_c, x, k)
    assert ei.value.code == -2
    with pytest.raises(hostemu.HostemuError):
        hs.interact_rows(s_c, x, k)


def test_host_build_many_blocks_and_the_xcd_block_map():
    """bundles of several hundred blocks with ray counts that are not multiples of anything: the XCD-contiguous block map
    of the march (grid a multiple of 8, partial rows by ray block) visits every ray exactly once"""
    from pyrate_amd import systems
    recs = systems.double_gauss_records()
    for n in (2 * 128 * 8 * 3 + 1, 2 * 128 * 19 - 3, 5000):
        (o, k, e0) = systems.double_gauss_bundle(n, field_deg=4.0)
        n = o.shape[1]
        hs = hostemu.HostSystem(recs)
        dense = hs.trace(o, k, e0)
        out = oracle.trace(recs, o, k, e0)
        assert hs.padding_untouched
        for s in range(len(recs)):
            assert np.array_equal(dense[s]["valid_out"].astype(bool), out[s]["valid_out"])
            m = out[s]["valid_out"]
            assert np.max(np.abs(dense[s]["x_hit"][:, m] - out[s]["x_hit"][:, m]), initial=0.0) < 1e-11
        # the uniform first segment (collimated bundle: only x0 is read) gives the array form's records
        if np.ptp(k, axis=1).max() == 0.0:
            uni = hs.trace(o, uniform=(k[:, 0], e0[:, 0], P_FIRST_E_UNIFORM()))
            for s in range(len(recs)):
                assert np.array_equal(uni[s]["valid_out"], dense[s]["valid_out"])
                m = dense[s]["valid_out"].astype(bool)
                assert np.array_equal(uni[s]["x_hit"][:, m], dense[s]["x_hit"][:, m])


def P_FIRST_E_UNIFORM():
    from pyrate_amd import _lib as P
    return P.FIRST_E_UNIFORM


SANITIZER_SCRIPT = r'''
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import _golden, hostemu
from pyrate_amd import _lib as P
from oracle import seqtrace_np as oracle
import test_hostemu as T
worst = 0.0
for name in %(cases)r:
    case = _golden.load_case(name)
    (hs, dense) = T.host_trace(case, want_nonconv=True)
    out = oracle.trace(case.table, case.x0, case.k0, case.E0)
    for s in range(case.n_surfaces):
        assert np.array_equal(dense[s]["valid_out"].astype(bool), out[s]["valid_out"]), (name, s)
        m = out[s]["valid_out"] & np.all(np.isfinite(np.real(out[s]["k_out"])), axis=0)
        ko = out[s]["k_out"] if np.iscomplexobj(dense[s]["k_out"]) else np.real(out[s]["k_out"])
        worst = max(worst, float(np.max(np.abs(dense[s]["k_out"][:, m] - ko[:, m]), initial=0.0)))
    n0 = case.x0.shape[1]
    T.host_trace(case, mode=P.MODE_IMAGE)
    T.host_trace(case, pitch=0)
    T.host_trace(case, in_pitch=n0 + 5)
    if hs.all_isotropic:
        T.host_trace(case, pitch=n0 + 1)
        T.host_trace(case, flags=True)
        for (pitch, n) in ((n0 + (n0 %% 2) + 2, n0), (None, n0 - 1), (n0 + 1 - (n0 %% 2), n0 - 1), (None, 1)):
            T._walk_surface_by_surface(case, pitch, n)
        T._walk_surface_by_surface(case, n0 + (n0 %% 2) + 2, n0, misalign=True)
        T.host_trace(case, misalign=True, pitch=n0 + 2 + n0 %% 2, in_pitch=n0 + 4 + n0 %% 2, want_nonconv=True)
    else:
        T.host_trace(case, want_fields=True, want_k_im=True)
T.test_host_build_many_blocks_and_the_xcd_block_map()
T.test_host_build_device_side_helpers()
T.test_host_build_bundle_generation_and_the_small_helpers()
T.test_host_build_fused_image_plane_moments()
T.test_host_build_empty_bundles_and_argument_errors()
T.test_host_build_poisoned_rays_are_masks_never_crashes()
T.test_host_build_wrappers_one_call_trace_timing_async_moments_and_the_image_redirect()
T.test_host_build_the_c_abi_from_several_host_threads()
T.test_host_build_size_sweep_around_the_block_and_wave_boundaries()
T.test_host_build_malformed_tables_are_error_codes()
print("RESULT " + json.dumps({"worst_k": worst}))
'''

SANITIZER_CASES = ["double_gauss_wide", "tilted_frames", "mirrors", "benchmark_divergent", "doublet_clipped",
                   "asphere_strong_field5", "xypoly_field5", "biconic_field5", "asphere_grazing_field30_tight",
                   "aniso_doublet_isoeps", "aniso_doublet_uniaxial", "aniso_doublet_biaxial", "aniso_absorbing_two_crystals"]


def test_host_build_under_address_and_undefined_behaviour_sanitizers():
    """the kernels' sources with -fsanitize=address,undefined (GPU sanitizers are not available on the pool): whole
    sequences in every layout and mode, the per-surface calls with odd sizes and unaligned arrays, the crystal march's
    concatenated layout, the moments / compaction / raster kernels -- on exact-size arrays, no report from either
    sanitizer, results equal the oracle's.  Runs in a subprocess: the instrumented library needs the sanitizer
    runtime preloaded into the interpreter."""
    try:
        lib = hostemu.build(sanitize=True)
    except Exception as exc:
        pytest.skip("no sanitizer build on this box: %s" % exc)
    asan = hostemu.sanitizer_preload()
    if asan is None:
        pytest.skip("clang's shared asan runtime not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = [c for c in SANITIZER_CASES if os.path.exists(os.path.join(_golden.GOLDEN_DIR, c + ".npz"))]
    script = SANITIZER_SCRIPT % dict(root=root, tests=os.path.join(root, "tests"), cases=cases)
    env = dict(os.environ, LD_PRELOAD=asan, PRT_HOSTEMU_LIBRARY=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23:detect_stack_use_after_return=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=1500)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["worst_k"] < 1e-11


def test_host_build_device_side_helpers():
    """the kernels that synchronise (fibres at __syncthreads, wave shuffles, ballots): deterministic bundle moments,
    order-preserving compaction, the shape evaluation entry point"""
    import ctypes
    lib = hostemu.load()
    rng = np.random.RandomState(5)
    for n in (1, 63, 64, 257, 5000):
        x = np.ascontiguousarray(rng.normal(size=(3, n)))
        mask = (rng.uniform(size=n) < 0.7).astype(np.uint8)
        out = (ctypes.c_double * 7)()
        ref3 = (ctypes.c_double * 3)(0.1, -0.2, 0.3)
        rc = lib.prt_bundle_moments(0, n, n, x.ctypes.data, mask.ctypes.data, 0, ref3, out, None)
        assert rc == 0, lib.prt_last_error()
        v = x[:, mask.astype(bool)] - np.array([0.1, -0.2, 0.3])[:, None]
        want = np.concatenate([[v.shape[1]], v.sum(axis=1), (v * v).sum(axis=1)])
        assert np.allclose(np.array(out[:]), want, rtol=1e-12, atol=1e-12)
        # compaction of two (3, n) arrays + ids behind the mask keeps the order
        src = [np.ascontiguousarray(rng.normal(size=n)) for _ in range(6)]
        dst = [np.full(n, np.nan) for _ in range(6)]
        srcp = (ctypes.c_void_p * 6)(*[a.ctypes.data for a in src])
        dstp = (ctypes.c_void_p * 6)(*[a.ctypes.data for a in dst])
        ids = np.arange(n, dtype=np.int64) * 3
        ids_out = np.full(n, -1, dtype=np.int64)
        scratch = np.zeros(int(lib.prt_compact_scratch_bytes(n)), dtype=np.uint8)
        count = ctypes.c_int64(-1)
        rc = lib.prt_compact(n, mask.ctypes.data, 6, srcp, dstp, ids.ctypes.data, ids_out.ctypes.data, None, None,
                             scratch.ctypes.data, ctypes.byref(count), None)
        assert rc == 0, lib.prt_last_error()
        m = mask.astype(bool)
        assert count.value == int(m.sum())
        for (a, b) in zip(src, dst):
            assert np.array_equal(b[:count.value], a[m])
        assert np.array_equal(ids_out[:count.value], ids[m])
    case = _golden.load_case("asphere_strong_field5")
    hs = hostemu.HostSystem(case.table)
    s = [i for (i, r) in enumerate(case.table) if r["shape"]["type"] != "conic"][0]
    (xx, yy) = (np.linspace(-3, 3, 77), np.linspace(2, -2, 77))
    (sag, grad) = hs.shape_eval(s, xx, yy)
    assert np.allclose(sag, oracle.shape_sag(case.table[s]["shape"], xx, yy), rtol=1e-13, atol=1e-14)


def test_the_sanitizer_build_does_see_an_overrun():
    """negative control of the test above: an output array five doubles short -> AddressSanitizer reports the write of
    the kernel (a frame of k_propagate_rows), inside a fibre"""
    try:
        lib = hostemu.build(sanitize=True)
    except Exception as exc:
        pytest.skip("no sanitizer build on this box: %s" % exc)
    asan = hostemu.sanitizer_preload()
    if asan is None:
        pytest.skip("clang's shared asan runtime not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, _golden, hostemu\n"
        "case = _golden.load_case('double_gauss_wide')\n"
        "hs = hostemu.HostSystem(case.table)\n"
        "n = case.x0.shape[1]\n"
        "x = np.ascontiguousarray(case.x0); k = np.ascontiguousarray(np.real(case.k0))\n"
        "short = np.empty(3 * n - 5); valid = np.zeros(n, dtype=np.uint8)\n"
        "hs.lib.prt_propagate_rows(hs._h, 0, n, x.ctypes.data, n, k.ctypes.data, n, None, None, None, 0, None,\n"
        "                          short.ctypes.data, n, valid.ctypes.data, None, None)\n" % (root, os.path.join(root, "tests")))
    env = dict(os.environ, LD_PRELOAD=asan, PRT_HOSTEMU_LIBRARY=lib, ASAN_OPTIONS="detect_leaks=0:exitcode=23")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 23 and "heap-buffer-overflow" in r.stderr and "k_propagate_rows" in r.stderr, r.stderr[-3000:]


def test_host_build_table_update_in_place_and_the_poisoned_system(monkeypatch):
    """prt_system_update (the optimiser's pattern): the updated system traces like a fresh one of the new table; a
    table that does not fit is refused with nothing touched; an enqueue that fails half way (PRT_TEST_FAIL_UPDATE=k,
    honoured only with PRT_TEST_HOOKS: conftest.py) poisons the system -- every later call is refused"""
    import ctypes
    from pyrate_amd import systems, _lib as P
    from pyrate_amd.surface_table import pack_table
    (o, k, e0) = systems.double_gauss_bundle(300, field_deg=2.0)
    (rec_d, rec_f) = (systems.double_gauss_records(), systems.double_gauss_records(486.1e-6))
    hs = hostemu.HostSystem(rec_d)
    fresh = hostemu.HostSystem(rec_f).trace(o, k, e0)
    before = hs.trace(o, k, e0)
    assert hs.lib.prt_system_update(hs._h, pack_table(rec_f), len(rec_f), None) == 0
    after = hs.trace(o, k, e0)
    assert not np.array_equal(before[-1]["x_hit"], after[-1]["x_hit"])
    for (a, b) in zip(after, fresh):
        assert np.array_equal(a["x_hit"], b["x_hit"], equal_nan=True) and np.array_equal(a["valid_out"], b["valid_out"])
    # another number of surfaces / an asphere where the system has no coefficient array: refused, the table stays
    assert hs.lib.prt_system_update(hs._h, pack_table(rec_f[:-1]), len(rec_f) - 1, None) == P.ERR_UNSUPPORTED
    asph = systems.asphere_records()
    assert hs.lib.prt_system_update(hs._h, pack_table((asph * 3)[:len(rec_f)]), len(rec_f), None) == P.ERR_UNSUPPORTED
    assert np.array_equal(hs.trace(o, k, e0)[-1]["x_hit"], after[-1]["x_hit"], equal_nan=True)
    for fail_at in (1, 2):
        victim = hostemu.HostSystem(rec_d)
        monkeypatch.setenv("PRT_TEST_FAIL_UPDATE", str(fail_at))
        rc = victim.lib.prt_system_update(victim._h, pack_table(rec_f), len(rec_f), None)
        monkeypatch.delenv("PRT_TEST_FAIL_UPDATE")
        assert rc == P.ERR_DEVICE and b"unusable" in victim.lib.prt_last_error()
        with pytest.raises(hostemu.HostemuError) as err:
            victim.trace(o, k, e0)
        assert err.value.code == P.ERR_DEVICE and "unusable" in str(err.value)
        assert victim.lib.prt_system_update(victim._h, pack_table(rec_f), len(rec_f), None) == P.ERR_DEVICE


def test_host_build_fused_image_plane_moments():
    """prt_trace_moments: the march's MOMENTS epilogue (wave shuffles, a block barrier, per-block partials, a second
    launch that adds them in a fixed order) = count / sum v / sum v^2 of the image-plane arrays the same launch wrote"""
    import ctypes
    from pyrate_amd import systems, _lib as P
    lib = hostemu.load()
    for (recs, n_ask) in ((systems.double_gauss_records(), 3000), (systems.asphere_records(), 700)):
        (o, k, e0) = systems.double_gauss_bundle(n_ask, field_deg=3.0) if len(recs) > 4 else \
            systems.double_gauss_bundle(n_ask, rpup=9.0, z0=-5.0)
        n = o.shape[1]
        hs = hostemu.HostSystem(recs)
        pitch = int(lib.prt_recommended_pitch(n))
        (x_img, k_img) = (np.full((3, pitch), np.nan), np.full((3, pitch), np.nan))
        (v_img, w_img) = (np.zeros(pitch, dtype=np.uint8), np.zeros(pitch, dtype=np.uint8))
        out7 = np.full(7, np.nan)
        scratch = np.zeros(int(lib.prt_trace_moments_scratch_doubles(n)))
        ref3 = (ctypes.c_double * 3)(0.0, 0.0, float(recs[-1]["g_shape"][2]))
        (oc, kc, ec) = [np.ascontiguousarray(a) for a in (o, k, np.real(e0))]
        rc = lib.prt_trace_moments(hs._h, n, 0, oc.ctypes.data, kc.ctypes.data, ec.ctypes.data, None, P.MODE_IMAGE, pitch,
                                   x_img.ctypes.data, k_img.ctypes.data, v_img.ctypes.data, w_img.ctypes.data, ref3,
                                   out7.ctypes.data, scratch.ctypes.data, None)
        assert rc == 0, lib.prt_last_error()
        dense = hs.trace(o, k, e0, mode=P.MODE_IMAGE)
        m = dense[0]["valid_out"].astype(bool)
        assert np.array_equal(w_img[:n].astype(bool), m) and m.sum() > 100
        assert np.array_equal(x_img[:, :n][:, m], dense[0]["x_hit"][:, m])
        v = dense[0]["x_hit"][:, m] - np.array(ref3[:])[:, None]
        want = np.concatenate([[m.sum()], v.sum(axis=1), (v * v).sum(axis=1)])
        assert np.allclose(out7, want, rtol=1e-12, atol=1e-12), (out7, want)


def test_the_stand_in_runtime_votes_over_active_lanes_and_shuffles_where_waves_reconverge(tmp_path):
    """self-test of tests/hostemu/hip/hip_runtime.h (selftest_votes.cpp): `__all` inside `if (i < N)` is a vote of the
    lanes that are there, the reduction behind the branch sees every lane"""
    clang = hostemu.find_clang()
    exe = str(tmp_path / "selftest_votes")
    subprocess.run([clang, "-x", "c++", "-std=c++17", "-O1", "-I" + hostemu.HERE, os.path.join(hostemu.HERE, "selftest_votes.cpp"),
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == ["640", "640", "220"], (r.stdout, r.stderr)


def test_host_build_bundle_generation_and_the_small_helpers():
    """the remaining kernels of the library on the host build: RectGrid + collimated bundle (bit-identical to the host
    raster), every deterministic raster against the REFERENCE's bundles (tests/golden/bundles.json: origins bit for
    bit, wave vectors to a few ulp of libm), Poynting directions, a unit E perpendicular to k, path sums"""
    import ctypes
    import math
    from pyrate_amd import systems, _lib as P
    from pyrate_amd.sampling2d import raster
    lib = hostemu.load()
    # RectGrid + collimated bundle: the whole raster and a shard of it
    for nray in (50, 1000, 5000):
        (o, k, e0) = systems.double_gauss_bundle(nray, field_deg=5.0)
        (npd, ndisk) = (ctypes.c_int64(), ctypes.c_int64())
        assert lib.prt_rect_grid_count(0, nray, ctypes.byref(npd), ctypes.byref(ndisk), None) == 0
        total = ndisk.value
        assert total == o.shape[1] and npd.value == int(round(math.sqrt(nray * 4.0 / math.pi)))
        field = 5.0 * math.pi / 180.
        prm = P.PrtCollimated()
        (prm.radius, prm.startx, prm.starty, prm.startz) = (5.0, 0.0, -10.0 * math.tan(field), -10.0)
        prm.k[:] = [0.0, math.sin(field), math.cos(field)]
        prm.e[:] = [0.0, math.cos(field), -math.sin(field)]
        for (lo, hi) in ((0, total), (total // 3, total // 3 + max(1, total // 2))):
            m = hi - lo
            (x, kk, ee) = (np.full((3, m), np.nan), np.full((3, m), np.nan), np.full((3, m), np.nan))
            rc = lib.prt_collimated_bundle(0, nray, lo, hi, ctypes.byref(prm), m, x.ctypes.data, kk.ctypes.data,
                                           ee.ctypes.data, None)
            assert rc == 0, lib.prt_last_error()
            assert np.array_equal(x, o[:, lo:hi]) and np.array_equal(kk, k[:, lo:hi]) and np.array_equal(ee, e0[:, lo:hi])
    ref = json.load(open(os.path.join(_golden.GOLDEN_DIR, "bundles.json")))
    objs = {"rect_60": raster.RectGrid(), "hex_45": raster.HexGrid(), "meridional_9": raster.MeridionalFan(),
            "sagital_8": raster.SagitalFan(), "circular_49": raster.CircularGrid()}
    for (key, case) in ref.items():
        pd = case["props"]
        tables_list = objs[case["raster"]].device_tables(case["nray"])
        (gx, gy) = objs[case["raster"]].getGrid(case["nray"])
        prm = P.PrtBundle()
        prm.kind = {"collimated": 0, "divergent": 1}[case["bundle"]]
        (prm.radius, prm.anglex, prm.angley, prm.index) = (pd["radius"], pd["anglex"], pd["angley"], case["index"])
        prm.start[:] = [pd["startx"], pd["starty"], pd["startz"]]
        if case["bundle"] == "collimated":
            unit = np.array([math.sin(pd["angley"]) * math.cos(pd["anglex"]), math.sin(pd["anglex"]),
                             math.cos(pd["angley"]) * math.cos(pd["anglex"])])
            prm.k[:] = list(case["index"] * unit)
            prm.e[:] = [1.0, 0.0, 0.0]
        structs = []
        counts = []
        for (xa, xb, ya, yb, clip) in tables_list:
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (xa, xb, ya, yb)]
            r = P.PrtRaster()
            (r.nj, r.ni) = (arrs[0].shape[0], arrs[1].shape[0])
            (r.xa, r.xb, r.ya, r.yb) = [a.ctypes.data for a in arrs]
            r.clip = 1 if clip else 0
            r._keep = arrs
            c = ctypes.c_int64()
            assert lib.prt_raster_count(0, ctypes.byref(r), ctypes.byref(c), None) == 0
            structs.append(r)
            counts.append(c.value)
        total = sum(counts)
        (xr, kr) = (np.array(case["x"]), np.array(case["k"]))
        assert total == xr.shape[1] == gx.shape[0], key
        (x, k, e) = (np.full((3, total), np.nan), np.full((3, total), np.nan), np.full((3, total), np.nan))
        pup = np.full((2, total), np.nan)
        base = 0
        for (r, c) in zip(structs, counts):
            if c:
                off = base * 8
                rc = lib.prt_raster_bundle(0, ctypes.byref(r), 0, c, ctypes.byref(prm), total, x.ctypes.data + off,
                                           k.ctypes.data + off, e.ctypes.data + off, pup.ctypes.data + off, None)
                assert rc == 0, lib.prt_last_error()
            base += c
        assert np.array_equal(pup, np.vstack((gx, gy))), key
        assert np.array_equal(x, xr), key
        assert np.abs(k - kr).max() < 2e-15, (key, np.abs(k - kr).max())
        assert np.abs(np.sqrt((k ** 2).sum(axis=0)) - case["index"]).max() < 4e-16, key
        if case["bundle"] == "divergent":
            assert np.abs((e * k).sum(axis=0)).max() < 1e-15 and np.abs((e ** 2).sum(axis=0) - 1).max() < 1e-15
    # Poynting direction of (k, E) (ray.py:136-152), a unit E perpendicular to k, path sums
    rng = np.random.RandomState(3)
    n = 777
    k = np.ascontiguousarray(rng.normal(size=(3, n)) + np.array([[0.], [0.], [3.]]))
    (er, ei) = (np.ascontiguousarray(rng.normal(size=(3, n))), np.ascontiguousarray(rng.normal(size=(3, n))))
    d = np.full((3, n), np.nan)
    assert lib.prt_poynting_dir(0, n, k.ctypes.data, er.ctypes.data, ei.ctypes.data, 0, d.ctypes.data, None) == 0
    E = er + 1j * ei
    S = np.real(np.sum(np.conj(E) * E, axis=0) * k - np.sum(E * k, axis=0) * np.conj(E))
    assert np.allclose(d, S / np.sqrt(np.sum(S ** 2, axis=0)), rtol=0, atol=1e-14)
    e = np.full((3, n), np.nan)
    assert lib.prt_efield_perp(0, n, k.ctypes.data, e.ctypes.data, None) == 0
    assert np.abs((e * k).sum(axis=0)).max() < 1e-14 and np.abs((e ** 2).sum(axis=0) - 1).max() < 1e-14
    xs = [np.ascontiguousarray(rng.normal(size=(3, n))) for _ in range(4)]
    out = np.full(n, np.nan)
    xt = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in xs])
    assert lib.prt_path_sums(0, 4, n, xt, None, 0, out.ctypes.data, None) == 0
    want = sum(np.sqrt(np.sum((xs[i + 1] - xs[i]) ** 2, axis=0)) for i in range(3))
    assert np.allclose(out, want, rtol=1e-14)


def test_host_build_empty_bundles_and_argument_errors():
    """edges of the C ABI on the host build (also under the sanitizers): empty bundles through every entry point (null
    arrays allowed), a single ray, and structural misuse -- which is an error code, never a crash: surface index out
    of range, a bad mode word, a struct of another size, a ray pitch smaller than the ray count, a negative count"""
    import ctypes
    from pyrate_amd import systems, _lib as P
    lib = hostemu.load()
    recs = systems.double_gauss_records()
    hs = hostemu.HostSystem(recs)
    crystal = hostemu.HostSystem(systems.aniso_doublet_records())
    empty = np.zeros((3, 0))
    for system in (hs, crystal):
        for mode in (P.MODE_PATH, P.MODE_IMAGE):
            a = P.PrtTraceArgs()
            a.struct_bytes = ctypes.sizeof(P.PrtTraceArgs)
            (a.mode, a.n0) = (mode, 0)
            assert lib.prt_trace_ex(system._h, ctypes.byref(a)) == 0
    assert lib.prt_propagate_rows(hs._h, 0, 0, None, 0, None, 0, None, None, None, 0, None, None, 0, None, None, None) == 0
    assert lib.prt_interact_rows(hs._h, 0, 0, None, 0, None, 0, None, None, 0, None, None, None) == 0
    assert lib.prt_surface_step_rows(hs._h, 0, 0, None, 0, None, 0, None, None, None, 0, None, None, None, 0, None, None,
                                     None, None) == 0
    assert lib.prt_propagate(hs._h, 0, 0, None, None, None, None, None, 0, None, None, None, None, None) == 0
    assert lib.prt_interact(hs._h, 0, 0, None, None, None, None, None, None, None, None, None) == 0
    assert lib.prt_efield_perp(0, 0, None, None, None) == 0 and lib.prt_poynting_dir(0, 0, None, None, None, 0, None, None) == 0
    out7 = (ctypes.c_double * 7)(*([9.0] * 7))
    assert lib.prt_bundle_moments(0, 0, 0, None, None, 0, None, out7, None) == 0 and list(out7) == [0.0] * 7
    kept = ctypes.c_int64(-1)
    assert lib.prt_compact(0, None, 0, None, None, None, None, None, None, None, ctypes.byref(kept), None) == 0
    assert kept.value == 0
    # one ray
    (o, k, e0) = systems.double_gauss_bundle(20, field_deg=1.0)
    one = hs.trace(o[:, :1], k[:, :1], e0[:, :1])
    many = hs.trace(o, k, e0)
    for (a1, am) in zip(one, many):
        assert np.array_equal(a1["x_hit"][:, 0], am["x_hit"][:, 0]) and a1["valid_out"][0] == am["valid_out"][0]
    # misuse
    x = np.ascontiguousarray(o[:, :8])
    bad = lambda rc: rc == P.ERR_INVALID_ARG            # noqa: E731
    assert bad(lib.prt_propagate_rows(hs._h, 99, 8, x.ctypes.data, 8, x.ctypes.data, 8, None, None, None, 0, None,
                                      x.ctypes.data, 8, None, None, None))
    assert bad(lib.prt_surface_step_rows(hs._h, -1, 8, x.ctypes.data, 8, x.ctypes.data, 8, None, None, None, 0, None,
                                         x.ctypes.data, x.ctypes.data, 8, None, None, None, None))
    assert bad(lib.prt_propagate_rows(hs._h, 0, -3, x.ctypes.data, 8, x.ctypes.data, 8, None, None, None, 0, None,
                                      x.ctypes.data, 8, None, None, None))
    a = P.PrtTraceArgs()
    a.struct_bytes = ctypes.sizeof(P.PrtTraceArgs) - 8
    (a.mode, a.n0) = (0, 8)
    assert bad(lib.prt_trace_ex(hs._h, ctypes.byref(a)))                 # a caller built against another layout
    a.struct_bytes = ctypes.sizeof(P.PrtTraceArgs)
    a.mode = 7
    (a.x0, a.k0) = (x.ctypes.data, x.ctypes.data)
    buf = np.zeros(3 * 12 * 8)
    mask = np.zeros(12 * 8, dtype=np.uint8)
    (a.x_hit, a.k_out, a.valid) = (buf.ctypes.data, buf.ctypes.data, mask.ctypes.data)
    assert bad(lib.prt_trace_ex(hs._h, ctypes.byref(a)))                 # bad mode
    a.mode = 0
    a.out_pitch = 4
    assert bad(lib.prt_trace_ex(hs._h, ctypes.byref(a)))                 # pitch < n0
    assert bad(lib.prt_trace_ex(None, ctypes.byref(a)))                  # no system
    assert b"" != lib.prt_last_error()
    n_in = (ctypes.c_int64 * 12)()
    assert bad(lib.prt_system_ray_counts(hs._h, -1, n_in, n_in))
    h = ctypes.c_void_p()
    from pyrate_amd.surface_table import pack_table
    assert bad(lib.prt_system_create(pack_table(recs), 0, 0, ctypes.byref(h)))


POISON_CASES = ["double_gauss_wide", "tilted_frames", "asphere_strong_field5", "xypoly_field5", "biconic_field5",
                "gridsag_field2", "zernike_fringe_field3", "zernike_combination_mirror", "aniso_doublet_uniaxial",
                "aniso_doublet_biaxial", "aniso_absorbing_two_crystals"]


def test_host_build_poisoned_rays_are_masks_never_crashes():
    """per-ray failure is a mask, never a crash (surface_shape.py:215-216, 321; helpers_math.py:32-37): rays whose inputs
    are NaN, +-Inf, 1e300, 1e-320 or a zero wave vector go through every shape evaluator (the sag grid's spline
    interval search, polynomial tables, the crystal solvers) -- under ASan / UBSan in the sanitizer test: no access
    outside an array, no undefined conversion -- and every OTHER ray of the bundle gets exactly the record it gets
    without them (a ray's result does not depend on its neighbours in the wave)"""
    poison = [np.nan, np.inf, -np.inf, 1e300, -1e300, 1e-320, 0.0]
    for name in POISON_CASES:
        if not os.path.exists(os.path.join(_golden.GOLDEN_DIR, name + ".npz")):
            continue
        case = _golden.load_case(name)
        (_, clean) = host_trace(case, want_nonconv=True)
        (x0, k0) = (np.array(case.x0, dtype=float), np.array(np.real(case.k0), dtype=float))
        n0 = x0.shape[1]
        hit = np.zeros(n0, dtype=bool)
        j = 0
        for (q, v) in enumerate(poison):
            for arr in (x0, k0):
                for comp in range(3):
                    col = (7 * j + 3) % n0
                    j += 1
                    if hit[col]:
                        continue
                    arr[comp, col] = v
                    hit[col] = True
        col = (7 * j + 3) % n0
        k0[:, col] = 0.0                       # a ray without a direction
        hit[col] = True
        e = np.asarray(case.E0)
        hs = hostemu.HostSystem(case.table)
        with np.errstate(all="ignore"):
            dirty = hs.trace(x0, k0, e.real, e.imag if np.iscomplexobj(e) else None, want_nonconv=True)
        assert (~hit).sum() > n0 // 3
        for s in range(case.n_surfaces):
            B_in = dirty[s]["x_hit"].shape[1] // n0
            B_out = dirty[s]["k_out"].shape[1] // n0
            keep_in = np.tile(~hit, B_in)
            keep_out = np.tile(~hit, B_out)
            assert np.array_equal(dirty[s]["valid"][keep_in], clean[s]["valid"][keep_in]), (name, s)
            assert np.array_equal(dirty[s]["valid_out"][keep_out], clean[s]["valid_out"][keep_out]), (name, s)
            if hs.all_isotropic:
                assert np.array_equal(dirty[s]["x_hit"][:, keep_in], clean[s]["x_hit"][:, keep_in], equal_nan=True), (name, s)
                assert np.array_equal(dirty[s]["k_out"][:, keep_out], clean[s]["k_out"][:, keep_out], equal_nan=True), (name, s)
            else:
                # the crystal solver picks its route by WAVE (votes: the expensive routes run only in waves that hold a
                # lane which needs them) -- a poisoned neighbour may send a ray through another, equally valid route:
                # the same record to rounding, not to the bit
                (a, b) = (dirty[s]["x_hit"][:, keep_in], clean[s]["x_hit"][:, keep_in])
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a, b, rtol=0, atol=1e-11, equal_nan=True), (name, s)
                (a, b) = (dirty[s]["k_out"][:, keep_out], clean[s]["k_out"][:, keep_out])
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a, b, rtol=0, atol=1e-12, equal_nan=True), (name, s)
        if hs.all_isotropic:
            # the same rays through the per-surface calls and the fused step
            (x, k, valid) = (x0, k0, None)
            for s in range(case.n_surfaces):
                first = dict(e_re=e.real, e_im=e.imag if np.iscomplexobj(e) else None) if s == 0 else dict(default_e=False)
                (xh, kk, v, vo) = hs.surface_step_rows(s, x, k, valid_in=valid, **first)
                (xh2, v2) = hs.propagate_rows(s, x, k, valid_in=valid, **first)
                (k2, vo2) = hs.interact_rows(s, xh2, k, valid_in=v2)
                assert np.array_equal(vo[~hit], dirty[s]["valid_out"][~hit]) and np.array_equal(vo2[~hit], vo[~hit]), (name, s)
                (x, k, valid) = (xh, kk, vo)


def _rotated_biaxial_pair():
    def rot(ax, ay, az):
        (ca, sa, cb, sb, cg, sg) = (np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az))
        rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
        ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
        rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
        return rz.dot(ry).dot(rx)
    (r1, r2) = (rot(0.4, 0.25, -0.3), rot(-0.2, 0.35, 0.15))              # the tensors of bench.py's aniso_biaxial config
    return (r1.dot(np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2])).dot(r1.T), r2.dot(np.diag([1.62 ** 2, 1.66 ** 2, 1.71 ** 2])).dot(r2.T))


def test_host_build_a_rotated_symmetric_tensor_is_used_as_its_symmetric_part():
    """A biaxial tensor rotated into place (R diag R^T) is symmetric up to an ulp; the solver's cheapest route (flux of
    the leaving pair from the adjugate of W, prt_aniso.h) wants exact symmetry and was dead for every such tensor -- the
    bench's aniso_biaxial config and every crystal test included (round 6: line coverage of the host build).
    prt_system_create now stores such a tensor as its symmetric part: the trace equals the trace with the explicitly
    symmetrised tensor BIT FOR BIT, agrees with the oracle on the caller's tensor, and a tensor with a real antisymmetric
    part is taken as given."""
    from pyrate_amd import systems
    (e1, e2) = _rotated_biaxial_pair()
    assert np.abs(e1 - e1.T).max() > 0 and np.abs(e1 - e1.T).max() < 1e-15
    (o, k, e0) = systems.double_gauss_bundle(600, rpup=11.43, z0=-5.0, field_deg=2.0)
    raw = hostemu.HostSystem(systems.aniso_doublet_records(e1, e2)).trace(o, k, e0)
    sym = hostemu.HostSystem(systems.aniso_doublet_records(0.5 * (e1 + e1.T), 0.5 * (e2 + e2.T))).trace(o, k, e0)
    with np.errstate(all="ignore"):
        out = oracle.trace(systems.aniso_doublet_records(e1, e2), o, k, e0)
    for (a, b, ref) in zip(raw, sym, out):
        assert np.array_equal(a["x_hit"], b["x_hit"], equal_nan=True) and np.array_equal(a["k_out"], b["k_out"], equal_nan=True)
        assert np.array_equal(a["valid_out"], b["valid_out"])
        ko = np.real(ref["k_out"])
        fin = np.all(np.isfinite(ko), axis=0)
        assert np.abs(a["k_out"][:, fin] - ko[:, fin]).max() < 1e-12
        v = ref["valid"] & np.all(np.isfinite(ref["x_hit"]), axis=0)
        assert np.abs(a["x_hit"][:, v] - ref["x_hit"][:, v]).max() < 1e-11
    # a real antisymmetric part (1e-9 of the entries: not rounding) stays: the trace differs from the symmetrised one
    skew = 1e-9 * np.array([[0, 1, -1], [-1, 0, 1], [1, -1, 0.0]])
    given = hostemu.HostSystem(systems.aniso_doublet_records(e1 + skew, e2)).trace(o, k, e0)
    assert not np.array_equal(given[-1]["x_hit"], raw[-1]["x_hit"], equal_nan=True)
    with np.errstate(all="ignore"):
        out = oracle.trace(systems.aniso_doublet_records(e1 + skew, e2), o, k, e0)
    ko = np.real(out[-1]["k_out"])
    fin = np.all(np.isfinite(ko), axis=0)
    assert np.abs(given[-1]["k_out"][:, fin] - ko[:, fin]).max() < 1e-12


def test_host_build_wrappers_one_call_trace_timing_async_moments_and_the_image_redirect():
    """the remaining entry points of include/prt.h on the host build: prt_trace_seq (no handle: tables cached by content,
    more tables than the cache holds), prt_trace / prt_trace_timed (wrappers of prt_trace_ex), prt_bundle_moments_async
    with the reference on the device (a point, a moments vector), the image-plane redirect of the last surface's record
    (what a ray-sharded trace deposits into the all-gather's receive buffer), the tight per-surface calls"""
    import ctypes
    from pyrate_amd import systems, _lib as P
    from pyrate_amd.surface_table import pack_table
    lib = hostemu.load()
    (o, k, e0) = systems.double_gauss_bundle(500, field_deg=3.0)
    n = o.shape[1]
    (oc, kc) = (np.ascontiguousarray(o), np.ascontiguousarray(k))
    d0 = np.ascontiguousarray(k / np.sqrt((k * k).sum(axis=0)))
    # prt_trace_seq: twelve different tables through an eight-entry cache, each twice
    base = systems.double_gauss_records()
    for rep in range(2):
        for q in range(12):
            recs = systems.double_gauss_records(587.6e-6 * (1 + 0.01 * q))
            S = len(recs)
            (x_hit, k_out, valid) = (np.full((S, 3, n), np.nan), np.full((S, 3, n), np.nan), np.zeros((S, n), dtype=np.uint8))
            rc = lib.prt_trace_seq(pack_table(recs), S, n, oc.ctypes.data, kc.ctypes.data, d0.ctypes.data if q % 2 else None, None, P.MODE_PATH,
                                   x_hit.ctypes.data, k_out.ctypes.data, valid.ctypes.data, None, 0, None)
            assert rc == 0, lib.prt_last_error()
            ref = hostemu.HostSystem(recs).trace(o, k, first_dir=P.FIRST_K, pitch=0)
            for s in range(S):
                assert np.array_equal(valid[s], ref[s]["valid"])
                if q % 2:       # the caller's unit directions (k / |k| computed here) against the library's own: rounding
                    assert np.allclose(x_hit[s], ref[s]["x_hit"], rtol=0, atol=1e-12, equal_nan=True)
                else:
                    assert np.array_equal(x_hit[s], ref[s]["x_hit"], equal_nan=True)
    # prt_trace and prt_trace_timed
    hs = hostemu.HostSystem(base)
    S = len(base)
    ref = hs.trace(o, k, e0, pitch=0)
    ec = np.ascontiguousarray(np.real(e0))
    for timed in (False, True):
        (x_hit, k_out) = (np.full((S, 3, n), np.nan), np.full((S, 3, n), np.nan))
        (valid, valid_out) = (np.zeros((S, n), dtype=np.uint8), np.zeros((S, n), dtype=np.uint8))
        if timed:
            ms = ctypes.c_double(-1.0)
            rc = lib.prt_trace_timed(hs._h, n, 0, oc.ctypes.data, kc.ctypes.data, ec.ctypes.data, None, P.MODE_PATH, 0,
                                     x_hit.ctypes.data, k_out.ctypes.data, valid.ctypes.data, valid_out.ctypes.data, None, 3,
                                     ctypes.byref(ms))
            assert rc == 0 and ms.value >= 0.0
        else:
            rc = lib.prt_trace(hs._h, n, 0, oc.ctypes.data, kc.ctypes.data, ec.ctypes.data, None, P.MODE_PATH, 0,
                               x_hit.ctypes.data, k_out.ctypes.data, valid.ctypes.data, valid_out.ctypes.data, None, None)
            assert rc == 0, lib.prt_last_error()
        for s in range(S):
            assert np.array_equal(k_out[s], ref[s]["k_out"], equal_nan=True) and np.array_equal(valid_out[s], ref[s]["valid_out"])
    # the image-plane redirect: the last record lands in rows of its own, the path arrays' last rows stay untouched
    pitch = int(lib.prt_recommended_pitch(n))
    a = P.PrtTraceArgs()
    a.struct_bytes = ctypes.sizeof(P.PrtTraceArgs)
    (a.mode, a.n0, a.in_pitch, a.out_pitch) = (P.MODE_PATH | P.MODE_FLAGS, n, pitch, pitch)
    (op, kp, ep) = [hostemu._rows(t, pitch) for t in (o, k, np.real(e0))]       # (the redirect wants 16-B aligned rows everywhere)
    (a.x0, a.k0, a.e0_re, a.first_dir) = (op.ctypes.data, kp.ctypes.data, ep.ctypes.data, P.FIRST_E)
    (x_hit, k_out, flags) = (np.full((S, 3, pitch), np.nan), np.full((S, 3, pitch), np.nan), np.full((S, pitch), 9, dtype=np.uint8))
    img_pitch = pitch + 512
    (x_img, k_img, f_img) = (np.full((3, img_pitch), np.nan), np.full((3, img_pitch), np.nan), np.full(img_pitch, 9, dtype=np.uint8))
    (a.x_hit, a.k_out, a.valid) = (x_hit.ctypes.data, k_out.ctypes.data, flags.ctypes.data)
    (a.x_img, a.k_img, a.valid_img, a.img_pitch) = (x_img.ctypes.data, k_img.ctypes.data, f_img.ctypes.data, img_pitch)
    assert lib.prt_trace_ex(hs._h, ctypes.byref(a)) == 0, lib.prt_last_error()
    assert np.all(np.isnan(x_hit[-1])) and np.all(flags[-1] == 9)
    m = ref[-1]["valid_out"].astype(bool)
    assert np.array_equal((f_img[:n] >> 1) & 1, ref[-1]["valid_out"]) and np.array_equal(f_img[:n] & 1, ref[-1]["valid"])
    assert np.allclose(x_img[:, :n][:, m], ref[-1]["x_hit"][:, m], rtol=0, atol=1e-13)
    assert np.allclose(x_hit[-2][:, :n][:, m], ref[-2]["x_hit"][:, m], rtol=0, atol=1e-13)
    # moments with the reference on the device: a point, then the centroid of a first pass
    x = np.ascontiguousarray(ref[-1]["x_hit"])
    mask = np.ascontiguousarray(ref[-1]["valid_out"])
    scratch = np.zeros(int(lib.prt_moments_scratch_doubles(n)))
    (first, second) = (np.full(7, np.nan), np.full(7, np.nan))
    point = np.array([0.0, 0.0, float(x[2, m].mean())])
    assert lib.prt_bundle_moments_async(0, n, 0, x.ctypes.data, mask.ctypes.data, 0, point.ctypes.data, 1, first.ctypes.data,
                                        scratch.ctypes.data, None) == 0
    v = x[:, m] - point[:, None]
    assert np.allclose(first, np.concatenate([[m.sum()], v.sum(axis=1), (v * v).sum(axis=1)]), rtol=1e-12, atol=1e-12)
    plain = np.full(7, np.nan)
    assert lib.prt_bundle_moments_async(0, n, 0, x.ctypes.data, mask.ctypes.data, 0, None, 0, plain.ctypes.data,
                                        scratch.ctypes.data, None) == 0
    assert lib.prt_bundle_moments_async(0, n, 0, x.ctypes.data, mask.ctypes.data, 0, plain.ctypes.data, 2, second.ctypes.data,
                                        scratch.ctypes.data, None) == 0
    v = x[:, m] - (x[:, m].sum(axis=1) / m.sum())[:, None]
    assert np.allclose(second[4:], (v * v).sum(axis=1), rtol=1e-9, atol=1e-12) and abs(second[1]) < 1e-9
    # the tight one-ray-per-thread calls (what the per-surface march of many crystals launches), isotropic surface
    (xh, v1) = hs.propagate(0, o, k, e_re=np.real(e0))
    assert np.array_equal(v1, ref[0]["valid"]) and np.allclose(xh[:, v1.astype(bool)], ref[0]["x_hit"][:, v1.astype(bool)], rtol=0, atol=1e-13)
    k2 = np.full((3, n), np.nan)
    w2 = np.zeros(n, dtype=np.uint8)
    xhc = np.ascontiguousarray(xh)
    assert lib.prt_interact(hs._h, 0, n, xhc.ctypes.data, kc.ctypes.data, v1.ctypes.data, k2.ctypes.data, None, None, None,
                            w2.ctypes.data, None) == 0
    assert np.array_equal(w2, ref[0]["valid_out"])
    assert np.allclose(k2[:, w2.astype(bool)], ref[0]["k_out"][:, w2.astype(bool)], rtol=0, atol=1e-14)


def test_host_build_the_c_abi_from_several_host_threads():
    """the library's shared host state under concurrent callers (ctypes releases the GIL inside a call; the stand-in
    runtime keeps its launch state per OS thread): prt_trace_seq's table cache (a mutex, more tables than entries, evictions
    while other threads launch), prt_system_create / update / destroy, thread-local error strings -- every result equals
    the single-threaded one, no thread sees another thread's error text.  Also run under AddressSanitizer (sanitizer test)."""
    import ctypes
    import threading
    from pyrate_amd import systems, _lib as P
    from pyrate_amd.surface_table import pack_table
    lib = hostemu.load()
    (o, k, e0) = systems.double_gauss_bundle(300, field_deg=2.0)
    n = o.shape[1]
    (oc, kc) = (np.ascontiguousarray(o), np.ascontiguousarray(k))
    tables = [systems.double_gauss_records(587.6e-6 * (1 + 0.01 * q)) for q in range(12)]
    packed = [pack_table(t) for t in tables]
    S = len(tables[0])
    want = [hostemu.HostSystem(t).trace(o, k, first_dir=P.FIRST_K, pitch=0) for t in tables]
    errors = []

    def worker(tid):
        try:
            rng = np.random.RandomState(tid)
            for it in range(30):
                q = int(rng.randint(0, 12))
                if it % 3 == 0:          # the one-call form: the shared table cache
                    (x_hit, k_out, valid) = (np.full((S, 3, n), np.nan), np.full((S, 3, n), np.nan), np.zeros((S, n), dtype=np.uint8))
                    rc = lib.prt_trace_seq(packed[q], S, n, oc.ctypes.data, kc.ctypes.data, None, None, P.MODE_PATH,
                                           x_hit.ctypes.data, k_out.ctypes.data, valid.ctypes.data, None, 0, None)
                    assert rc == 0
                    got = x_hit[-1]
                elif it % 3 == 1:        # a system of this thread's own, updated in place to another table
                    hs = hostemu.HostSystem(tables[(q + 1) % 12], lib=lib)
                    assert lib.prt_system_update(hs._h, packed[q], S, None) == 0
                    got = hs.trace(o, k, first_dir=P.FIRST_K, pitch=0)[-1]["x_hit"]
                    hs.close()
                else:                    # an error of this thread's own: its text must not leak into another thread
                    bad = lib.prt_propagate_rows(None, tid, 1, None, 0, None, 0, None, None, None, 0, None, None, 0, None, None, None)
                    assert bad == P.ERR_INVALID_ARG and b"prt_propagate_rows" in lib.prt_last_error()
                    hs = hostemu.HostSystem(tables[q], lib=lib)
                    got = hs.trace(o, k, first_dir=P.FIRST_K, pitch=0)[-1]["x_hit"]
                    assert b"prt_propagate_rows" in lib.prt_last_error()       # still this thread's last error
                    hs.close()
                assert np.array_equal(got, want[q][-1]["x_hit"], equal_nan=True), (tid, it, q)
        except BaseException as exc:          # noqa: BLE001
            errors.append((tid, repr(exc)[:300]))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_host_build_under_thread_sanitizer():
    """race detection on the library's shared host state (the table cache of the handle-free trace, system creation and
    destruction, error strings): the concurrent callers of the test above under ThreadSanitizer -- no report; and the
    negative control: two threads updating ONE system at once, which include/prt.h rules out, IS reported (a data race in
    prt_system_update).  The stand-in runtime tells the sanitizer about its fibres."""
    try:
        lib = hostemu.build(sanitize="thread")
    except Exception as exc:
        pytest.skip("no ThreadSanitizer build on this box: %s" % exc)
    tsan = hostemu.sanitizer_preload("tsan")
    if tsan is None:
        pytest.skip("clang's shared tsan runtime not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = "import sys\nsys.path.insert(0, %r); sys.path.insert(0, %r)\n" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, LD_PRELOAD=tsan, PRT_HOSTEMU_LIBRARY=lib, TSAN_OPTIONS="exitcode=23:report_signal_unsafe=0")
    good = head + "import test_hostemu as T\nT.test_host_build_the_c_abi_from_several_host_threads()\nprint('RESULT ok')\n"
    r = subprocess.run([sys.executable, "-c", good], env=env, capture_output=True, text=True, timeout=1500)
    assert "ThreadSanitizer" not in r.stderr and r.returncode == 0 and "RESULT ok" in r.stdout, r.stderr[-3000:]
    bad = head + (
        "import threading, hostemu\n"
        "from pyrate_amd import systems\n"
        "from pyrate_amd.surface_table import pack_table\n"
        "lib = hostemu.load()\n"
        "recs = systems.double_gauss_records()\n"
        "hs = hostemu.HostSystem(recs, lib=lib)\n"
        "tabs = [pack_table(systems.double_gauss_records(587.6e-6 * (1 + 0.01 * q))) for q in range(4)]\n"
        "def w(t):\n"
        "    for i in range(200):\n"
        "        lib.prt_system_update(hs._h, tabs[(t + i) % 4], len(recs), None)\n"
        "ts = [threading.Thread(target=w, args=(t,)) for t in range(2)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n")
    r = subprocess.run([sys.executable, "-c", bad], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 23 and "ThreadSanitizer: data race" in r.stderr and "prt_system_update" in r.stderr, r.stderr[-2000:]


def test_the_product_refuses_the_host_build(monkeypatch):
    """pyrate_amd/_lib.py has a switch for alternative builds of the same ABI (PRT_LIBRARY, A/B experiments): pointed at the
    host build it raises ImportError -- the host build answers prt_abi_version() with 1000 + the sources' version"""
    import importlib
    from pyrate_amd import _lib as product
    path = hostemu.build()
    assert hostemu.load().prt_abi_version() == 1000 + product.ABI_VERSION
    monkeypatch.setattr(product, "LIB_PATH", path)
    monkeypatch.setattr(product, "_lib", None)
    with pytest.raises(ImportError) as err:
        product.load()
    assert "ABI version" in str(err.value)
    assert product._lib is None


def test_host_build_size_sweep_around_the_block_and_wave_boundaries():
    """ray counts around every boundary of the launch geometry -- a thread owns two rays, a wave 128, a block of the march
    256, the XCD-contiguous block map wants grids that are multiples of 8 -- for the fused march (path, image, flags), the
    fused image-plane moments (whose tail threads join the reduction without a ray) and the crystal march (blocks of 256
    threads, one ray each): every count gives the oracle's masks and values, and the moments of exactly its rays"""
    import ctypes
    from pyrate_amd import systems, _lib as P
    lib = hostemu.load()
    recs = systems.double_gauss_records()
    (o, k, e0) = systems.double_gauss_bundle(6000, field_deg=4.0)
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, o, k, e0)
    hs = hostemu.HostSystem(recs)
    sizes = sorted(set(list(range(1, 12)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 767, 1023, 1024, 1025,
                                             2047, 2048, 2049, 2303, 2305, 4095, 4097, o.shape[1]]))
    for n in sizes:
        (on, kn, en) = (o[:, :n], k[:, :n], e0[:, :n])
        for kw in (dict(), dict(mode=P.MODE_IMAGE), dict(flags=True), dict(pitch=0)):
            d = hs.trace(on, kn, en, **kw)
            for (s, rec) in enumerate(d):
                so = (len(recs) - 1) if kw.get("mode") == P.MODE_IMAGE else s
                assert np.array_equal(rec["valid_out"].astype(bool), out[so]["valid_out"][:n]), (n, kw, s)
                m = out[so]["valid_out"][:n]
                assert np.max(np.abs(rec["x_hit"][:, m] - out[so]["x_hit"][:, :n][:, m]), initial=0.0) < 1e-11, (n, kw, s)
        # fused moments
        pitch = int(lib.prt_recommended_pitch(n))
        (x_img, k_img) = (np.full((3, pitch), np.nan), np.full((3, pitch), np.nan))
        (v_img, w_img) = (np.zeros(pitch, dtype=np.uint8), np.zeros(pitch, dtype=np.uint8))
        out7 = np.full(7, np.nan)
        scratch = np.zeros(int(lib.prt_trace_moments_scratch_doubles(n)))
        (oc, kc, ec) = [hostemu._rows(a, pitch) for a in (on, kn, en)]
        rc = lib.prt_trace_moments(hs._h, n, pitch, oc.ctypes.data, kc.ctypes.data, ec.ctypes.data, None, P.MODE_IMAGE, pitch,
                                   x_img.ctypes.data, k_img.ctypes.data, v_img.ctypes.data, w_img.ctypes.data, None,
                                   out7.ctypes.data, scratch.ctypes.data, None)
        assert rc == 0, lib.prt_last_error()
        m = out[-1]["valid_out"][:n]
        v = out[-1]["x_hit"][:, :n][:, m] - np.asarray(recs[-1]["g_shape"], dtype=float)[:, None]
        want = np.concatenate([[m.sum()], v.sum(axis=1), (v * v).sum(axis=1)])
        assert np.allclose(out7, want, rtol=1e-11, atol=1e-9), (n, out7, want)
    # the crystal march
    case = _golden.load_case("aniso_doublet_biaxial")
    with np.errstate(all="ignore"):
        outc = oracle.trace(case.table, case.x0, case.k0, case.E0)
    hc = hostemu.HostSystem(case.table)
    n0 = case.x0.shape[1]
    big = 3 * 256 + 5
    reps = -(-big // n0)
    (xb, kb, eb) = [np.tile(np.real(a), (1, reps))[:, :big] for a in (case.x0, case.k0, case.E0)]
    idx = np.arange(big) % n0
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 511, 513, big):
        d = hc.trace(xb[:, :n], kb[:, :n], eb[:, :n])
        for s in range(case.n_surfaces):
            B = d[s]["k_out"].shape[1] // n
            ref_k = np.real(outc[s]["k_out"]).reshape(3, B, n0)[:, :, idx[:n]].reshape(3, B * n)
            fin = np.all(np.isfinite(ref_k), axis=0)
            assert np.max(np.abs(d[s]["k_out"][:, fin] - ref_k[:, fin]), initial=0.0) < 1e-11, (n, s)


def test_host_build_malformed_tables_are_error_codes():
    """a C caller's prt_surface_t with fields out of range (an unknown shape, coefficient counts beyond the arrays, a sag
    grid without data, polynomial powers outside the tables, enums out of range): prt_system_create / prt_system_update
    answer with an error code and a text that names the surface -- under the sanitizers too: nothing is read beyond the
    record before it is validated"""
    import copy
    from pyrate_amd import systems, _lib as P
    from pyrate_amd.surface_table import pack_table
    lib = hostemu.load()
    good = systems.asphere_records()
    xy = _golden.load_case("xypoly_field5").table
    i_xy = [i for (i, r) in enumerate(xy) if r["shape"]["type"] not in ("conic",)][0]

    def create(table, n):
        import ctypes
        h = ctypes.c_void_p()
        rc = lib.prt_system_create(table, n, 0, ctypes.byref(h))
        if rc == 0:
            lib.prt_system_destroy(h)
        return rc, lib.prt_last_error().decode()
    assert create(pack_table(good), len(good))[0] == 0

    def broken(records, surface, **fields):
        t = pack_table(copy.deepcopy(records))
        t2 = type(t)()                       # (pack_table may hand out a cached array: work on a copy)
        import ctypes
        ctypes.memmove(t2, t, ctypes.sizeof(t))
        for (name, value) in fields.items():
            if "[" in name:
                (arr, idx) = name[:-1].split("[")
                getattr(t2[surface], arr)[int(idx)] = value
            else:
                setattr(t2[surface], name, value)
        return t2
    i_as = [i for (i, r) in enumerate(good) if r["shape"]["type"] == "asphere"][0]
    cases = [(good, i_as, dict(shape_type=99), P.ERR_UNSUPPORTED), (good, i_as, dict(shape_type=-1), P.ERR_UNSUPPORTED),
             (good, i_as, dict(n_coeffs=-3), P.ERR_INVALID_ARG), (good, i_as, dict(n_coeffs=100000), P.ERR_INVALID_ARG),
             (good, 0, dict(ap_type=7), P.ERR_INVALID_ARG), (good, 0, dict(interaction=5), P.ERR_INVALID_ARG),
             (good, 0, dict(mat_type=-2), P.ERR_INVALID_ARG),
             (xy, i_xy, {"xpow[0]": 10000}, P.ERR_INVALID_ARG), (xy, i_xy, {"ypow[0]": -1}, P.ERR_INVALID_ARG)]
    for (records, surface, fields, want) in cases:
        (rc, text) = create(broken(records, surface, **fields), len(records))
        assert rc == want and ("surface %d" % surface) in text, (fields, rc, text)
    # a healthy system refuses such a table in an update as well, and keeps working
    hs = hostemu.HostSystem(good)
    (o, k, e0) = systems.double_gauss_bundle(100, rpup=9.0, z0=-5.0)
    before = hs.trace(o, k, e0)
    assert lib.prt_system_update(hs._h, broken(good, i_as, n_coeffs=-3), len(good), None) == P.ERR_INVALID_ARG
    after = hs.trace(o, k, e0)
    assert np.array_equal(before[-1]["x_hit"], after[-1]["x_hit"], equal_nan=True)
