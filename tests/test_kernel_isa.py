"""
What the compiler made of the march kernels (hipcc cross-compiles gfx950 without a GPU; skipped where there
is no hipcc): the properties DESIGN.md 4 relies on, read from the assembly by benchmarks/isa_report.py.

  * no flat_load / flat_store in any march kernel: every table access is a scalar load (constant address
    space) or a global access (a generic pointer costs a vector-memory round trip per coefficient and a
    vmcnt(0) that also drains the path stores -- found in round 2 on the asphere and polynomial shapes)
  * no s_waitcnt vmcnt inside the surface loop of the kernels that only store there (conic and asphere
    marches): gfx950 counts loads and stores in one counter
  * registers / occupancy of the instantiations the BASELINE configs run
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))


@pytest.fixture(scope="module")
def isa():
    from pyrate_amd import build
    if build.find_hipcc() is None:
        pytest.skip("no hipcc on this box")
    import isa_report
    rep = isa_report.report()
    return {k["kernel"]: k for k in rep["kernels"]}


def test_no_flat_memory_operations_in_the_march_kernels(isa):
    assert len(isa) >= 30
    for (name, k) in isa.items():
        assert k["flat_memory_ops"] == 0, name


def _iso_args(name):
    """k_trace_iso<MODE,VEC_IN,VEC_OUT,SHAPES,MOMENTS,UNI,IMG> -> the seven template arguments"""
    return [int(v) for v in name[len("k_trace_iso<"):-1].split(",")]


def test_store_only_loops_never_wait_on_the_memory_counter(isa):
    """conic (SHAPES = 0), asphere (1) and polynomial / biconic (2) instantiations, path mode: their surface loop contains stores and scalar loads only, so no vmcnt wait may appear in it --
    with the uniform first segment and the fused moments as well"""
    seen = 0
    for (name, k) in isa.items():
        if not name.startswith("k_trace_iso<"):
            continue
        (mode, _vi, _vo, shapes, _mom, _uni, _img) = _iso_args(name)
        if mode == 0 and shapes in (0, 1, 2):
            assert k["vmcnt_waits_in_loops"] == 0 and k["scratch_bytes_per_lane"] == 0, name
            seen += 1
    assert seen >= 20


def test_registers_and_occupancy_of_the_baseline_instantiations(isa):
    for uni in (0, 1):                            # arrays k0 / E0, and the uniform first segment of collimated bundles
        head = isa["k_trace_iso<0,1,1,0,0,%d,0>" % uni]   # BASELINE configs[1]: path mode, 2 rays per lane, conics only
        assert head["scratch_bytes_per_lane"] == 0 and head["waves_per_simd"] >= 7 and head["vgprs"] <= 72
        asph = isa["k_trace_iso<0,1,1,1,0,%d,0>" % uni]   # configs[2]: conics + even aspheres
        assert asph["scratch_bytes_per_lane"] == 0 and asph["waves_per_simd"] >= 5 and asph["vgprs"] <= 96
        # conics + aspheres + XY polynomials + biconics (north_star's shapes + SURVEY 8 f3): <= 96 VGPRs = 5 waves
        poly = isa["k_trace_iso<0,1,1,2,0,%d,0>" % uni]
        assert poly["scratch_bytes_per_lane"] == 0 and poly["waves_per_simd"] >= 5 and poly["vgprs"] <= 96
        image = isa["k_trace_iso<1,1,1,0,0,%d,0>" % uni]  # image mode of the conic march: capped at 64 VGPRs for 8 waves
        assert image["waves_per_simd"] == 8
    # configs[3]: conic surfaces, uniaxial crystals, parking slots in LDS, no E output (k_trace_general<MODE, GENERAL,
    # PARK_LDS, WANT_E, SHAPES>): no scratch, NO vmcnt wait inside the walk, and -- the ray directions coming from
    # closed forms instead of eigenvectors, a parked child being (x, k, flags) -- few enough registers for SIX waves
    # per SIMD (the 98 B of LDS per thread allow six blocks per CU)
    crystal = isa["k_trace_general<0,0,1,0,0>"]
    assert crystal["scratch_bytes_per_lane"] == 0 and crystal["waves_per_simd"] >= 6 and crystal["vgprs"] <= 80
    assert crystal["vmcnt_waits_in_loops"] == 0
    # path rows through a scalar base + lane offset; only the byte masks keep 64-bit lane addresses
    assert crystal["scalar_base_stores"] >= 12 and crystal["vector_address_stores"] <= 6
    allshapes = isa["k_trace_iso<0,1,1,3,0,0,0>"]   # every explicit shape compiled in (sag grids, combinations)
    assert allshapes["scratch_bytes_per_lane"] == 0 and allshapes["waves_per_simd"] >= 4


def test_the_per_surface_kernels_of_big_bundles(isa):
    """k_propagate_rows<VEC, SHAPES> / k_interact_iso_rows<VEC, SHAPES> (round 6: Material.propagate / refract at plugin
    granularity): the instantiation for a conic surface -- what the double Gauss runs, 24 launches per sweep -- keeps the
    register count of a streaming kernel (8 waves per SIMD), no instantiation spills or uses flat addressing, and the
    two-rays-per-thread form moves a row with global_load / global_store dwordx4"""
    conic_p = isa["k_propagate_rows<1,0>"]
    conic_i = isa["k_interact_iso_rows<1,0>"]
    assert conic_p["vgprs"] <= 64 and conic_p["waves_per_simd"] == 8 and conic_i["vgprs"] <= 64 and conic_i["waves_per_simd"] == 8
    names = [n for n in isa if n.startswith(("k_propagate_rows<", "k_interact_iso_rows<"))]
    assert len(names) == 8, names          # propagate: 4 shape classes (VEC) + the unaligned fall-back; interact: conic, general, fall-back
    for n in names:
        assert isa[n]["scratch_bytes_per_lane"] == 0 and isa[n]["flat_memory_ops"] == 0, n
    for level in (1, 2):                   # aspheres; XY polynomials / biconics: their own Newton code only
        assert isa["k_propagate_rows<1,%d>" % level]["vgprs"] <= 72, level


def test_the_fused_surface_step_kernel(isa):
    """k_surface_step_rows<VEC, SHAPES> (prt_surface_step_rows: propagate + refract of one surface in one launch): the
    conic instantiation stays a streaming kernel (8 waves), the Newton levels keep the fused march's five waves, nothing
    spills or uses flat addressing"""
    conic = isa["k_surface_step_rows<1,0>"]
    assert conic["vgprs"] <= 64 and conic["waves_per_simd"] == 8
    names = [n for n in isa if n.startswith("k_surface_step_rows<")]
    assert len(names) == 5, names          # 4 shape classes (VEC) + the unaligned fall-back
    for n in names:
        assert isa[n]["scratch_bytes_per_lane"] == 0 and isa[n]["flat_memory_ops"] == 0, n
    for level in (1, 2):
        assert isa["k_surface_step_rows<1,%d>" % level]["waves_per_simd"] >= 5, level
