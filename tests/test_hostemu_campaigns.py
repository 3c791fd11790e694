"""The random-system campaigns of the `-m gpu` suite (tests/test_gpu_fuzz.py), run on the HOST build of libprt's sources
(tests/hostemu): the bodies of the GPU tests are called as they are, with ``pyrate_amd.engine.DeviceSystem`` /
``to_device_rays`` swapped (pytest monkeypatch, this test only) for the host build's adapter.  CPU only.  What it covers that
the golden cases do not: Newton on harsh shapes (misses, NaN domains, the iteration cap), random crystals incl. the
wave-uniform fall-backs of the biaxial solver (votes across the lanes of a wave), mirrors inside crystals, partially
evanescent interfaces, the per-surface march through more crystal interfaces than the fused walk parks."""
import pytest
import torch

hostemu = pytest.importorskip("hostemu")


@pytest.fixture(scope="module", autouse=True)
def _host_library():
    """built on first use (not at import: `pytest -m gpu` on the GPU box collects this module and deselects all of it)"""
    try:
        hostemu.build()
    except Exception as exc:                                    # no clang++ on this box
        pytest.skip("no host build of libprt: %s" % exc)

import test_gpu_fuzz as F           # noqa: E402  (plain functions: the gpu marker belongs to the module's collection)
from hostemu import adapter        # noqa: E402

HOST = torch.device("cpu")


@pytest.fixture
def host_engine():
    from pyrate_amd import engine
    mp = pytest.MonkeyPatch()          # (its own instance: a GPU test body may undo() the function-scoped one)
    mp.setattr(engine, "DeviceSystem", adapter.HostDeviceSystem)
    mp.setattr(engine, "to_device_rays", adapter.to_device_rays)
    yield engine
    mp.undo()


@pytest.mark.parametrize("seed", range(24))
def test_random_systems_on_the_host_build(seed, host_engine):
    F.test_random_systems_match_oracle(HOST, seed)


@pytest.mark.parametrize("seed", range(12))
def test_random_crystals_on_the_host_build(seed, host_engine):
    F.test_random_crystals_match_oracle(HOST, seed)


def test_many_crystal_interfaces_on_the_host_build(host_engine):
    F.test_many_crystal_interfaces_use_the_per_surface_march(HOST)


def test_partially_evanescent_interface_on_the_host_build(host_engine):
    F.test_partially_evanescent_crystal_interface_keeps_the_propagating_mode_in_its_slot(HOST)


@pytest.mark.parametrize("seed", range(24))
def test_extreme_systems_on_the_host_build(seed, host_engine, monkeypatch):
    F.test_extreme_systems_match_oracle(HOST, seed, monkeypatch)


@pytest.mark.parametrize("seed", range(16))
def test_extreme_crystal_stacks_on_the_host_build(seed, host_engine):
    F.test_extreme_crystal_stacks_match_oracle(HOST, seed)


# ---- more bodies of the `-m gpu` suite that need nothing but whole-sequence traces and the per-surface calls -------------
import test_gpu_absorbing as A            # noqa: E402
import test_gpu_parity as P               # noqa: E402
import test_gpu_uniform as U              # noqa: E402
import test_gpu_zz_first_contact as Z     # noqa: E402


@pytest.fixture
def no_device_sync(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)


def _params(fn, name):
    for m in getattr(fn, "pytestmark", []):
        if m.name == "parametrize" and m.args[0] == name:
            return list(m.args[1])
    raise KeyError(name)


@pytest.mark.parametrize("name", _params(P.test_per_surface_api_matches_fused, "name"))
def test_per_surface_api_on_the_host_build(name, host_engine, no_device_sync):
    P.test_per_surface_api_matches_fused(name, HOST)


@pytest.mark.parametrize("name", _params(P.test_image_mode_matches_path_mode, "name"))
def test_image_mode_on_the_host_build(name, host_engine, no_device_sync):
    P.test_image_mode_matches_path_mode(name, HOST)


@pytest.mark.parametrize("index", range(len(_params(P.test_side_array_layouts_of_the_explicit_shapes, "surface"))))
def test_side_array_layouts_on_the_host_build(index, host_engine, no_device_sync):
    P.test_side_array_layouts_of_the_explicit_shapes(_params(P.test_side_array_layouts_of_the_explicit_shapes, "surface")[index], HOST)


@pytest.mark.parametrize("body", [P.test_nan_hit_points_on_explicit_shapes_are_dropped_by_the_refraction,
                                  P.test_shape_eval_closed_forms, P.test_shape_eval_biconic_closed_form,
                                  P.test_nonconvergence_mask_flags_capped_newton_rays_and_leaves_valid_alone,
                                  P.test_sharded_crystal_trace_reassembles_to_the_whole_bundle,
                                  U.test_uniform_bundle_through_more_crystals_than_the_fused_walk_parks],
                         ids=lambda f: f.__name__[5:])
def test_single_bodies_on_the_host_build(body, host_engine, no_device_sync):
    body(HOST)


@pytest.mark.parametrize("seed", _params(A.test_hip_vs_oracle_random_absorbing_crystals, "seed"))
def test_random_absorbing_crystals_on_the_host_build(seed, host_engine, no_device_sync):
    A.test_hip_vs_oracle_random_absorbing_crystals(seed, HOST)


@pytest.mark.parametrize("seed", _params(A.test_hip_vs_oracle_random_sequences_that_end_in_an_isotropic_medium, "seed"))
def test_random_sequences_behind_absorbing_crystals_on_the_host_build(seed, host_engine, no_device_sync):
    A.test_hip_vs_oracle_random_sequences_that_end_in_an_isotropic_medium(seed, HOST)


@pytest.mark.parametrize("name", _params(Z.test_surface_step_matches_the_fused_march_and_the_two_calls, "name"))
def test_first_contact_surface_step_body_on_the_host_build(name, host_engine, no_device_sync):
    """the body of the `-m gpu` test of the fused surface step, which no device has run yet (the adapter's surface_step
    is prt_surface_step_rows itself; DeviceSystem.surface_step's own Python -- array placement -- needs a device)"""
    Z.test_surface_step_matches_the_fused_march_and_the_two_calls(name, HOST)


@pytest.mark.parametrize("n", _params(Z.test_surface_step_on_odd_and_tiny_bundles_and_tight_arrays, "n"))
def test_first_contact_surface_step_sizes_on_the_host_build(n, host_engine, no_device_sync):
    Z.test_surface_step_on_odd_and_tiny_bundles_and_tight_arrays(n, HOST)


@pytest.mark.parametrize("eps_kind", ["biaxial", "isotropic"])      # ("uniaxial" goes on into the drop-in layer: device only)
def test_evanescent_modes_as_complex_wave_vectors_on_the_host_build(eps_kind, host_engine, no_device_sync):
    """k_evanescent_fill (the post-pass that reports evanescent modes as complex k) and the E-field stores of the march"""
    U.test_evanescent_modes_come_back_as_complex_wave_vectors(eps_kind, HOST)


@pytest.mark.parametrize("seed", range(24))
def test_random_systems_surface_by_surface_on_the_host_build(seed):
    """the random isotropic systems of the fuzz campaign (tilted frames, explicit shapes, mirrors, apertures; wide
    bundles with misses and total internal reflection) walked surface by surface: prt_surface_step_rows and the pair
    prt_propagate_rows + prt_interact_rows against the oracle on every surface -- masks identical, values to 1e-10"""
    import numpy as np
    import _golden
    from oracle import seqtrace_np as oracle
    rng = np.random.RandomState(1000 + seed)
    recs = F.random_table(rng, int(rng.randint(3, 8)), seed % 2 == 1, seed % 3 != 0, seed % 4 == 3)
    n = int(rng.choice([257, 1000, 1535]))
    x0 = np.vstack((rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), np.full(n, -2.0)))
    u = np.vstack((rng.uniform(-0.12, 0.12, n), rng.uniform(-0.12, 0.12, n), np.ones(n)))
    k0 = u / np.sqrt(np.sum(u ** 2, axis=0))
    e0 = np.cross(k0, np.array([1., 0.3, 0.]), axisa=0, axisb=0).T.copy()
    with np.errstate(all="ignore"):
        out = oracle.trace(recs, x0, k0, e0)
    hs = hostemu.HostSystem(recs)
    pitch = int(hs.lib.prt_recommended_pitch(n)) if seed % 2 else None
    (x, k, valid) = (x0, k0, None)
    (xs, ks, valid_s) = (x0, k0, None)
    for s in range(len(recs)):
        first = dict(e_re=e0) if s == 0 else dict(default_e=False)
        (xh, v) = hs.propagate_rows(s, x, k, valid_in=valid, pitch=pitch, **first)
        (k2, vo) = hs.interact_rows(s, xh, k, valid_in=v, pitch=pitch)
        (xh_f, k2_f, v_f, vo_f) = hs.surface_step_rows(s, xs, ks, valid_in=valid_s, pitch=pitch, **first)
        for (gx, gk, gv, gw) in ((xh, k2, v, vo), (xh_f, k2_f, v_f, vo_f)):
            assert np.array_equal(gv.astype(bool), out[s]["valid"]), (seed, s)
            assert np.array_equal(gw.astype(bool), out[s]["valid_out"]), (seed, s)
            m = out[s]["valid_out"]
            xo = out[s]["x_hit"][:, m]
            assert np.max(np.abs(gx[:, m] - xo) / _golden.relative_scale(xo), initial=0.0) < 1e-10, (seed, s)
            assert np.max(np.abs(gk[:, m] - np.real(out[s]["k_out"])[:, m]), initial=0.0) < 1e-10, (seed, s)
        (x, k, valid) = (xh, k2, vo)
        (xs, ks, valid_s) = (xh_f, k2_f, vo_f)


HOST_MODE_FILES = ["test_gpu_dropin.py", "test_gpu_analysis.py", "test_gpu_absorbing.py", "test_gpu_uniform.py",
                   "test_gpu_parity.py", "test_gpu_fuzz.py", "test_gpu_zz_first_contact.py", "test_gpu_demos.py"]
# what needs the device itself: bundles of 1e6 rays and more (minutes each in the emulation), the placement arena (no
# virtual-memory API on the host), the bench / multi-rank processes
HOST_MODE_DESELECT = ("not full_size and not 1e8 and not scales_exactly and not preflight and not arena and not 1000000 "
                      "and not collimated_host_arrays and not bench")


def test_the_gpu_suite_with_the_products_own_python_on_the_host_build():
    """`PRT_TESTS_ENGINE_ON_HOST=1 pytest -m gpu`: the `-m gpu` tests as they are, with pyrate_amd/engine.py and the drop-in
    layer (raytracer/, dropin.py, the analysis classes) running UNCHANGED on the host build of libprt's sources with CPU
    tensors (tests/hostemu/engine_on_host.py patches the places where engine.py asks torch for a CUDA device or stream;
    tests/conftest.py hands the tests a CPU device).  Eight of the ten GPU test files (the demo scripts included), everything in them that
    does not need the device itself: >= 354 tests, none failing.  This is the Python the build container otherwise never executes
    -- DeviceSystem.surface_step and the symmetric-tensor change of round 6 met it here first."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRT_TESTS_ENGINE_ON_HOST="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "300", "-n", "6",
           "-k", HOST_MODE_DESELECT] + [os.path.join(root, "tests", f) for f in HOST_MODE_FILES]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    m = re.search(r"(\d+) passed", tail)
    assert r.returncode == 0 and m and "failed" not in tail and "error" not in tail, (r.stdout[-3000:], r.stderr[-1500:])
    assert int(m.group(1)) >= 354, tail


def test_the_bench_sweeps_surface_by_surface_on_the_host_build():
    """benchmarks/configs.plugin_sweep -- what bench.py's `plugin` and (guarded, round 6) `surface_step` records time -- with
    engine.py on the host build: the pair of calls per surface and the fused step both end on the fused march's image-plane
    record (masks bit for bit, values to rounding), which is the check measure_plugin applies on the device"""
    from hostemu.engine_on_host import engine_on_host
    with engine_on_host():
        from benchmarks import configs
        from pyrate_amd import _lib
        for fused in (False, True):
            (sweep, ctx) = configs.plugin_sweep(HOST, 5000, placement="torch", fused=fused)
            seen = []
            ctx["tap"] = lambda s, xh, v, k, w: seen.append(s)
            sweep()
            (last, wl, sysd) = (ctx["last"], ctx["wl"], ctx["sysd"])
            assert seen == list(range(wl["S"]))
            ob = sysd.alloc_outputs(wl["n_local"], _lib.MODE_IMAGE, packed_flags=False, placement="torch")
            sysd.trace_into(wl["x0"], wl["k0"], ob, wl["e0"])
            res = sysd.views(ob)
            mk = res.valid_out[0].bool()
            assert torch.equal(last["valid"], res.valid_out[0]) and torch.equal(last["hit"], res.valid[0]) and int(mk.sum()) > 4000
            assert float((last["x"][:, mk] - res.x_hit[0][:, mk]).abs().max()) < 1e-12
            assert float((last["k"][:, mk] - res.k_out[0][:, mk]).abs().max()) < 1e-13


def test_the_default_bench_run_python_on_the_host_build(monkeypatch):
    """benchmarks/configs.run_single_gpu -- everything `python bench.py` does on one GPU between parsing its arguments and
    printing the line: the nine configurations, the guarded tenth (fused surface step), image mode with fused moments, the
    end-to-end drop-in call, verification of what the launches wrote, the compact line -- with engine.py on the host
    build, bundle sizes clamped to a thousand rays, HIP events replaced by a clock, no profiler passes, no arena, no
    1e8-ray point.  Numbers mean nothing here; the Python does: every record is there and verified, the line has the
    contract's keys and fits."""
    import importlib.util
    import json
    import os
    import sys
    import time
    import types
    from hostemu.engine_on_host import engine_on_host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    class Clock(object):
        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self, stream=None):
            self.t = time.perf_counter()

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3
    with engine_on_host():
        monkeypatch.setattr(torch.cuda, "Event", Clock)
        monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", "--traffic", "none", "--no-scaling-point", "--steps", "2",
                                          "--warmup", "1"])
        spec = importlib.util.spec_from_file_location("bench_on_host", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        args = bench.parse_args()
        from benchmarks import configs
        cap = 1000
        (ms, mp_, mi, me) = (configs.measure_single, configs.measure_plugin, configs.measure_image_moments, configs.measure_e2e)
        monkeypatch.setattr(configs, "measure_single",
                            lambda c, a, d, rays, **kw: ms(c, a, d, (40 if c == "aniso_chain" else min(rays, cap)), **kw))
        monkeypatch.setattr(configs, "measure_plugin", lambda a, d, rays, **kw: mp_(a, d, min(rays, cap), **kw))
        monkeypatch.setattr(configs, "measure_image_moments", lambda a, d, rays: mi(a, d, min(rays, cap)))
        monkeypatch.setattr(configs, "measure_e2e", lambda d, **kw: me(d, rays=cap, small_rays=300, calls=2))
        wd = types.SimpleNamespace(stage="", done=lambda: None)
        (recs, scaling_point, e2e, arena_stats, wall) = configs.run_single_gpu(args, HOST, wd, "doublegauss")
        base = {"metric": "ray_surface_ops_per_s", "unit": "ray-surface-ops/s", "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
                "data": "synthetic"}
        line = bench.compact_single(base, recs[0], recs, scaling_point, e2e, arena_stats, {"sha256_16": "host build"}, wall)
    names = [r["name"] for r in recs]
    assert names == ["doublegauss", "asphere", "aniso", "xypoly", "benchmark", "aniso_biaxial", "aniso_chain", "plugin",
                     "surface_step", "image_moments"], names
    assert all(r["verified"]["ok"] for r in recs), [(r["name"], r["verified"]["ok"]) for r in recs]
    text = json.dumps(line)
    assert len(text) < 8192 and set(names) <= set(line["config"]["configs_summary"]) and line["e2e"] and line["verified"]["ok"]
    for key in ("metric", "value", "unit", "ms_per_step", "roofline", "config", "verified", "e2e"):
        assert key in line, key


@pytest.mark.parametrize("exchange,world", [("auto", 2), ("stats", 2), ("auto", 4)])
def test_the_multi_rank_bench_program_with_two_ranks_on_the_host_build(exchange, world):
    """benchmarks/multirank.run_multi -- what `bench.py --gpus N` runs per rank -- with two and with four ranks (one process each, gloo), engine.py
    on the host build: the bundle is split, every rank traces and verifies its shard, the per-step exchange runs; with
    `--exchange auto` and PRT_BENCH_PROBE_DRY=1 the start-up PROBE times both forms of the image-plane gather (in-place
    collective, direct peer writes through shared buffers) and all ranks take the faster one.  The N > 1 program has never run
    on more than one GPU (no node in any round) and its probe was written after round 6's last GPU lease: this is where its
    Python first ran with two ranks."""
    import json
    import os
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "hostemu", "bench_ranks_on_host.py")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PRT_BENCH_PROBE_DRY="1")
        procs.append(subprocess.Popen([sys.executable, script, "--exchange", exchange], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    lines = {}
    for (so, _) in outs:
        for ln in so.splitlines():
            if ln.startswith("RANK "):
                lines[int(ln.split()[1])] = json.loads(ln.split(" ", 2)[2])
    assert all(lines[r] is None for r in range(1, world)) and lines[0] is not None
    line = lines[0]
    assert line["n_gpus"] == world and line["value"] > 0 and line["bytes"] < 8192
    assert line["verified"]["ok"] and line["verified"]["all_ranks_ok"] and line["verified"]["ok_per_rank"] == [True] * world
    assert 3500 < line["config"]["rays_total"] <= 4000
    if exchange == "auto":
        probe = line["config"]["exchange_probe_ms"]
        assert set(probe) == {"gather", "gather-direct"} and probe["gather"] > 0
        chosen = line["config"]["exchange"]
        assert chosen in probe and probe[chosen] == min(v for v in probe.values() if v is not None)
    else:
        assert line["config"]["exchange"] == "stats" and line["config"]["exchange_probe_ms"] is None
