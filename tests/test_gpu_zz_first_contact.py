"""``-m gpu`` tests of what round 6 wrote AFTER its last GPU lease (the pool closed to this repository for the rest of the
round): they have run on the HOST build of the kernels' sources (tests/test_hostemu.py: same C ABI, same launch-site logic,
oracle and golden vectors, ASan / UBSan) but never on a device.  The file sorts last on purpose: the driver runs the
suite with -x, and a first contact that goes wrong here cannot hide the result of any test that has been green on a GPU
before."""
import numpy as np
import pytest
import torch

import _golden
from test_gpu_parity import engine_trace
from test_gpu_perf import _free_port

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["double_gauss_wide", "tilted_frames", "xypoly_field5", "mirrors", "asphere_strong_field5",
                                  "biconic_field5", "doublet_clipped"])
def test_surface_step_matches_the_fused_march_and_the_two_calls(name, gpu_device):
    """prt_surface_step_rows (the loop body of OpticalElement.seqtrace, optical_element.py:336-375, as one launch per
    surface) against the fused march (masks bit for bit, values to rounding) and against prt_propagate_rows +
    prt_interact_rows on the same arrays"""
    from pyrate_amd import engine
    case = _golden.load_case(name)
    (sysd, res) = engine_trace(case, gpu_device)
    x = engine.to_device_rays(case.x0, gpu_device)
    k = engine.to_device_rays(case.k0, gpu_device)
    e = np.asarray(case.E0)
    e_re = engine.to_device_rays(e.real, gpu_device)
    e_im = engine.to_device_rays(e.imag, gpu_device) if np.iscomplexobj(e) else None
    valid = None
    for s in range(case.n_surfaces):
        first = dict(e_re=e_re, e_im=e_im) if s == 0 else dict(default_e=False)
        (xh, k2, v, vo) = sysd.surface_step(s, x, k, valid_in=valid, **first)
        (xh_b, v_b) = sysd.propagate(s, x, k, valid_in=valid, **first)
        (k2_b, _d, vo_b, _, _) = sysd.interact(s, xh_b, k, valid_in=v_b)
        assert torch.equal(v, res.valid[s]) and torch.equal(vo, res.valid_out[s])
        assert torch.equal(v, v_b) and torch.equal(vo, vo_b)
        m = vo.bool()
        assert torch.allclose(xh[:, m], res.x_hit[s][:, m], rtol=0, atol=1e-11)
        assert torch.allclose(k2[:, m], res.k_out[s][:, m], rtol=0, atol=1e-12)
        assert torch.allclose(xh[:, m], xh_b[:, m], rtol=0, atol=1e-11) and torch.allclose(k2[:, m], k2_b[:, m], rtol=0, atol=1e-12)
        (x, k, valid) = (xh, k2, vo)


@pytest.mark.parametrize("n", [1, 2, 3, 129, 1000, 1001])
def test_surface_step_on_odd_and_tiny_bundles_and_tight_arrays(n, gpu_device):
    """the two-rays-per-thread form with an odd ray count, the unaligned fall-back (tight arrays of odd length), and the
    nonconv mask; crystals are refused"""
    from pyrate_amd import engine, systems
    recs = systems.asphere_records(coefficients=(1e-3, -1e-6, 1e-8), curv=-1. / 30., cc=-1.5)
    (o, k, e0) = systems.double_gauss_bundle(max(n, 20), rpup=9.0, z0=-5.0, field_deg=5.0)
    (o, k, e0) = [np.ascontiguousarray(a[:, :n]) for a in (o, k, e0)]
    sysd = engine.DeviceSystem(recs, 0)
    res = sysd.trace(*[engine.to_device_rays(a, gpu_device) for a in (o, k, e0)], want_nonconv=True)
    for pitched in (True, False):
        (x, kk, ee) = [engine.to_device_rays(a, gpu_device, pitched=pitched) for a in (o, k, e0)]
        valid = None
        for s in range(len(recs)):
            first = dict(e_re=ee) if s == 0 else dict(default_e=False)
            (xh, k2, v, vo, nc) = sysd.surface_step(s, x, kk, valid_in=valid, want_nonconv=True, **first)
            assert torch.equal(v, res.valid[s]) and torch.equal(vo, res.valid_out[s]) and torch.equal(nc, res.nonconv[s])
            m = vo.bool()
            assert torch.allclose(xh[:, m], res.x_hit[s][:, m], rtol=0, atol=1e-11)
            assert torch.allclose(k2[:, m], res.k_out[s][:, m], rtol=0, atol=1e-12)
            (x, kk, valid) = (xh, k2, vo)
    case = _golden.load_case("aniso_doublet_uniaxial")
    crystal = engine.DeviceSystem(case.table, 0)
    s_c = [s for (s, r) in enumerate(case.table) if r["material"]["type"] == "anisotropic"][0]
    with pytest.raises(ValueError):
        crystal.surface_step(s_c, x, kk)



def test_scale_preflight_dry_run_and_the_exchange_probe(gpu_device, tmp_path):
    """benchmarks/scale_preflight.sh -- the script for the first contact with a multi-GPU node -- dry-run on this one
    GPU with gloo: N = 1 and N = 2, both exchanges, one compact line per run, every rank verified; and the start-up
    PROBE that chooses between the in-place all-gather and the direct peer writes (bench.py --exchange auto), run for
    real with two ranks on this GPU (PRT_BENCH_PROBE_DRY=1: gloo instead of RCCL, which refuses two ranks on one
    device): both forms are timed, one is chosen, the line says which."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRT_ARENA_BUDGET_GIB="24", PRT_BENCH_WATCHDOG="300")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = str(tmp_path / "scale")
    r = subprocess.run(["bash", os.path.join(root, "benchmarks", "scale_preflight.sh"), out, "2", "--backend", "gloo",
                        "--rays-total", "2000000", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=1200, cwd=root)
    rows = [json.loads(l) for l in open(os.path.join(out, "summary.jsonl")) if l.strip()]
    assert len(rows) == 4, (r.stdout[-800:], r.stderr[-800:])
    for row in rows:
        assert not row.get("error"), row
        assert row["value"] > 0 and all(row["ok_per_rank"]) and len(row["ok_per_rank"]) == row["n_gpus"]
        assert row["scaling_point"]["ok"] and abs(row["scaling_point"]["rays"] - 2e6) < 0.02 * 2e6
        assert row["exchange"] == ("gather" if row["exchange_asked"] == "auto" else "stats")
        assert row["ms_total"] > 0 and row["ms_trace"] > 0
    n1 = json.loads(open(os.path.join(out, "n1_default.json")).read().strip().splitlines()[-1])
    assert n1["n_gpus"] == 1 and n1["verified"]["ok"] and len(json.dumps(n1)) < 8192
    # the probe itself, two ranks on this GPU
    env2 = dict(env, PRT_BENCH_PROBE_DRY="1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rays-total",
                        "2000000", "--steps", "3", "--warmup", "1"], env=env2, capture_output=True, text=True, timeout=900,
                       cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    line = json.loads(lines[0])
    probe = line["config"]["exchange_probe_ms"]
    assert set(probe) == {"gather", "gather-direct"} and probe["gather"] > 0
    chosen = line["config"]["exchange"]
    assert chosen in ("gather", "gather-direct") and probe[chosen] == min(v for v in probe.values() if v is not None)
    assert line["verified"]["ok"] and line["verified"]["all_ranks_ok"] and "error" not in line


def test_a_rotated_symmetric_tensor_is_used_as_its_symmetric_part(gpu_device):
    """prt_system_create stores a real tensor that is symmetric up to rounding (R diag R^T) as its symmetric part, which
    opens the crystal solver's cheapest route (flux from the adjugate of W, prt_aniso.h) to it -- bench.py's aniso_biaxial
    tensors: the trace equals the trace with the explicitly symmetrised tensors bit for bit and agrees with the oracle on
    the caller's tensors (host build: tests/test_hostemu.py, same assertion)"""
    from pyrate_amd import engine, systems
    from oracle import seqtrace_np as oracle
    def rot(ax, ay, az):
        (ca, sa, cb, sb, cg, sg) = (np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az))
        rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
        ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
        rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
        return rz.dot(ry).dot(rx)
    (r1, r2) = (rot(0.4, 0.25, -0.3), rot(-0.2, 0.35, 0.15))
    (e1, e2) = (r1.dot(np.diag([1.55 ** 2, 1.60 ** 2, 1.68 ** 2])).dot(r1.T), r2.dot(np.diag([1.62 ** 2, 1.66 ** 2, 1.71 ** 2])).dot(r2.T))
    (o, k, e0) = systems.double_gauss_bundle(600, rpup=11.43, z0=-5.0, field_deg=2.0)
    rays = [engine.to_device_rays(a, gpu_device) for a in (o, k, e0)]
    raw = engine.DeviceSystem(systems.aniso_doublet_records(e1, e2), 0).trace(*rays)
    sym = engine.DeviceSystem(systems.aniso_doublet_records(0.5 * (e1 + e1.T), 0.5 * (e2 + e2.T)), 0).trace(*rays)
    with np.errstate(all="ignore"):
        out = oracle.trace(systems.aniso_doublet_records(e1, e2), o, k, e0)
    for s in range(len(out)):
        assert torch.equal(raw.valid_out[s], sym.valid_out[s])
        assert np.array_equal(raw.x_hit[s].cpu().numpy(), sym.x_hit[s].cpu().numpy(), equal_nan=True)
        assert np.array_equal(raw.k_out[s].cpu().numpy(), sym.k_out[s].cpu().numpy(), equal_nan=True)
        ko = np.real(out[s]["k_out"])
        fin = np.all(np.isfinite(ko), axis=0)
        assert np.abs(raw.k_out[s].cpu().numpy()[:, fin] - ko[:, fin]).max() < 1e-12
