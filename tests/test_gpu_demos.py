"""smoke tests of the demo scripts (the reference's tests/smoke_test.py:35-109 pattern:
run the demo, it must not raise)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_smoke_doublet(gpu_device, capsys):
    from demos import demo_doublet
    r = demo_doublet.main()
    assert len(r.raybundles) == 7
    assert "20 rays reach the image plane" in capsys.readouterr().out


def test_smoke_doublegauss(gpu_device):
    from demos import demo_doublegauss
    demo_doublegauss.main(2000)


def test_smoke_asphere(gpu_device, capsys):
    from demos import demo_asphere
    demo_asphere.main(121)
    out = capsys.readouterr().out
    assert "asphere:" in out


def test_smoke_anisotropic_doublet(gpu_device, capsys):
    from demos import demo_anisotropic_doublet
    demo_anisotropic_doublet.main(11)
    assert capsys.readouterr().out.count("4 ray paths") == 2


def test_smoke_benchmark(gpu_device, capsys):
    from demos import demo_benchmark
    demo_benchmark.main(20000)
    assert "ray-surface-operations per second" in capsys.readouterr().out


def test_smoke_optimize_asphere(gpu_device):
    """parameter changes through FloatVariable.set_value reach the device table on the next
    seqtrace; the merit function decreases"""
    from demos import demo_optimize_asphere
    (m0, m1) = demo_optimize_asphere.main(maxiter=150)
    assert m1 < 0.5 * m0
    # same loop with the merit taken from OpticalSystem.image_moments (no path, 7 doubles back)
    (f0, f1) = demo_optimize_asphere.main(maxiter=150, fast=True)
    assert abs(f0 - m0) < 1e-9 * m0 and f1 < 0.5 * f0


def test_smoke_zmx(gpu_device, capsys):
    """demos/demo_zmx.py on the reference's own lenssystem.ZMX: four object-height fields"""
    from demos import demo_zmx
    res = demo_zmx.main(nrays=500)
    assert len(res) == 4 and all(n > 0 for (n, _) in res)
    assert capsys.readouterr().out.count("RMS spot") == 4


def test_smoke_spd(gpu_device, tmp_path, capsys):
    """demos/demo_spd.py on a synthetic WinLens file of the double Gauss: the traced chief-ray
    heights follow the file's paraxial image heights"""
    import systems_zoo as zoo
    from demos import demo_spd
    f = str(tmp_path / "dg.spd")
    zoo.synthetic_double_gauss_spd(f)
    res = demo_spd.main(f, nrays=500)
    assert len(res) == 9
    for (fpy, cy, rms) in res:
        assert rms < 0.2
    out = capsys.readouterr().out
    assert "efl 115.74" in out


def test_smoke_prism(gpu_device, capsys):
    from demos import demo_prism
    out = demo_prism.main()
    assert out["blue"] != out["red"] and "dispersion" in capsys.readouterr().out


def test_smoke_anisotropic_mirror(gpu_device, capsys):
    from demos import demo_anisotropic_mirror
    (forks, stacked) = demo_anisotropic_mirror.main(10)
    assert len(forks) == 4 and stacked.raybundles[-1].num_rays == 4 * forks[0].raybundles[-1].num_rays


def test_bench_contract_line(gpu_device):
    """bench.py prints ONE JSON line, last on stdout, with the keys the driver reads (small bundle)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rays", "300000", "--steps", "4",
                          "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["dtype"] == "f64" and "workload" in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["value"] > 1e8 and d["cpu_baseline"]["kind"] == "port"


def test_smoke_rainbow(gpu_device, capsys):
    """water droplet, one internal reflection through one surface visited twice: the whole fan comes back, 24-29
    degrees from the antisolar direction, the two colours half a degree apart"""
    from demos import demo_rainbow
    bow = demo_rainbow.main(11)
    assert 22.0 < bow["red"] < 32.0 and 22.0 < bow["blue"] < 32.0 and 0.2 < abs(bow["red"] - bow["blue"]) < 1.0
    assert capsys.readouterr().out.count("11 of 11 rays come back") == 2


def test_smoke_mirrors(gpu_device, capsys):
    """three tilted spherical mirrors + an off-axis paraboloid: every ray of every field reaches the first two image
    planes; the axial field focuses best behind the paraboloid"""
    from demos import demo_mirrors
    out = demo_mirrors.main(300)
    n0 = out[(0.0, "image1")][0]
    assert n0 > 250 and all(out[(f, "image1")][0] == n0 for f in (0.0, 0.5, -0.5))
    assert out[(0.0, "image2")][1] < out[(0.0, "image1")][1]
    assert capsys.readouterr().out.count("mirrors, field") == 3


def test_smoke_hud(gpu_device, capsys):
    """free-form prism with biconic faces, one of them used in transmission and in reflection: the axial fan gets through"""
    from demos import demo_hud
    out = demo_hud.main(9)
    assert out[0.0] == 9 and out[15.0] > 0 and out[-15.0] > 0
    assert capsys.readouterr().out.count("hud, field") == 3


def test_smoke_anisotropic_ord_eo(gpu_device, capsys):
    """uniaxial plate, divergent fan: splitup forks into the ordinary and the extraordinary path, which separate
    off the axis; without splitup the doubled rays travel in one path"""
    from demos import demo_anisotropic_ord_eo
    r = demo_anisotropic_ord_eo.main(10)
    assert r["paths"] == 2 and 0.01 < r["gap_max"] < 1.0 and r["rays_one_path"] == 2 * r["rays_in"]
    assert "splitted: True" in capsys.readouterr().out


def test_smoke_tilted_image(gpu_device, capsys):
    """the image frame is tilted after the system was built (set_value + update): the next trace sees it -- the fans
    land elsewhere on the tilted surface, the off-axis fans no longer symmetric"""
    from demos import demo_tilted_image
    (before, after) = demo_tilted_image.main(11)
    assert abs(before[1][0] + before[2][0]) < 1e-9 and abs(before[0][0]) < 1e-9          # symmetric before the tilt
    assert abs(after[1][0] - before[1][0]) > 1e-3 and abs(after[1][0] + after[2][0]) > 1e-6
    assert all(a[2] == 11 for a in after)
    assert capsys.readouterr().out.count("tilted image, field") == 3
