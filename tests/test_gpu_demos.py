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
