"""The random-system campaigns of the `-m gpu` suite (tests/test_gpu_fuzz.py), run on the HOST build of libprt's sources
(tests/hostemu): the bodies of the GPU tests are called as they are, with ``pyrate_amd.engine.DeviceSystem`` /
``to_device_rays`` swapped (pytest monkeypatch, this test only) for the host build's adapter.  CPU only.  What it covers that
the golden cases do not: Newton on harsh shapes (misses, NaN domains, the iteration cap), random crystals incl. the
wave-uniform fall-backs of the biaxial solver (votes across the lanes of a wave), mirrors inside crystals, partially
evanescent interfaces, the per-surface march through more crystal interfaces than the fused walk parks."""
import pytest
import torch

hostemu = pytest.importorskip("hostemu")
try:
    hostemu.build()
except Exception as exc:
    pytest.skip("no host build of libprt: %s" % exc, allow_module_level=True)

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
