"""Prescription importers: the object graph built from a file must flatten to the
same surface table as the one the reference's own parser builds from that file."""
import json
import os

import numpy as np
import pytest

import _golden
import systems_zoo as zoo
from pyrate_amd.surface_table import UnsupportedError, flatten_sequence
from test_host_logic import _assert_tables_equal

ZMX = os.path.join(_golden.GOLDEN_DIR, "lenssystem.ZMX")


@pytest.fixture(scope="module")
def api():
    return zoo.mirror_api()


def _zmx_system(api):
    from pyrate_amd.raytracer.io.zmx import ZMXParser
    zp = ZMXParser(ZMX, name="zmx")
    lc = api.LocalCoordinates.p(name="tmp")
    return (zp,) + tuple(zp.create_optical_system({"BK7": api.ConstantIndexGlass.p(lc, 1.5168)}))


def test_zmx_system_table_equals_reference_parser(api):
    (zp, s, seq) = _zmx_system(api)
    case = _golden.load_case("zmx_lenssystem")
    (recs, lengths) = flatten_sequence(s, seq, case.wave)
    assert lengths == case.elem_lengths == [14]
    _assert_tables_equal(recs, case.table)
    # coordinate breaks are shapeless surfaces in tilted frames; the stop is flagged
    assert [n for (n, o) in seq[0][1] if o.get("is_stop")] == ["surf5"]
    assert recs[8]["shape"]["type"] == "conic" and recs[8]["shape"]["curv"] == 0.0


def test_zmx_field_and_initial_bundles_equal_reference_parser(api):
    (zp, _, _) = _zmx_system(api)
    ref = json.load(open(os.path.join(_golden.GOLDEN_DIR, "zmx_lenssystem_field.json")))
    fd = zp.read_field()
    assert json.loads(json.dumps(fd)) == ref["field"]
    assert json.loads(json.dumps(zp.create_initial_bundle())) == ref["bundles"]
    assert list(zp.read_name_and_notes()) == ref["name_notes"]


def test_zmx_without_materials_refuses_like_the_reference(api):
    from pyrate_amd.raytracer.io.zmx import ZMXParser
    (s, seq) = ZMXParser(ZMX).create_optical_system()
    assert s is None and seq == [("zmxelem", [])]            # zmx.py:594-596


def test_zmx_text_variants(api, tmp_path):
    """ASCII file, mirror, model glass, apertures, biconic, unsupported sag types"""
    from pyrate_amd.raytracer.io import zmx
    text = "\n".join([
        "VERS 1", "NAME folded test", "NOTE 0 hello",
        "SURF 0", "  TYPE STANDARD", "  CURV 0.0", "  DISZ INFINITY",
        "SURF 1", "  STOP", "  TYPE STANDARD", "  CURV 0.01", "  CONI -1.0", "  DISZ 5.0",
        "  GLAS MYGLASS 1 0 1.6 0.0", "  CLAP 0 4.0",
        "SURF 2", "  TYPE BICONICX", "  CURV -0.02", "  PARM 1 80.0", "  PARM 2 -0.5", "  DISZ 10.0",
        "  GLAS DISPERSIVE 1 0 1.5168 64.17 0 0 0 0", "  SQAP 3.0 2.0", "  OBDC 0.1 -0.2",
        "SURF 3", "  TYPE STANDARD", "  CURV 0.0", "  DISZ -10.0", "  GLAS MIRROR 0 0 1.5 40",
        "SURF 4", "  TYPE STANDARD", "  CURV 0.0", "  DISZ 0"]) + "\n"
    f = tmp_path / "t.zmx"
    f.write_text(text)
    zp = zmx.ZMXParser(str(f))
    assert zp.read_name_and_notes() == ("folded test", ["hello"])
    (s, seq) = zp.create_optical_system()
    (recs, _) = flatten_sequence(s, seq, 0.5876e-3)
    assert [r["shape"]["type"] for r in recs] == ["conic", "biconic", "conic", "conic"]
    assert recs[0]["shape"]["cc"] == -1.0 and recs[0]["aperture"]["type"] == "circular"
    assert recs[0]["material"]["n"] == pytest.approx(1.6)
    assert recs[1]["aperture"]["type"] == "rectangular" and recs[1]["aperture"]["width"] == pytest.approx(6.0)
    assert recs[1]["material"]["n"] == pytest.approx(1.5168, abs=2e-6)      # Conrady through nd at the d line
    assert recs[2]["interaction"] == "mirror" and recs[2]["material"]["n"] == recs[1]["material"]["n"]
    # normal-line Conrady model reproduces nF - nC = (nd - 1) / vd
    mat = s.elements["zmxelem"].materials[[k for k in s.elements["zmxelem"].materials if "surf2" in k][0]]
    d = mat.get_optical_index(None, 0.4861327e-3) - mat.get_optical_index(None, 0.6562725e-3)
    assert d == pytest.approx((1.5168 - 1) / 64.17, rel=2e-3)
    f.write_text(text.replace("TYPE BICONICX", "TYPE GRID_SAG"))
    with pytest.raises(UnsupportedError):
        zmx.ZMXParser(str(f)).create_optical_system()
