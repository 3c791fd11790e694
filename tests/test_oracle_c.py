"""Pins the C / OpenMP oracle (oracle/seqtrace_c.c, the multi-core CPU baseline of bench.py)
against the golden vectors from the real reference and against the NumPy oracle.  CPU only."""
import numpy as np
import pytest

import _golden
from oracle import seqtrace_c, seqtrace_np


@pytest.mark.parametrize("name", _golden.ISO_CASES)
def test_c_oracle_vs_reference(name):
    case = _golden.load_case(name)
    out = seqtrace_c.trace(case.table, case.x0, case.k0, case.E0)
    res = _golden.compare_dense_to_reference(case, _golden.dense_from_oracle(out), rtol_x=1e-12, atol_k=1e-12)
    assert res["n_compared"] > 0


def test_c_oracle_matches_numpy_oracle_dense_and_threads():
    case = _golden.load_case("double_gauss_wide")
    a = seqtrace_np.trace(case.table, case.x0, case.k0, case.E0)
    for nt in (1, 3):
        (x_hit, k_out, valid, valid_out, used) = seqtrace_c.trace_arrays(case.table, case.x0, case.k0, case.E0,
                                                                         nthreads=nt)
        assert used == nt
        for s in range(case.n_surfaces):
            assert np.array_equal(valid[s].astype(bool), a[s]["valid"])
            assert np.array_equal(valid_out[s].astype(bool), a[s]["valid_out"])
            v = a[s]["valid_out"]
            assert np.max(np.abs(x_hit[s][:, v] - a[s]["x_hit"][:, v])) < 1e-12
            assert np.max(np.abs(k_out[s][:, v] - a[s]["k_out"][:, v])) < 1e-13


def test_c_oracle_rejects_out_of_scope_tables():
    with pytest.raises(ValueError):
        seqtrace_c.flat_table(_golden.load_case("asphere_mild_axis").table)
