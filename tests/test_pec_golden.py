"""Pins the CPU oracle's PEC field boundary (first "next" row of SURVEY.md 8(f)) against the reference's
own golden checksums and analysis for Examples/Tests/pec/inputs_test_3d_pec_field.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import pec_case

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pec_run(oracle):
    sim = pec_case.make_sim(oracle)
    sim.evolve(pec_case.MAX_STEP)
    return sim


def test_golden_checksums(oracle, pec_run):
    gold = json.load(open(os.path.join(HERE, "golden", "pec_field_3d_checksums.json")))
    for name, want in gold["checksums"]["lev=0"].items():
        v = pec_run.field_view(name)
        got = oracle.cell_centered_abs_sum(C.byref(v))
        print(f"{name}: got {got:.16e} want {want:.16e} rel {abs(got - want) / want:.2e}")
        assert abs(got - want) / want < gold["rtol"]


def test_reference_analysis(pec_run):
    """Examples/Tests/pec/analysis_pec.py: standing wave of twice the incident amplitude (1 %), Ey = 0 on the walls."""
    ey = pec_run.field_valid("Ey")
    e_th = 2.0 * pec_case.EY_IN
    assert abs(ey.max() - e_th) / e_th < 0.01
    assert abs(ey.min() + e_th) / e_th < 0.01
    assert np.all(ey[:, :, 0] == 0.0) and np.all(ey[:, :, -1] == 0.0)


def test_host_layer_reproduces_the_oracle_stepper(oracle, pec_run):
    """The product's C++ host layer (WarpX::EvolveB/EvolveE -> ApplyB/EfieldBoundary, BrickComm with a
    non-periodic direction) on the CPU kernels against the independent oracle stepper: bit for bit."""
    from tests.oracle_lib import load_host_cpu
    sim = pec_case.make_sim(load_host_cpu())
    sim.evolve(pec_case.MAX_STEP)
    for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz"):
        assert np.array_equal(sim.field(name), pec_run.field(name)), name
