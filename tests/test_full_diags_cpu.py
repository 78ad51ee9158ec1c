"""Full diagnostics of the host layer (warpx_amd/csrc/host/FullDiagnostics.hpp): `<diag>.diag_type = Full` with
format = plotfile -- the plotfiles a WarpX run leaves under diags/ -- on the CPU build of the host layer: which steps are
written (intervals, the dump before the first step, the forced dump of the last one, m_already_done), the file names
(amrex::Concatenate(file_prefix, istep, file_min_digits)), the selection of fields and species, and the content read
back with the strict plotfile reader of tests/test_plotfile_cpu.py."""
import os

import numpy as np
import pytest

from tests.oracle_lib import load_host_cpu
from tests.test_plotfile_cpu import checksum_of, read_plotfile
from tests.test_reduced_diags_cpu import two_species_sim
from warpx_amd import _capi
from warpx_amd.sim import WarpXSim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECK = os.path.join(ROOT, "tests", "decks", "langmuir_multi_3d.inputs")


@pytest.fixture(scope="module")
def host_cpu():
    return load_host_cpu()


def test_which_steps_are_written(host_cpu, tmp_path):
    sim, ids = two_species_sim(host_cpu)
    sim.add_full_diag("diag1", "2", str(tmp_path / "diags" / "diag1"))
    sim.add_full_diag("slim", "3:", str(tmp_path / "x" / "y" / "slim"), file_min_digits=4, fields=["Ez", "rho", "F"],
                      write_species=False, dump_last_timestep=False)
    with pytest.raises(_capi.WxaError, match="defined twice"):
        sim.add_full_diag("slim", "1")
    sim.evolve(3)
    sim.evolve(2)
    assert sorted(os.listdir(tmp_path / "diags")) == ["diag1000000", "diag1000002", "diag1000004"]
    sim.flush_diags_last_timestep()      # a run built through the API ends when its caller says so
    sim.flush_diags_last_timestep()      # ... once
    assert sorted(os.listdir(tmp_path / "diags")) == ["diag1000000", "diag1000002", "diag1000004", "diag1000005"]
    assert sorted(os.listdir(tmp_path / "x" / "y")) == ["slim0003", "slim0004", "slim0005"]
    last = read_plotfile(str(tmp_path / "diags" / "diag1000005"))
    assert last["names"] == ["Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"] and last["step"] == 5
    assert sorted(last["species"]) == ["species0", "species1"]
    direct = sim.checksum()
    got = checksum_of(last)
    for group, vals in got.items():
        for k, v in vals.items():
            assert abs(v - direct[group][k]) <= 1e-12 * max(abs(direct[group][k]), 1e-300), (group, k)
    first = read_plotfile(str(tmp_path / "diags" / "diag1000000"))
    assert first["step"] == 0 and first["time"] == 0.0 and np.all(first["fields"]["Ex"] == 0.0)
    slim = read_plotfile(str(tmp_path / "x" / "y" / "slim0005"))
    assert slim["names"] == ["Ez", "rho"] and slim["species"] == {}      # F (div E cleaning): left out with a warning
    assert np.array_equal(slim["fields"]["Ez"], last["fields"]["Ez"])
    assert abs(np.sum(np.abs(slim["fields"]["rho"])) - direct["lev=0"]["rho"]) <= 1e-12 * direct["lev=0"]["rho"]
    sim.close()


def test_a_deck_writes_the_reference_s_output_tree(host_cpu, tmp_path):
    """diagnostics.diags_names of a deck: plotfiles at the intervals and at max_step, reduced diagnostics next to them
    -- what `python -m warpx_amd.run deck` leaves behind, as the reference's executable does."""
    prefix = str(tmp_path / "diags" / "diag1")
    over = ["warpx_amd.write_diagnostics=1", "my_constants.nx=16", "max_step=6", "my_constants.every=4", "diag1.intervals=0:nx*10:every", f"diag1.file_prefix={prefix}",
            "diag1.fields_to_plot=Ex Ey Ez jx rho", "diag1.species=positrons",
            "warpx.reduced_diags_names=EF", "EF.type=FieldEnergy", "EF.intervals=3", f"EF.path={tmp_path}/diags/reducedfiles/"]
    sim = WarpXSim.from_inputs(host_cpu, DECK, overrides=over)
    sim.evolve(4)
    sim.evolve(2)      # reaches max_step: the forced dump of the last time step
    direct = sim.checksum()
    sim.close()
    assert sorted(os.listdir(tmp_path / "diags")) == ["diag1000000", "diag1000004", "diag1000006", "reducedfiles"]
    pf = read_plotfile(prefix + "000006")
    assert pf["names"] == ["Ex", "Ey", "Ez", "jx", "rho"] and list(pf["species"]) == ["positrons"] and pf["step"] == 6
    got = checksum_of(pf)
    for group, vals in got.items():
        for k, v in vals.items():
            assert abs(v - direct[group][k]) <= 1e-12 * max(abs(direct[group][k]), 1e-300), (group, k)
    rows = np.atleast_2d(np.genfromtxt(str(tmp_path / "diags" / "reducedfiles" / "EF.txt")))
    assert list(rows[:, 0]) == [0, 3, 6]
    # formats this library does not write, and the switch the test-suite runs the decks with
    for extra in (["diag1.format=openpmd"], ["warpx_amd.write_diagnostics=0"], ["diag1.diag_type=TimeAveraged"]):
        out = tmp_path / ("none_" + extra[0].split("=")[1])
        o2 = [v for v in over if not v.startswith("diag1.file_prefix") and not v.startswith("EF.")] + \
             [f"diag1.file_prefix={out}/diag1", "warpx.reduced_diags_names="] + extra
        sim = WarpXSim.from_inputs(host_cpu, DECK, overrides=o2)
        sim.evolve(sim.max_step)
        sim.close()
        assert not os.path.exists(out), extra
    with pytest.raises(_capi.WxaError, match="names no field"):
        WarpXSim.from_inputs(host_cpu, DECK, overrides=over + ["diag1.fields_to_plot=none"])
    with pytest.raises(_capi.WxaError, match="unknown species"):
        WarpXSim.from_inputs(host_cpu, DECK, overrides=over + ["diag1.species=muons"])
    with pytest.raises(_capi.WxaError, match="intervals must be set"):
        WarpXSim.from_inputs(host_cpu, DECK, overrides=["warpx_amd.write_diagnostics=1", "diagnostics.diags_names=d2", "d2.diag_type=Full"])


def test_div_e_and_part_per_cell_in_the_plotfile(host_cpu, tmp_path):
    """fields_to_plot = ... divE part_per_cell (DivEFunctor.cpp, PartPerCellFunctor.cpp; round 4): part_per_cell counts
    every macro-particle once in the cell that holds it; div E on the nodes, averaged to the cell centres, obeys the
    discrete Gauss law of the charge-conserving deposition -- eps0 div E - rho is a constant of the run (the two species
    start displaced from each other with E = 0, so the constant is not zero), to round-off, cell by cell."""
    from scipy.constants import epsilon_0
    sim, ids = two_species_sim(host_cpu)
    fields = ["Ex", "rho", "divE", "part_per_cell"]
    sim.add_full_diag("g", "0,6", str(tmp_path / "g"), fields=fields, write_species=False, dump_last_timestep=False)
    sim.evolve(6)
    a = read_plotfile(str(tmp_path / "g000000"))
    b = read_plotfile(str(tmp_path / "g000006"))
    assert a["names"] == fields and b["names"] == fields
    n = sum(sim.particle_view(i).np for i in ids)
    for pf in (a, b):
        ppc = pf["fields"]["part_per_cell"]
        assert np.all(ppc == np.round(ppc)) and ppc.min() >= 0 and ppc.sum() == n
    ra = epsilon_0 * a["fields"]["divE"] - a["fields"]["rho"]
    rb = epsilon_0 * b["fields"]["divE"] - b["fields"]["rho"]
    scale = np.max(np.abs(b["fields"]["rho"]))
    assert np.max(np.abs(b["fields"]["divE"])) * epsilon_0 > 1e-3 * scale          # the wave has built up a field
    assert np.max(np.abs(ra - rb)) < 1e-9 * scale, (np.max(np.abs(ra - rb)), scale)
    assert np.all(a["fields"]["divE"] == 0.0)                                       # E = 0 at the start
    sim.close()
