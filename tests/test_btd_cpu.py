"""Back-transformed diagnostics, fields and particles (host/BTDiagnostics.hpp against Source/Diagnostics/BTDiagnostics.cpp,
ComputeDiagFunctors/BackTransformFunctor.cpp and BackTransformParticleFunctor.cpp).  No golden file of the reference pins it (its only 3-D BTD deck draws a
random beam) and AMReX's slice interpolation is not on disk, so the pins are: the independent restatement in the oracle
stepper, and the physics a back-transformation must deliver -- a laser emitted in a gamma = 2 frame is found in the lab
frame at the lab position, with the lab wavelength and the lab amplitude."""
import os

import numpy as np
import pytest

from tests import pec_case
from warpx_amd import plasma
from warpx_amd.sim import WarpXSim

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host_cpu():
    from tests.oracle_lib import load_host_cpu
    return load_host_cpu()


def test_btd_host_layer_against_the_oracle_stepper(oracle, host_cpu):
    """Config 5 in small (gamma = 5, window at c, CKC, Vay, NCI, antenna, injected plasma) with three lab-frame snapshots:
    the host layer's BTDiagnostics and the oracle stepper's own give the same slices, component by component."""
    steps = 50
    out, parts = [], []
    for lib in (host_cpu, oracle):
        sim, e = pec_case.make_boosted_lwfa_sim(lib)
        dt_snap = 12 * sim.dt * pec_case.BOOST_GAMMA   # a new snapshot plane enters the boosted domain every ~12 steps
        sim.add_btd(3, dt_snap, buffer_size=32)
        sim.evolve(steps)
        info = [sim.btd_info(i) for i in range(3)]
        data = [{c: sim.btd_snapshot(i, c) for c in WarpXSim.BTD_COMPONENTS} for i in range(3)]
        parts.append([sim.btd_particles(i, e) for i in range(3)])
        sim.close()
        out.append((info, data))
    (ih, dh), (io, do) = out
    # the electrons each snapshot's plane has met (BackTransformParticleFunctor): the same particles, the same lab-frame values
    for i in range(3):
        a, b = parts[0][i], parts[1][i]
        assert a.shape == b.shape and (i > 0 or a.shape[1] > 100)
        # the host layer re-sorts its tile by cell, the oracle keeps injection order: same set, different sequence
        a, b = (q[:, np.lexsort((q[2], np.round(q[1] / 1e-10), np.round(q[0] / 1e-10)))] for q in (a, b))
        for row in range(7):
            assert np.max(np.abs(a[row] - b[row])) <= 1e-9 * max(np.max(np.abs(b[row])), 1e-300), (i, row)
    for i in range(3):
        assert ih[i]["n"] == io[i]["n"] and ih[i]["n"][2] % 32 == 0
        assert ih[i]["slices"] == io[i]["slices"] and ih[i]["full"] == io[i]["full"]
        assert abs(ih[i]["t_lab"] - io[i]["t_lab"]) <= 1e-15 * abs(io[i]["t_lab"]) + 1e-30
        for a, b in zip(ih[i]["z_lab"], io[i]["z_lab"]):
            assert abs(a - b) <= 1e-12 * max(abs(b), 1e-6)
        for c in WarpXSim.BTD_COMPONENTS:
            a, b = dh[i][c], do[i][c]
            scale = np.max(np.abs(b))
            assert np.max(np.abs(a - b)) <= 1e-9 * scale, (i, c)
    # the first snapshot has received one slice per step, the later ones fewer; untouched planes are zero
    assert ih[0]["slices"] == steps and 0 < ih[2]["slices"] < ih[1]["slices"] < steps
    rho = dh[0]["rho"]   # slices are written from the top of the lab-frame box downwards: nothing below the last one
    assert np.all(rho[:, :, : rho.shape[2] - steps - 1] == 0.0) and np.any(rho != 0.0)


def test_btd_laser_pulse_in_the_lab_frame(host_cpu, tmp_path):
    """tests/decks/boosted_laser_3d.inputs (gamma = 2, window at c, a plane pulse of 0.8 um and 1e12 V/m emitted at
    z = -1 um around t = 20 fs, all lab-frame numbers) with a lab-frame snapshot at 60 fs: the pulse sits at
    -1 um + c (60 - 20) fs with the lab wavelength and amplitude, Bx = -Ey / c -- the Lorentz transform, the slice times
    and the lab-frame indexing at once.  In the boosted frame the same pulse has 3.7 x the wavelength and 0.27 x the
    amplitude (test_boosted_frame_laser_antenna)."""
    deck = os.path.join(HERE, "decks", "boosted_laser_3d.inputs")
    sim = WarpXSim.from_inputs(host_cpu, deck, overrides=("max_step=700",))
    sim.add_btd(2, 60e-15, buffer_size=64)
    sim.evolve(700)
    info = sim.btd_info(1)
    ey = sim.btd_snapshot(1, "Ey")
    bx = sim.btd_snapshot(1, "Bx")
    ex = sim.btd_snapshot(1, "Ex")
    # the snapshot as a plotfile (wxa_sim_btd_write_plotfile), read back with the strict reader of test_plotfile_cpu.py:
    # the same numbers, lab-frame geometry and time
    from tests.test_plotfile_cpu import read_plotfile
    sim.btd_write_plotfile(1, str(tmp_path / "lab_snapshot_1"))
    pf = read_plotfile(str(tmp_path / "lab_snapshot_1"))
    assert pf["names"] == list(WarpXSim.BTD_COMPONENTS) and pf["time"] == info["t_lab"] and pf["step"] == 1
    assert np.array_equal(pf["fields"]["Ey"], ey) and np.array_equal(pf["fields"]["Bx"], bx)
    sim.close()
    assert info["full"] or info["slices"] > 400
    nz = info["n"][2]
    zlo, zhi = info["z_lab"]
    dz = (zhi - zlo) / nz
    z = zlo + (np.arange(nz) + 0.5) * dz
    line = ey[ey.shape[0] // 2, ey.shape[1] // 2, :]
    k = int(np.argmax(np.abs(line)))
    amp = np.max(np.abs(line))
    c = plasma.C_LIGHT
    assert abs(z[k] - (-1e-6 + c * 40e-15)) < 1.5e-6                     # where a lab-frame observer sees it at 60 fs
    # e_max, less what the diagnostic's own averaging takes: cell-centring (two z nodes, 13.7 cells per boosted-frame
    # wavelength: x 0.974), the linear slice interpolation, and 6.5 lab-frame samples per wavelength around the crest
    assert 0.88e12 < amp < 1.02e12
    # wavelength from the zero crossings around the peak
    w = (np.abs(z - z[k]) < 2.0e-6)
    seg, zs = line[w], z[w]
    cross = np.flatnonzero(seg[:-1] * seg[1:] < 0)
    zc = zs[cross] + (zs[cross + 1] - zs[cross]) * seg[cross] / (seg[cross] - seg[cross + 1])
    lam = 2.0 * np.mean(np.diff(zc))
    assert abs(lam - 0.8e-6) < 0.04e-6
    # a plane wave along +z polarised along y: Bx = -Ey / c, nothing in Ex
    assert np.max(np.abs(bx + ey / c)) < 0.03 * amp / c
    assert np.max(np.abs(ex)) < 1e-6 * amp


def test_btd_from_an_inputs_file(host_cpu):
    """<diag>.diag_type = BackTransformed in a deck (BTDiagnostics::ReadParameters, BTDiagnostics.cpp:206-292): snapshots every
    dz_snapshots_lab / c, the lab-frame box of the reference's InitializeBufferData (:333-506) -- the lab-frame domain
    [-30 um, 0] plus half a lab cell, its length rounded up to whole buffers."""
    deck = os.path.join(HERE, "decks", "boosted_laser_3d.inputs")
    over = ("max_step=20", "diagnostics.diags_names=d1", "d1.diag_type=BackTransformed", "d1.do_back_transformed_fields=1",
            "d1.num_snapshots_lab=2", "d1.dz_snapshots_lab=18.e-6", "d1.buffer_size=64", "d1.format=plotfile",
            "d1.fields_to_plot=Ex Ey Ez Bx By Bz jx jy jz rho")
    sim = WarpXSim.from_inputs(host_cpu, deck, overrides=over)
    sim.evolve(20)
    a, b = sim.btd_info(0), sim.btd_info(1)
    sim.close()
    c = plasma.C_LIGHT
    assert a["t_lab"] == 0.0 and abs(b["t_lab"] - 18e-6 / c) < 1e-25      # prob_hi = 0: no offset (:346-347)
    assert a["n"][:2] == (8, 8) and a["n"][2] % 64 == 0
    dz_lab = (a["z_lab"][1] - a["z_lab"][0]) / a["n"][2]
    assert abs(a["z_lab"][1] - 0.5 * dz_lab) < 1e-12                      # the lab-frame domain ends at z = 0
    assert a["z_lab"][0] < -30e-6 < a["z_lab"][0] + 64 * dz_lab           # ... and starts at -30 um, rounded up by < 1 buffer
    assert abs((b["z_lab"][1] - a["z_lab"][1]) - c * b["t_lab"]) < 1e-12  # the snapshots ride with the window (:472-475)
    assert a["slices"] == 20 and not a["full"]
    with pytest.raises(Exception):
        WarpXSim.from_inputs(host_cpu, deck, overrides=over[:4] + ("d1.intervals=0:3",))


def test_btd_particles_of_a_plasma_at_rest_in_the_lab(host_cpu, tmp_path):
    """tests/decks/boosted_injection_3d.inputs: a tenuous plasma at rest in the lab frame seen from a gamma = 3 frame, where
    every electron streams backwards with u_z = -gamma beta c.  Back-transformed (BackTransformParticleFunctor.H:106-168) the
    particles a snapshot's plane has met are at rest again, sit inside the snapshot's lab-frame extent, and carry the
    weights they were injected with."""
    deck = os.path.join(HERE, "decks", "boosted_injection_3d.inputs")
    sim = WarpXSim.from_inputs(host_cpu, deck)
    dt_snap = 5 * sim.lib.sim_dt(sim._h) * 3.0
    sim.add_btd(2, dt_snap, buffer_size=16, write_species=True)
    sim.evolve(sim.max_step)
    info = sim.btd_info(1)
    p = sim.btd_particles(1, 0)
    empty = sim.btd_particles(0, 0)
    live = sim.particles(0)
    # the snapshot's plotfile carries them as species "electrons": positions, weight, momenta m u (lab frame)
    from tests.test_plotfile_cpu import read_plotfile
    sim.btd_write_plotfile(1, str(tmp_path / "lab_snapshot_1"))
    sp = read_plotfile(str(tmp_path / "lab_snapshot_1"))["species"]["electrons"]
    sim.close()
    assert np.array_equal(sp["particle_position_z"], p[2]) and np.array_equal(sp["particle_weight"], p[3])
    assert np.array_equal(sp["particle_momentum_x"], p[4] * plasma.M_E) and np.array_equal(sp["particle_position_x"], p[0])
    assert empty.shape[1] == 0        # at t_lab = 0 the lab window [-16 um, 0] lies below the plasma (z > 1 um)
    assert p.shape[1] > 50 and np.all(p[2] > 1e-6 - 1e-9)   # ... a few femtoseconds later its head has entered it
    c = plasma.C_LIGHT
    assert np.max(np.abs(p[6])) < 1e-9 * 3.0 * c           # u_z back to 0 (it is -gamma beta c = -2.8 c in the boosted frame)
    assert np.max(np.abs(p[4])) < 1e-9 * c and np.max(np.abs(p[5])) < 1e-9 * c
    assert np.all(p[2] > info["z_lab"][0]) and np.all(p[2] < info["z_lab"][1])
    assert np.all(np.abs(live[6] + np.sqrt(8.0) * c) < 1e-6 * c)   # the boosted-frame drift the transform removes
    assert np.allclose(p[3], live[3][0], rtol=1e-12)        # one weight for all: the lab density x the lab cell volume / ppc


def test_btd_snapshots_flushed_to_disk_buffer_by_buffer(host_cpu, tmp_path):
    """wxa_sim_btd_set_flush (BTDiagnostics::Flush + MergeBuffersForPlotfile, BTDiagnostics.cpp:1027-1314): with a file prefix
    every full buffer of 16 slices becomes one more grid of the snapshot's plotfile and only the buffer being filled stays
    in memory; the forced flush after the last step writes the partly filled ones.  Read back with the strict reader of
    test_plotfile_cpu.py the grids hold, bit for bit, what the in-memory diagnostic of a second run holds."""
    from tests.test_plotfile_cpu import read_plotfile
    steps, nsnap, bs = 50, 3, 16
    runs = []
    for flush in (False, True):
        sim, e = pec_case.make_boosted_lwfa_sim(host_cpu)
        sim.add_btd(nsnap, 12 * sim.dt * pec_case.BOOST_GAMMA, buffer_size=bs)
        if flush:
            sim.btd_set_flush(str(tmp_path / "lab"), 5)
        sim.evolve(steps)
        if flush:
            sim.btd_flush()
            with pytest.raises(Exception):
                sim.btd_write_plotfile(0, str(tmp_path / "no"))     # the diagnostic writes its snapshots itself
        info = [sim.btd_info(i) for i in range(nsnap)]
        box = [sim.btd_box(i) for i in range(nsnap)]
        mem = None if flush else [({c: sim.btd_snapshot(i, c) for c in WarpXSim.BTD_COMPONENTS}, sim.btd_particles(i, e))
                                  for i in range(nsnap)]
        runs.append((info, box, mem))
        sim.close()
    (info, box, mem), (info_f, box_f, _) = runs
    assert [a["slices"] for a in info] == [a["slices"] for a in info_f] and box == box_f
    for i in range(nsnap):
        path = str(tmp_path / ("lab%05d" % i))
        pf = read_plotfile(path)
        fabs = sorted(f for f in os.listdir(os.path.join(path, "Level_0")) if f.startswith("Cell_D_"))
        # one grid per started buffer: the planes arrive one per step from the top of the snapshot's box downwards
        assert len(fabs) == -(-info[i]["slices"] // bs) and fabs[0] == "Cell_D_00000"
        assert pf["time"] == info[i]["t_lab"] and pf["step"] == steps and pf["names"] == list(WarpXSim.BTD_COMPONENTS)
        (lo, hi), nz = box[i], info[i]["n"][2]
        k1 = nz - len(fabs) * bs                     # the grids cover the top len(fabs) buffers of the snapshot's box
        for c in WarpXSim.BTD_COMPONENTS:
            assert pf["fields"][c].shape == (info[i]["n"][0], info[i]["n"][1], len(fabs) * bs)
            assert np.array_equal(pf["fields"][c], mem[i][0][c][:, :, k1:]), (i, c)
            assert not np.any(mem[i][0][c][:, :, :k1])                  # nothing below them yet
        sp = pf["species"]["species0"]
        want = mem[i][1]
        got = np.array([sp["particle_position_x"], sp["particle_position_y"], sp["particle_position_z"], sp["particle_weight"],
                        sp["particle_momentum_x"] / plasma.M_E, sp["particle_momentum_y"] / plasma.M_E,
                        sp["particle_momentum_z"] / plasma.M_E])
        assert got.shape == want.shape and (i > 0 or got.shape[1] > 100)
        assert np.array_equal(got[:4], want[:4])                        # arrival order is flush order
        assert np.allclose(got[4:], want[4:], rtol=1e-15, atol=0.0)


def test_btd_file_prefix_in_an_inputs_file(host_cpu, tmp_path):
    """<diag>.file_prefix in a deck turns the flushes on (the reference always writes; here the snapshots otherwise stay in
    memory): 20 steps of the gamma = 2 antenna deck with buffers of 8 slices leave two complete grids and, after the forced
    flush, a third."""
    from tests.test_plotfile_cpu import read_plotfile
    deck = os.path.join(HERE, "decks", "boosted_laser_3d.inputs")
    prefix = str(tmp_path / "diags" / "lab")
    over = ("max_step=20", "diagnostics.diags_names=d1", "d1.diag_type=BackTransformed", "d1.do_back_transformed_fields=1",
            "d1.num_snapshots_lab=1", "d1.dz_snapshots_lab=18.e-6", "d1.buffer_size=8", "d1.format=plotfile",
            f"d1.file_prefix={prefix}", "d1.file_min_digits=3")
    sim = WarpXSim.from_inputs(host_cpu, deck, overrides=over, diagnostics=True)
    sim.evolve(19)
    assert len(read_plotfile(prefix + "000")["fields"]["Ey"][0, 0, :]) == 16
    with pytest.raises(Exception):
        sim.btd_snapshot(0, "Ey")            # flushed, not kept
    sim.evolve(1)                            # the deck's max_step: the forced flush of the last time step, BTD included
    sim.close()                              # (MultiDiagnostics::FilterComputePackFlushLastTimestep; round 4)
    pf = read_plotfile(prefix + "000")
    assert pf["fields"]["Ey"].shape[2] == 24 and pf["step"] == 20
    # diagnostics=False (warpx_amd.write_diagnostics = 0): nothing on disk, the snapshot stays in memory
    prefix2 = str(tmp_path / "diags2" / "lab")
    quiet = WarpXSim.from_inputs(host_cpu, deck, overrides=over[:-2] + (f"d1.file_prefix={prefix2}", "d1.file_min_digits=3"),
                                 diagnostics=False)
    quiet.evolve(20)
    assert not os.path.exists(str(tmp_path / "diags2")) and quiet.btd_snapshot(0, "Ey").shape[2] >= 24
    quiet.close()
