"""The plotfile writer (wxa_sim_write_plotfile, host/Plotfile.hpp) closes the loop "this library's output -> the
reference's regression checksum": a deck runs on the CPU build of the host layer, the state is written as an AMReX
plotfile, and the checksum of Regression/Checksum/checksum.py:78-140 (sum |Q| per cell-centred field over the covering
grid, sum |Q| per particle attribute, names with yt's `particle_` prefix) is recomputed FROM THE FILES and held against
the reference's golden JSON.  The files are parsed twice: by a strict reader of the AMReX layouts written here after
BTD_Plotfile_Header_Impl.cpp's own readers, and -- in this container, where /root/reference exists -- by the
reference's Tools/PostProcessing/read_raw_data.py (its VisMF header / FAB reader), imported where it lies."""
import importlib.util
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests.test_inputs_cpu import DECKS, HERE, compare_with_golden
from tests.test_inputs_cpu import lib  # noqa: F401  (fixture)
from warpx_amd.sim import WarpXSim
from tests.ports import free_port

REFERENCE = "/root/reference"


def _read_box(text):
    m = re.match(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\) \((\d),(\d),(\d)\)\)", text.strip())
    assert m, text
    v = [int(x) for x in m.groups()]
    return np.array(v[0:3]), np.array(v[3:6]), np.array(v[6:9])


def read_plotfile(path):
    """Header -> names, geometry; Level_0/Cell_H -> boxes and FabOnDisk; Cell_D_* -> data; <species>/Header + DATA."""
    lines = open(os.path.join(path, "Header")).read().split("\n")
    assert lines[0] == "HyperCLaw-V1.1"
    ncomp = int(lines[1])
    names = lines[2:2 + ncomp]
    p = 2 + ncomp
    assert int(lines[p]) == 3
    time = float(lines[p + 1])
    assert int(lines[p + 2]) == 0                                  # finest level
    prob_lo = [float(x) for x in lines[p + 3].split()]
    prob_hi = [float(x) for x in lines[p + 4].split()]
    assert lines[p + 5] == ""                                      # no refinement ratios
    dlo, dhi, dtype = _read_box(lines[p + 6])
    assert np.all(dtype == 0)
    step = int(lines[p + 7])
    dx = [float(x) for x in lines[p + 8].split()]
    assert int(lines[p + 9]) == 0 and int(lines[p + 10]) == 0       # Cartesian, bwidth
    lev, ngrids, t2 = lines[p + 11].split()
    assert int(lev) == 0 and float(t2) == time
    ngrids = int(ngrids)
    assert int(lines[p + 12]) == step
    q = p + 13
    for g in range(ngrids):
        for d in range(3):
            lo, hi = (float(x) for x in lines[q].split())
            assert lo < hi
            q += 1
    assert lines[q] == "Level_0/Cell"
    assert np.allclose((np.array(prob_hi) - np.array(prob_lo)) / np.array(dx), dhi - dlo + 1)
    # VisMF header
    ch = open(os.path.join(path, "Level_0", "Cell_H")).read().split("\n")
    assert ch[0] == "1" and int(ch[2]) == ncomp and int(ch[3]) == 0
    m = re.match(r"\((\d+) (\d+)", ch[4])
    nb = int(m.group(1))
    boxes = [_read_box(ch[5 + i]) for i in range(nb)]
    assert ch[5 + nb] == ")" and int(ch[6 + nb]) == nb
    fod = [ch[7 + nb + i].split() for i in range(nb)]
    fields = {n: np.zeros(dhi - dlo + 1) for n in names}
    for (lo, hi, _), (tag, fname, off) in zip(boxes, fod):
        assert tag == "FabOnDisk:"
        with open(os.path.join(path, "Level_0", fname), "rb") as f:
            f.seek(int(off))
            head = f.readline().decode()
            assert head.startswith("FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))")
            hlo, hhi, _ = _read_box(head[head.index(")))") + 3:head.rindex(" ")])
            assert np.all(hlo == lo) and np.all(hhi == hi) and int(head.split()[-1]) == ncomp
            shape = hi - lo + 1
            arr = np.fromfile(f, "<f8", ncomp * int(np.prod(shape)))
            sl = tuple(slice(a, b + 1) for a, b in zip(lo - dlo, hi - dlo))
            for i, n in enumerate(names):
                fields[n][sl] = arr[i * int(np.prod(shape)):(i + 1) * int(np.prod(shape))].reshape(shape, order="F")
    species = {}
    for entry in sorted(os.listdir(path)):
        hdr = os.path.join(path, entry, "Header")
        if entry == "Level_0" or not os.path.isfile(hdr):
            continue
        h = open(hdr).read().split("\n")
        assert h[0] == "Version_Two_Dot_One_double" and int(h[1]) == 3
        nreal = int(h[2])
        rnames = h[3:3 + nreal]
        q = 3 + nreal
        nint = int(h[q])
        q += 1 + nint
        assert int(h[q]) == 0                                      # not a checkpoint
        npart, next_id, finest = int(h[q + 1]), int(h[q + 2]), int(h[q + 3])
        assert finest == 0 and next_id > 0
        ngr = int(h[q + 4])
        cols = ["position_x", "position_y", "position_z"] + rnames
        data = np.zeros((0, len(cols)))
        for g in range(ngr):
            which, count, where = (int(x) for x in h[q + 5 + g].split())
            if count == 0:      # a grid without particles has no DATA file (BTDiagnostics.cpp:1274, 1290)
                continue
            with open(os.path.join(path, entry, "Level_0", "DATA_%05d" % which), "rb") as f:
                f.seek(where)
                rec = np.fromfile(f, "<f8", count * len(cols)).reshape(count, len(cols))
                data = np.vstack([data, rec])
        assert data.shape[0] == npart
        assert open(os.path.join(path, entry, "Level_0", "Particle_H")).read().startswith("(%d " % ngr)
        species[entry] = {"particle_" + c: data[:, i] for i, c in enumerate(cols)}
    return {"names": names, "fields": fields, "species": species, "time": time, "step": step}


def checksum_of(pf):
    """Regression/Checksum/checksum.py:98-140 on the parsed plotfile."""
    out = {"lev=0": {n: float(np.sum(np.abs(a))) for n, a in pf["fields"].items()}}
    for s, cols in pf["species"].items():
        out[s] = {k: float(np.sum(np.abs(v))) for k, v in cols.items()}
    return out


@pytest.mark.parametrize("deck,golden", [("langmuir_multi_3d.inputs", "langmuir_multi_3d_checksums.json"),
                                         ("laser_wakefield_3d.inputs", "laser_acceleration_3d_checksums.json")])
def test_plotfile_carries_the_golden_checksums(lib, tmp_path, deck, golden):  # noqa: F811
    gold = json.load(open(os.path.join(HERE, "golden", golden)))
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, deck))
    sim.evolve(sim.max_step)
    plt = str(tmp_path / "plt")
    sim.write_plotfile(plt)
    direct = sim.checksum()
    sim.close()
    pf = read_plotfile(plt)
    got = checksum_of(pf)
    # (i) the files carry what the library's own reducer reports (to the summation order; part_per_cell is not plotted here)
    for group, vals in got.items():
        for k, v in vals.items():
            assert abs(v - direct[group][k]) <= 1e-12 * max(abs(direct[group][k]), 1e-300), (group, k)
    # (ii) and reach the reference's golden file at the reference's tolerance
    gold_cs = {g: {k: v for k, v in vals.items() if k != "part_per_cell"} for g, vals in gold["checksums"].items()}
    worst = compare_with_golden(got, gold_cs, gold["rtol"])
    print(deck, "plotfile -> checksum: worst relative deviation from the reference's golden file", worst)
    assert pf["step"] == sim.max_step


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this machine")
def test_the_reference_reader_opens_the_field_data(lib, tmp_path):  # noqa: F811
    """Tools/PostProcessing/read_raw_data.py (the reference's own VisMF / FAB reader) on Level_0/Cell_H."""
    spec = importlib.util.spec_from_file_location("ref_read_raw_data", os.path.join(REFERENCE, "Tools/PostProcessing/read_raw_data.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    if not hasattr(np, "product"):
        np.product = np.prod   # the reference tool predates numpy 2
    sim = WarpXSim.from_inputs(lib, os.path.join(DECKS, "langmuir_multi_3d.inputs"))
    sim.evolve(4)
    plt = str(tmp_path / "plt")
    sim.write_plotfile(plt)
    sim.close()
    mine = read_plotfile(plt)
    boxes, file_names, offsets, header = ref._read_header(os.path.join(plt, "Level_0", "Cell_H"))
    assert header.version == 1 and header.ncomp == len(mine["names"]) and file_names == ["Cell_D_00000"]
    theirs = ref._read_buffer(plt, os.path.join(plt, "Level_0", "Cell_H"), mine["names"])
    for n in mine["names"]:
        assert np.array_equal(theirs[n], mine["fields"][n]), n
    assert np.max(np.abs(mine["fields"]["Ex"])) > 0


@pytest.mark.parametrize("nb,port", [((1, 2, 2), 29681)])
def test_bricks_write_one_plotfile(tmp_path, nb, port):
    """A run on four bricks (gloo) writes ONE plotfile, as a parallel run of the reference does: a FAB and a particle file
    per brick, the headers by brick 0 listing every grid.  Read back, it carries the golden checksums of the deck -- the
    loop "multi-brick output -> the reference's regression checksum" closed through files."""
    gold = json.load(open(os.path.join(HERE, "golden", "langmuir_multi_3d_checksums.json")))
    deck = os.path.join(DECKS, "langmuir_multi_3d.inputs")
    plt = str(tmp_path / "plt")
    n = nb[0] * nb[1] * nb[2]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port(port)), os.path.join(HERE, "deck_worker.py"),
           *[str(v) for v in nb], deck, str(tmp_path / "sum.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, OMP_NUM_THREADS="2", WXA_TEST_PLOTFILE=plt))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert sorted(os.listdir(os.path.join(plt, "Level_0"))) == ["Cell_D_%05d" % i for i in range(n)] + ["Cell_H"]
    pf = read_plotfile(plt)
    got = checksum_of(pf)
    gold_cs = {g: {k: v for k, v in vals.items() if k != "part_per_cell"} for g, vals in gold["checksums"].items()}
    worst = compare_with_golden(got, gold_cs, gold["rtol"])
    print("four bricks, one plotfile -> checksum: worst relative deviation from the reference's golden file", worst)
    direct = json.load(open(str(tmp_path / "sum.json")))   # the bricks' own checksums, added up
    for group, vals in got.items():
        for k, v in vals.items():
            assert abs(v - direct[group][k]) <= 1e-12 * max(abs(direct[group][k]), 1e-300), (group, k)
