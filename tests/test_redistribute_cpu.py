"""The CPU restatement of the Redistribute primitives (oracle/pic_oracle.cpp: orc_wrap_and_classify,
orc_pack_leavers, retired particles in orc_sort_particles_by_cell) against plain numpy: these are the
checkers of tests/test_kernels_gpu.py::test_redistribute_ops and the backend of the gloo multi-brick
tests, so they are pinned on their own here."""
import ctypes as C

import numpy as np

from tests import helpers as H
from warpx_amd.containers import ParticleArrays

RETIRED = np.uint64(0xFFFFFFFFFFFFFFFF)


def _particles(n, seed, blo, bhi, dx):
    rng = np.random.default_rng(seed)
    pos = [blo[d] - dx[d] + (bhi[d] - blo[d] + 2 * dx[d]) * rng.random(n) for d in range(3)]
    parts = pos + [1e9 * (0.5 + rng.random(n))] + [1e6 * rng.standard_normal(n) for _ in range(3)]
    ids = np.arange(1, n + 1, dtype=np.uint64)
    ids[rng.random(n) < 0.03] = RETIRED
    return parts, ids


def test_wrap_and_classify_against_numpy(oracle):
    ncell = (24, 20, 16)
    dx = H.LX / np.asarray(ncell)
    plo, phi = np.full(3, -H.LX / 2), np.full(3, H.LX / 2)
    blo, bhi = plo.copy(), np.array([0.0, phi[1], 0.0])
    n = 20000
    parts, ids = _particles(n, 5, blo, bhi, dx)
    pc = ParticleArrays.from_numpy(parts, "cpu", ids.copy())
    split = (1, 0, 1)
    lists = np.full(6 * n, -1, dtype=np.int32)
    cnt = (C.c_int64 * 6)()
    oracle.wrap_and_classify(C.byref(pc.view), 0, n, H.d3(plo), H.d3(phi), H.i3((1, 1, 1)), H.d3(blo), H.d3(bhi),
                             H.i3(split), lists.ctypes.data, n, cnt, None, None)
    x = np.array(parts[:3])
    code = np.full(n, -1)
    for d in (2, 1, 0):   # first split direction wins: assign in reverse order
        if split[d]:
            code = np.where(x[d] >= bhi[d], 2 * d + 1, code)
            code = np.where(x[d] < blo[d], 2 * d, code)
    code[ids == RETIRED] = -1
    for c in range(6):
        want = np.nonzero(code == c)[0]
        assert cnt[c] == want.size
        assert np.array_equal(lists[c * n:c * n + cnt[c]], want)        # the CPU version keeps index order
    L = phi - plo
    # retired particles outside the brick along a split direction are parked on its side of the faces (not wrapped)
    ret_out = (ids == RETIRED) & np.any([(x[d] < blo[d]) | (x[d] >= bhi[d]) for d in range(3) if split[d]], axis=0)
    assert ret_out.sum() > 20
    for d in range(3):
        if split[d]:
            x[d] = np.where(ret_out, np.minimum(np.maximum(x[d], blo[d]), np.nextafter(bhi[d], blo[d])), x[d])
    wrapped = np.array([np.where(x[d] >= phi[d], x[d] - L[d], np.where(x[d] < plo[d], x[d] + L[d], x[d]))
                        for d in range(3)])
    got = pc.to_numpy()
    assert np.array_equal(got[:3], wrapped)
    assert np.all(got[:3] >= plo[:, None]) and np.all(got[:3] < phi[:, None])
    assert np.array_equal(got[3:], np.array(parts[3:]))


def test_wrap_and_classify_dest_against_numpy(oracle):
    """wxa_wrap_and_classify_dest's CPU restatement: the leavers listed by the offset of their destination brick (27
    lists, decided on the unwrapped position along the split directions), retired particles parked, positions wrapped."""
    ncell = (24, 20, 16)
    dx = H.LX / np.asarray(ncell)
    plo, phi = np.full(3, -H.LX / 2), np.full(3, H.LX / 2)
    blo, bhi = plo.copy(), np.array([0.0, phi[1], 0.0])
    n = 20000
    parts, ids = _particles(n, 15, blo, bhi, dx)
    pc = ParticleArrays.from_numpy(parts, "cpu", ids.copy())
    split = (1, 0, 1)
    lists = np.full(27 * n, -1, dtype=np.int32)
    cnt = (C.c_int64 * 27)()
    oracle.wrap_and_classify_dest(C.byref(pc.view), 0, n, H.d3(plo), H.d3(phi), H.i3((1, 1, 1)), H.d3(blo), H.d3(bhi),
                                  H.i3(split), lists.ctypes.data, n, cnt, None, None)
    x = np.array(parts[:3])
    off = [np.where(x[d] < blo[d], -1, np.where(x[d] >= bhi[d], 1, 0)) if split[d] else np.zeros(n, dtype=int)
           for d in range(3)]
    code = (off[0] + 1) + 3 * (off[1] + 1) + 9 * (off[2] + 1)
    code[ids == RETIRED] = 13
    assert cnt[13] == 0
    seen = 0
    for c in range(27):
        if c == 13:
            continue
        want = np.nonzero(code == c)[0]
        assert cnt[c] == want.size, c
        assert np.array_equal(lists[c * n:c * n + cnt[c]], want)
        seen += want.size
    # faces, and the edges where both split directions are left at once; nothing along the unsplit direction
    assert cnt[12] > 100 and cnt[14] > 100 and cnt[4] > 100 and cnt[22] > 100
    assert cnt[3] + cnt[5] + cnt[21] + cnt[23] > 10
    assert all(cnt[c] == 0 for c in range(27) if (c // 3) % 3 != 1)
    assert seen == int((code != 13).sum())
    # the positions: the same as after wxa_wrap_and_classify
    pc2 = ParticleArrays.from_numpy(parts, "cpu", ids.copy())
    l6, c6 = np.zeros(6 * n, dtype=np.int32), (C.c_int64 * 6)()
    oracle.wrap_and_classify(C.byref(pc2.view), 0, n, H.d3(plo), H.d3(phi), H.i3((1, 1, 1)), H.d3(blo), H.d3(bhi),
                             H.i3(split), l6.ctypes.data, n, c6, None, None)
    assert np.array_equal(pc.to_numpy(), pc2.to_numpy())
    assert sum(c6) == seen


def test_pack_retire_and_sort(oracle):
    ncell = (12, 20, 8)
    dx = H.LX / np.asarray((24, 20, 16))
    plo = np.full(3, -H.LX / 2)
    blo, bhi = plo.copy(), np.array([0.0, H.LX / 2, 0.0])
    n = 5000
    parts, ids = _particles(n, 6, blo, bhi, dx)
    pc = ParticleArrays.from_numpy(parts, "cpu", ids.copy())
    before = pc.to_numpy()
    lst = np.ascontiguousarray(np.nonzero((before[0] >= bhi[0]) & (ids != RETIRED))[0].astype(np.int32))
    m = lst.size
    assert m > 50
    row_len, off = m + 3, 2
    msg = np.zeros(8 * row_len)
    oracle.pack_leavers(C.byref(pc.view), lst.ctypes.data, m, msg.ctypes.data, row_len, off, 1, H.d3(blo), H.d3(bhi),
                        None)
    rows = msg.reshape(8, row_len)
    assert np.array_equal(rows[:7, off:off + m], before[:, lst])
    assert np.array_equal(rows[7, off:off + m].view(np.uint64), ids[lst])
    assert np.all(rows[:, :off] == 0) and np.all(rows[:, off + m:] == 0)
    after = pc.to_numpy()
    assert np.all(after[3:, lst] == 0.0) and np.all(pc.idcpu[lst] == RETIRED)
    for d in range(3):
        assert np.all(after[d, lst] >= blo[d]) and np.all(after[d, lst] < bhi[d])
    keep = np.setdiff1d(np.arange(n), lst)
    assert np.array_equal(after[:, keep], before[:, keep])
    # sort: live particles first (grouped by cell), the retired ones behind them
    out = ParticleArrays(n, "cpu", with_id=True)
    live = np.zeros(1, dtype=np.int64)
    oracle.sort_particles_by_cell(C.byref(pc.view), C.byref(out.view), H.d3(blo), H.d3(1.0 / dx), H.i3((0, 0, 0)),
                                  (C.c_int32 * 3)(*ncell), live.ctypes.data, None)
    nlive = int((pc.idcpu != RETIRED).sum())
    assert live[0] == nlive
    assert np.all(out.idcpu[:nlive] != RETIRED) and np.all(out.idcpu[nlive:] == RETIRED)
    assert np.array_equal(np.sort(out.idcpu[:nlive]), np.sort(pc.idcpu[pc.idcpu != RETIRED]))
    s = out.to_numpy()[:, :nlive]
    cell = [np.clip(np.floor((s[d] - blo[d]) / dx[d]).astype(np.int64), 0, ncell[d] - 1) for d in range(3)]
    key = cell[0] + ncell[0] * (cell[1] + ncell[1] * cell[2])
    assert np.all(np.diff(key) >= 0)
