"""Multi-brick path on the HIP kernels with ONE GPU: the bricks are threads of this process that
share the device, and a test-only in-process transport stands in for torch.distributed (same
wxa_comm callbacks as warpx_amd.distributed.TorchBrickTransport).  Covers what the gloo tests on the
CPU build cannot: device-side halo pack/unpack, leaver lists / retirement / arrivals as tile tails,
the LDS-tile kernels on a tile that Redistribute has touched.  Reference: single-domain CPU oracle."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from tests import helpers as H
from warpx_amd import _capi, plasma
from warpx_amd.distributed import _as_tensor, brick_coord
from warpx_amd.sim import WarpXSim, particle_moments
from tests.ports import free_port

pytestmark = pytest.mark.gpu
L = 40e-6
FIELDS = ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz")


def thread_transport_state():
    """What the bricks (threads) of one test share: the mailboxes, a condition for arrivals, the turn lock."""
    return {"box": {}, "cbox": {}, "arrival": threading.Condition(), "turn": threading.Lock(), "failed": []}


def thread_transport_abort(shared):
    with shared["arrival"]:
        shared["failed"].append(True)
        shared["arrival"].notify_all()


class ThreadBrickTransport:
    """Mailbox exchange between the bricks (threads) of one process.  Matching is per pair of bricks, like RCCL's and
    gloo's: the k-th message a brick sends to a peer is the k-th that peer receives from it, whatever the other bricks do
    meanwhile (bricks along a moving window do not all take part in every exchange)."""

    def __init__(self, rank, nranks, shared):
        self.rank, self.nranks, self.shared = rank, nranks, shared
        self.n_exchanges = 0
        self._sent, self._received = {}, {}       # (mailbox, peer) -> messages so far
        self._exchange_cb = _capi.EXCHANGE_FN(self._exchange)
        self._counts_cb = _capi.EXCHANGE_COUNTS_FN(self._exchange_counts)
        self.comm = _capi.Comm()
        self.comm.ctx = None
        self.comm.rank, self.comm.nranks = rank, nranks
        self.comm.exchange, self.comm.exchange_counts = self._exchange_cb, self._counts_cb

    def _post(self, mailbox, peer, item):
        k = self._sent.get((mailbox, peer), 0)
        self._sent[(mailbox, peer)] = k + 1
        with self.shared["arrival"]:
            self.shared[mailbox][(self.rank, peer, k)] = item
            self.shared["arrival"].notify_all()

    def _take(self, mailbox, peer):
        k = self._received.get((mailbox, peer), 0)
        self._received[(mailbox, peer)] = k + 1
        key = (peer, self.rank, k)
        # One brick at a time runs native code (the lock is handed over only while a brick waits for a message): the
        # bricks of a real run are separate processes, concurrency inside one process is not what these tests are about.
        self.shared["turn"].release()
        try:
            with self.shared["arrival"]:
                ok = self.shared["arrival"].wait_for(lambda: key in self.shared[mailbox] or self.shared["failed"], timeout=120)
                if not ok or key not in self.shared[mailbox]:
                    raise RuntimeError("no message from brick %d" % peer)
                return self.shared[mailbox].pop(key)
        finally:
            self.shared["turn"].acquire()

    def _exchange(self, ctx, nmsg, send_peer, send_buf, send_bytes, recv_peer, recv_buf, recv_bytes, stream):
        try:
            H.device_sync()
            # an empty message is no message, on either side (the RCCL and the torch.distributed transports skip them too:
            # a brick with nothing to hand to a peer and nothing to expect from it may not make the call at all)
            for i in range(nmsg):
                n = int(send_bytes[i])
                if n:
                    self._post("box", int(send_peer[i]), _as_tensor(send_buf[i], n, H.ON_GPU).clone())
            for i in range(nmsg):
                n = int(recv_bytes[i])
                if n:
                    t = self._take("box", int(recv_peer[i]))
                    assert t.numel() == n, (self.rank, int(recv_peer[i]), n, t.numel())
                    _as_tensor(recv_buf[i], n, H.ON_GPU).copy_(t)
            H.device_sync()
            self.n_exchanges += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"[ThreadBrickTransport] exchange failed on rank {self.rank}: {e!r}", flush=True)
            thread_transport_abort(self.shared)
            return -1

    def _exchange_counts(self, ctx, nmsg, send_peer, send_val, recv_peer, recv_val):
        try:
            for i in range(nmsg):
                self._post("cbox", int(send_peer[i]), int(send_val[i]))
            for i in range(nmsg):
                recv_val[i] = self._take("cbox", int(recv_peer[i]))
            return 0
        except Exception as e:
            print(f"[ThreadBrickTransport] exchange_counts failed on rank {self.rank}: {e!r}", flush=True)
            thread_transport_abort(self.shared)
            return -1


@pytest.mark.parametrize("nb", [(1, 1, 2), (2, 2, 2)])
def test_bricks_with_overlapped_halo_exchange(oracle, product, nb):
    """overlap_halo = 1 on the HIP path: J's guard sum on the exchange stream behind the first half update of B."""
    test_bricks_on_one_gpu_match_single_domain(oracle, product, nb, 3, 1, overlap=1)


@pytest.mark.parametrize("nb,order,filt", [
    ((1, 1, 2), 3, 1),   # the 2-GPU layout of bench.py
    ((1, 2, 2), 3, 1),   # the 4-GPU layout: edges/corners travel through two exchanged directions
    ((2, 1, 1), 2, 0),
    ((2, 2, 2), 3, 1),   # the 8-GPU layout: corners travel through all three directions
])
def test_bricks_on_one_gpu_match_single_domain(oracle, product, nb, order, filt, overlap=0):
    n_cell = (32, 32, 32)
    steps = 7
    results, parts, bn = run_bricks(product, nb, order, filt, overlap, n_cell, steps)
    prob_lo, prob_hi = (-L / 2,) * 3, (L / 2,) * 3
    ref = WarpXSim(oracle, n_cell, prob_lo, prob_hi, nox=order, use_filter=filt)
    rid = ref.add_species(-plasma.Q_E, plasma.M_E, list(parts))
    ref.evolve(steps)
    rmom = particle_moments(ref, rid)
    assert sum(r["np"] for r in results) == parts.shape[1]      # nobody lost, duplicated or left retired
    assert all(r["inside"] and r["live"] for r in results)
    assert results[0]["exchanges"] > 0
    for n in FIELDS:
        full = ref.field_valid(n)
        scale = max(np.max(np.abs(full)), 1e-300)
        for r in results:
            c, a = r["coord"], r["fields"][n]
            sl = tuple(slice(c[d] * bn[d], c[d] * bn[d] + a.shape[d]) for d in range(3))
            assert float(np.max(np.abs(a - full[sl])) / scale) < 1e-10, n
    ek = sum(r["ekin"] for r in results)
    assert abs(ek - rmom["ekin"]) / rmom["ekin"] < 1e-11
    ap = np.sum([r["abs_p"] for r in results], axis=0)
    assert np.max(np.abs(ap - np.array(rmom["abs_momentum"])) / np.array(rmom["abs_momentum"])) < 1e-11


@pytest.mark.parametrize("nb", [(1, 1, 2), (2, 2, 2)])
def test_guard_layer_update_equals_the_guard_exchange(product, nb, monkeypatch):
    """The all-periodic step updates the first guard layer of B itself (wxa_evolve_b_guard_layer) instead of exchanging
    it after EvolveB, and drops FillBoundaryE after EvolveE: the redundant guard values must be the neighbour's bit
    for bit.  WXA_NO_GUARD_LAYER=1 brings the reference's exchanges back; both schedules leave the same fields and
    particles on every brick."""
    n_cell, steps = (32, 32, 32), 7
    monkeypatch.delenv("WXA_NO_GUARD_LAYER", raising=False)
    fast, _, _ = run_bricks(product, nb, 3, 1, 0, n_cell, steps)
    monkeypatch.setenv("WXA_NO_GUARD_LAYER", "1")
    plain, _, _ = run_bricks(product, nb, 3, 1, 0, n_cell, steps)
    assert plain[0]["exchanges"] > fast[0]["exchanges"]          # the exchanges really came back
    # bit for bit where a run is reproducible (the CPU execution model); on the GPU the order of the deposition's
    # atomics differs from run to run, so two runs of one schedule already differ at round-off: a stale or wrong guard
    # value would show at the scale of the fields themselves
    for a, b in zip(fast, plain):
        assert a["np"] == b["np"]
        for n in FIELDS:
            if H.ON_GPU:
                scale = max(np.max(np.abs(b["fields"][n])), 1e-300)
                assert float(np.max(np.abs(a["fields"][n] - b["fields"][n])) / scale) < 1e-11, n
            else:
                assert np.array_equal(a["fields"][n], b["fields"][n]), n
        assert abs(a["ekin"] - b["ekin"]) <= (1e-12 if H.ON_GPU else 0.0) * b["ekin"]


def run_bricks(product, nb, order, filt, overlap, n_cell, steps, before=None, after=None):
    """before(sim, rank) / after(sim, rank): called on every brick before the first step and after the last one (collective
    output calls such as the plotfile go here: all bricks make them, in the same order)."""
    nranks = nb[0] * nb[1] * nb[2]
    prob_lo, prob_hi = (-L / 2,) * 3, (L / 2,) * 3
    # hot plasma: particles cross brick faces (and tile boundaries) within a few steps
    parts = np.array(plasma.uniform_plasma(n_cell, prob_lo, prob_hi, (1, 2, 1), 1e25, 0.3, seed=11))
    bn = [n_cell[d] // nb[d] for d in range(3)]
    dx = [L / n_cell[d] for d in range(3)]
    shared = thread_transport_state()
    results, errors = [None] * nranks, []

    def brick(rank):
        shared["turn"].acquire()
        try:
            coord = brick_coord(rank, nb)
            lo = [prob_lo[d] + coord[d] * bn[d] * dx[d] for d in range(3)]
            hi = [prob_lo[d] + (coord[d] + 1) * bn[d] * dx[d] for d in range(3)]
            mine = np.ones(parts.shape[1], dtype=bool)
            for d in range(3):
                mine &= (parts[d] >= lo[d]) & (parts[d] < hi[d])
            tr = ThreadBrickTransport(rank, nranks, shared)
            sim = WarpXSim(product, n_cell, prob_lo, prob_hi, nox=order, use_filter=filt, sort_interval=3,
                           nbricks=nb, coord=coord, comm=tr.comm, overlap_halo=overlap)
            assert sim.halo_overlap == bool(overlap)
            sid = sim.add_species(-plasma.Q_E, plasma.M_E, list(parts[:, mine]))
            if before:
                before(sim, rank)
            sim.evolve(steps)
            if after:
                after(sim, rank)
            p = sim.particles(sid)
            mom = particle_moments(sim, sid)
            results[rank] = {
                "coord": coord, "fields": {n: sim.field_valid(n) for n in FIELDS}, "np": p.shape[1],
                "inside": all(np.all((p[d] >= lo[d]) & (p[d] < hi[d])) for d in range(3)),
                "live": bool(np.all(p[3] > 0.0)), "ekin": mom["ekin"], "abs_p": mom["abs_momentum"],
                "exchanges": tr.n_exchanges}
            sim.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            thread_transport_abort(shared)
        finally:
            shared["turn"].release()

    threads = [threading.Thread(target=brick, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r is not None for r in results)

    return results, parts, bn


@pytest.mark.parametrize("nb", [(1, 1, 2), (2, 2, 1)])
def test_one_plotfile_and_reduced_diags_for_all_bricks(oracle, product, nb, tmp_path):
    """The output side of a multi-brick run on the HIP path: every brick writes its FAB and its particles into ONE plotfile
    (brick 0 the headers: wxa_sim_write_plotfile), and brick 0 writes the reduced-diagnostics rows summed over the bricks
    (device reductions + ReduceRealSum).  The plotfile, read back with the strict reader of tests/test_plotfile_cpu.py,
    holds the single-domain oracle's cell-centred fields and every particle once; the rows equal the oracle stepper's."""
    from tests.test_plotfile_cpu import read_plotfile
    n_cell, steps, order, filt = (32, 32, 32), 5, 3, 1
    plt = str(tmp_path / "plt")
    rd_path = str(tmp_path / "reduced") + "/"

    def before(sim, rank):
        for name, kind in (("EF", "FieldEnergy"), ("EP", "ParticleEnergy"), ("NP", "ParticleNumber")):
            sim.add_reduced_diag(name, kind, "1", rd_path)

    def after(sim, rank):
        sim.write_plotfile(plt)

    results, parts, bn = run_bricks(product, nb, order, filt, 0, n_cell, steps, before=before, after=after)
    pf = read_plotfile(plt)
    prob_lo, prob_hi = (-L / 2,) * 3, (L / 2,) * 3
    ref = WarpXSim(oracle, n_cell, prob_lo, prob_hi, nox=order, use_filter=filt)
    rid = ref.add_species(-plasma.Q_E, plasma.M_E, list(parts))
    os.makedirs(str(tmp_path / "oracle"), exist_ok=True)
    for name, kind in (("EF", "FieldEnergy"), ("EP", "ParticleEnergy"), ("NP", "ParticleNumber")):
        ref.add_reduced_diag(name, kind, "1", str(tmp_path / "oracle") + "/")
    ref.evolve(steps)
    ref.compute_rho()
    assert pf["step"] == steps and len(open(os.path.join(plt, "Level_0", "Cell_H")).read().split("FabOnDisk:")) == 1 + nb[0] * nb[1] * nb[2]
    for n in FIELDS + ("rho",):
        full = ref.field(n)                       # with guards, staggered: average to the cell centres like the writer
        v = ref.field_view(n)
        g, st = tuple(v.ng), tuple(v.stag)
        a = full[g[0]:full.shape[0] - g[0], g[1]:full.shape[1] - g[1], g[2]:full.shape[2] - g[2]]
        for d in range(3):
            if st[d]:
                lo = [slice(None)] * 3
                hi = [slice(None)] * 3
                lo[d], hi[d] = slice(0, -1), slice(1, None)
                a = 0.5 * (a[tuple(lo)] + a[tuple(hi)])
        assert a.shape == pf["fields"][n].shape == n_cell, n
        assert float(np.max(np.abs(pf["fields"][n] - a)) / max(np.max(np.abs(a)), 1e-300)) < 1e-10, n
    sp = pf["species"]["species0"]
    assert sp["particle_weight"].size == parts.shape[1] and np.all(sp["particle_weight"] > 0.0)
    rp = ref.particles(rid)
    for k, row, fac in (("particle_position_x", 0, 1.0), ("particle_position_z", 2, 1.0), ("particle_momentum_y", 5, plasma.M_E)):
        assert abs(np.sum(np.abs(sp[k])) - fac * np.sum(np.abs(rp[row]))) <= 1e-11 * fac * np.sum(np.abs(rp[row])), k
    for name in ("EF", "EP", "NP"):
        a = np.atleast_2d(np.genfromtxt(rd_path + name + ".txt"))
        b = np.atleast_2d(np.genfromtxt(str(tmp_path / "oracle") + "/" + name + ".txt"))
        assert a.shape == b.shape and a.shape[0] == steps + 1, name
        scale = np.maximum(np.max(np.abs(b[:, 2:]), axis=0), 1e-300)
        assert np.max(np.abs(a[:, 2:] - b[:, 2:]) / scale) < 1e-10, name


@pytest.mark.skipif(not H.ON_GPU, reason="RCCL needs a GPU (the CPU execution model has no transport of its own)")
def test_rccl_transport_loopback(product):
    """The library's own transport (csrc/rccl_comm.hip) on one GPU: a one-rank RCCL communicator in loop-back mode, so
    that the messages BrickComm would send to a neighbour really travel through an ncclSend / ncclRecv group on the
    caller's stream -- two messages each way to the same peer (the 2 x 2 x 2 case: both faces of a direction lead to
    one rank), matched by posting order; then the 8-byte count exchange on the transport's own stream."""
    import ctypes as C

    import torch

    from warpx_amd import _capi
    from warpx_amd.distributed import RcclBrickTransport
    tr = RcclBrickTransport(product, rank=0, nranks=1, loopback=True, timing=True)
    n0, n1 = 1 << 20, 3000
    s0 = torch.arange(n0, dtype=torch.float64, device="cuda")
    s1 = -torch.arange(n1, dtype=torch.float64, device="cuda")
    r0 = torch.zeros(n0, dtype=torch.float64, device="cuda")
    r1 = torch.zeros(n1, dtype=torch.float64, device="cuda")
    peers = (C.c_int32 * 2)(0, 0)
    sbuf = (C.c_void_p * 2)(s0.data_ptr(), s1.data_ptr())
    rbuf = (C.c_void_p * 2)(r0.data_ptr(), r1.data_ptr())
    sbytes = (C.c_int64 * 2)(8 * n0, 8 * n1)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = tr.comm.exchange(tr.comm.ctx, 2, peers, sbuf, sbytes, peers, rbuf, sbytes, stream)
        assert rc == 0, product.last_error()
    torch.cuda.synchronize()
    assert torch.equal(r0, s0) and torch.equal(r1, s1)
    sv = (C.c_int64 * 2)(123456789012, 7)
    rv = (C.c_int64 * 2)(0, 0)
    assert tr.comm.exchange_counts(tr.comm.ctx, 2, peers, sv, peers, rv) == 0, product.last_error()
    assert list(rv) == [123456789012, 7]
    st = tr.stats()
    assert st["n_exchanges"] == 3 and st["n_messages"] == 6 and st["bytes_sent"] == 3 * 8 * (n0 + n1)
    assert st["timed_exchanges"] == 3 and st["timed_ms"] > 0.0
    print("rccl loop-back:", st)
    tr.close()


@pytest.mark.skipif(not H.ON_GPU, reason="RCCL needs a GPU (the CPU execution model has no transport of its own)")
def test_rccl_count_round_does_not_wait_for_a_queued_data_exchange(product):
    """The 2 x 2 x 2 posting pattern on the loop-back transport -- six messages each way to ONE peer (both faces of every
    direction lead to the same rank), six distinct sizes, matched by posting order -- enqueued on a stream that is still
    busy with a long kernel; the count round is then called on the host.  It has a communicator and a stream of its own
    (ncclCommSplit, round 4): it returns the right counts while the data exchange has not even started, where on the
    shared communicator it queued behind it (the host then waited for the whole step on every rank)."""
    import ctypes as C
    import time

    import torch

    from warpx_amd.distributed import RcclBrickTransport
    tr = RcclBrickTransport(product, rank=0, nranks=1, loopback=True, timing=False)
    sizes = [1 << 20, 3000, 77, 1 << 18, 12345, 9]
    send = [torch.arange(n, dtype=torch.float64, device="cuda") * (i + 1) for i, n in enumerate(sizes)]
    recv = [torch.zeros(n, dtype=torch.float64, device="cuda") for n in sizes]
    peers = (C.c_int32 * 6)(*([0] * 6))
    sbuf = (C.c_void_p * 6)(*[t.data_ptr() for t in send])
    rbuf = (C.c_void_p * 6)(*[t.data_ptr() for t in recv])
    sbytes = (C.c_int64 * 6)(*[8 * n for n in sizes])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.cuda._sleep(int(3e9))          # ~1.5 s of the stream's time in front of the exchange
        rc = tr.comm.exchange(tr.comm.ctx, 6, peers, sbuf, sbytes, peers, rbuf, sbytes, side.cuda_stream)
        assert rc == 0, product.last_error()
        done = torch.cuda.Event()
        done.record(side)
    sv = (C.c_int64 * 6)(*[10 ** 10 + i for i in range(6)])
    rv = (C.c_int64 * 6)(*([0] * 6))
    t0 = time.perf_counter()
    assert tr.comm.exchange_counts(tr.comm.ctx, 6, peers, sv, peers, rv) == 0, product.last_error()
    waited = time.perf_counter() - t0
    still_queued = not done.query()
    assert list(rv) == list(sv)
    side.synchronize()
    for s, r in zip(send, recv):
        assert torch.equal(s, r)
    print(f"count round returned after {1e3 * waited:.1f} ms; data exchange still queued: {still_queued}")
    if os.environ.get("WXA_RCCL_SHARED_COMM") != "1":
        assert still_queued and waited < 0.5, (waited, still_queued)
    tr.close()


def test_back_transformed_diagnostics_on_bricks(product):
    """<diag>.diag_type = BackTransformed on the HIP path with four bricks stacked along z, the boost and window direction
    (config 5 in small, 50 steps, three lab-frame snapshots): every brick fills the slices whose plane lies in its cells,
    the plane behind a z face travels through the transport from device memory (BTDiagnostics::exchange_guard_planes),
    particles are selected on the brick they live on.  The shares add up to the one-brick run of the same library: fields
    at 1e-9 of their scale, the same back-transformed electrons."""
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decks", "laser_wakefield_boosted_3d.inputs")
    nb, nsteps, nsnap = (1, 1, 4), 50, 3
    probe = WarpXSim.from_inputs(product, deck)
    dt_snap = 12 * probe.dt * 5.0
    probe.close()
    over = ("diagnostics.diags_names=d1", "d1.diag_type=BackTransformed", "d1.do_back_transformed_fields=1",
            f"d1.num_snapshots_lab={nsnap}", f"d1.dt_snapshots_lab={dt_snap!r}", "d1.buffer_size=32", "d1.format=plotfile",
            "d1.fields_to_plot=Ex Ey Ez Bx By Bz jx jy jz rho")

    def snapshots(sim):
        return [{"box": sim.btd_box(i), "slices": sim.btd_info(i)["slices"],
                 "data": {c: sim.btd_snapshot(i, c) for c in WarpXSim.BTD_COMPONENTS},
                 "particles": sim.btd_particles(i, 0)} for i in range(nsnap)]

    one = WarpXSim.from_inputs(product, deck, overrides=over)
    one.evolve(nsteps)
    want = snapshots(one)
    one.close()

    nranks = nb[0] * nb[1] * nb[2]
    shared = thread_transport_state()
    shares, errors = [None] * nranks, []

    def brick(rank):
        shared["turn"].acquire()
        try:
            tr = ThreadBrickTransport(rank, nranks, shared)
            sim = WarpXSim.from_inputs(product, deck, overrides=over, nbricks=nb, coord=brick_coord(rank, nb), comm=tr.comm)
            sim.evolve(nsteps)
            shares[rank] = snapshots(sim)
            sim.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            thread_transport_abort(shared)
        finally:
            shared["turn"].release()

    threads = [threading.Thread(target=brick, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    for i in range(nsnap):
        assert all(sh[i]["slices"] == want[i]["slices"] and sh[i]["box"] == want[i]["box"] for sh in shares)
        for c in WarpXSim.BTD_COMPONENTS:
            whole = sum(sh[i]["data"][c] for sh in shares)       # same x-y box on every brick: the shares differ in z only
            b = want[i]["data"][c]
            assert np.max(np.abs(whole - b)) <= 1e-9 * np.max(np.abs(b)), (i, c)
        a, b = np.concatenate([sh[i]["particles"] for sh in shares], axis=1), want[i]["particles"]
        assert a.shape == b.shape and (i > 0 or a.shape[1] > 100)
        a, b = (q[:, np.lexsort((q[2], np.round(q[1] / 1e-10), np.round(q[0] / 1e-10)))] for q in (a, b))
        for row in range(7):
            assert np.max(np.abs(a[row] - b[row])) <= 1e-9 * max(np.max(np.abs(b[row])), 1e-300), (i, row)


def test_bricks_flush_one_back_transformed_plotfile_on_gpu(product, tmp_path):
    """<diag>.file_prefix on the HIP path with two columns of two bricks (threads of one process): one plotfile per lab-frame
    snapshot for the whole run, written by brick 0 from the bricks' shares (BTDiagnostics::flush_bricks; the shares travel
    through the transport from device staging buffers, GatherRealToRoot) -- the grids, fields (1e-9) and back-transformed
    electrons of the one-brick run of the same library.  tests/test_multibrick_cpu.py has the same over gloo on the CPU
    kernels."""
    from tests.test_plotfile_cpu import read_plotfile
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decks", "laser_wakefield_boosted_3d.inputs")
    nb, nsteps, nsnap = (2, 1, 2), 50, 3
    probe = WarpXSim.from_inputs(product, deck)
    dt_snap = 12 * probe.dt * 5.0
    probe.close()

    def overrides(prefix):
        return ("diagnostics.diags_names=d1", "d1.diag_type=BackTransformed", "d1.do_back_transformed_fields=1",
                f"d1.num_snapshots_lab={nsnap}", f"d1.dt_snapshots_lab={dt_snap!r}", "d1.buffer_size=16", "d1.format=plotfile",
                "d1.fields_to_plot=Ex Ey Ez Bx By Bz jx jy jz rho", f"d1.file_prefix={prefix}", "d1.file_min_digits=3",
                f"max_step={nsteps}")
    one_prefix, bricks_prefix = str(tmp_path / "one" / "lab"), str(tmp_path / "bricks" / "lab")
    one = WarpXSim.from_inputs(product, deck, overrides=overrides(one_prefix), diagnostics=True)
    one.evolve(one.max_step)
    one.close()

    nranks = nb[0] * nb[1] * nb[2]
    shared = thread_transport_state()
    errors = []

    def brick(rank):
        shared["turn"].acquire()
        try:
            tr = ThreadBrickTransport(rank, nranks, shared)
            sim = WarpXSim.from_inputs(product, deck, overrides=overrides(bricks_prefix), nbricks=nb, coord=brick_coord(rank, nb),
                                       comm=tr.comm, diagnostics=True)
            sim.evolve(sim.max_step)
            sim.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            thread_transport_abort(shared)
        finally:
            shared["turn"].release()

    threads = [threading.Thread(target=brick, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert sorted(os.listdir(str(tmp_path / "bricks"))) == ["lab%03d" % i for i in range(nsnap)]
    keys = ["particle_position_x", "particle_position_y", "particle_position_z", "particle_weight",
            "particle_momentum_x", "particle_momentum_y", "particle_momentum_z"]
    some_particles = False
    for i in range(nsnap):
        a, b = read_plotfile(one_prefix + "%03d" % i), read_plotfile(bricks_prefix + "%03d" % i)
        la, lb = (sorted(os.listdir(os.path.join(pfx + "%03d" % i, "Level_0"))) for pfx in (one_prefix, bricks_prefix))
        assert la == lb and len(la) >= 2
        assert a["time"] == b["time"] and a["step"] == b["step"] == nsteps and a["names"] == b["names"]
        for c in a["names"]:
            scale = np.max(np.abs(a["fields"][c]))
            assert a["fields"][c].shape == b["fields"][c].shape and scale > 0
            assert np.max(np.abs(a["fields"][c] - b["fields"][c])) <= 1e-9 * scale, (i, c)
        assert list(a["species"]) == list(b["species"]) and len(a["species"]) == 1
        for name in a["species"]:
            pa, pb = (np.array([sp[name][k] for k in keys]) for sp in (a["species"], b["species"]))
            assert pa.shape == pb.shape
            some_particles = some_particles or pa.shape[1] > 100
            pa, pb = (q[:, np.lexsort((q[2], np.round(q[1] / 1e-10), np.round(q[0] / 1e-10)))] for q in (pa, pb))
            for row in range(7):
                assert np.max(np.abs(pa[row] - pb[row])) <= 1e-9 * max(np.max(np.abs(pa[row])), 1e-300), (i, name, row)
    assert some_particles


# ---- the product as N REAL processes on the one MI355X of a box (round 6) ---------------------------------------------
# Every rank is a process of its own with its own HIP context on cuda:0, one brick each; RCCL refuses two ranks on one
# device, so the slabs travel device -> pinned host -> gloo -> pinned host -> device on the exchange's stream
# (warpx_amd.distributed.TorchBrickTransport(staged=True)).  What the thread bricks above cannot cover: the count round,
# the leaver lists and the overlapped schedule between processes that do not share an address space or a turn lock.
def _spawn_bricks(nb, order, filt, overlap, ncell, port, tmp_path, steps=6, extra_env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "report.json")
    n = nb[0] * nb[1] * nb[2]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port(port)),
           os.path.join(root, "tests", "multibrick_worker.py"), *[str(v) for v in nb], str(order), str(filt), out, str(overlap)]
    env = dict(os.environ, OMP_NUM_THREADS=str(max(1, (os.cpu_count() or 8) // n)), WXA_WORKER_LIB="product",
               WXA_TEST_NCELL=" ".join(str(v) for v in ncell), WXA_TEST_STEPS=str(steps), **(extra_env or {}))
    # One more attempt when a child was killed by a GPU exception: seen twice in ~25 runs of the whole suite on the pool's
    # boxes (a "GPU core dump" in one of eight processes, in the first sort of a brick, the parent pytest process holding
    # its own context on the same device), never in 40 runs of these tests on their own (profiles/round6/README.md,
    # session zf) -- not understood; the retry is reported, not hidden.
    for attempt in (1, 2):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        text = r.stdout + r.stderr
        if r.returncode != 0 and attempt == 1 and ("GPU core dump" in text or "Memory access fault" in text):
            head = text[max(0, text.find("GPU core dump") - 1500):][:3000] if "GPU core dump" in text else text[-3000:]
            print(f"[test_multibrick_gpu] a child died of a GPU exception, running the {n} processes once more:\n{head}", flush=True)
            log_dir = os.path.join(root, "gpurun_out")   # (kept across a gpurun call: what the exception was)
            if os.path.isdir(log_dir):
                with open(os.path.join(log_dir, "gpu_exception_retries.txt"), "a") as fh:
                    fh.write(f"==== {nb} overlap {overlap} port {port} ====\n{head}\n")
            continue
        break
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    rep = json.load(open(out))
    rep["gpu_exception_retries"] = attempt - 1
    return rep


@pytest.mark.skipif(not H.ON_GPU, reason="processes that share the one GPU of the box")
@pytest.mark.parametrize("nb,ncell,overlap,port", [
    ((1, 1, 2), (64, 64, 128), 1, 29711),    # bench.py's 2-GPU layout, bricks of 64^3, the overlapped schedule
    ((2, 2, 2), (128, 128, 128), 1, 29712),  # the 8-GPU layout (BASELINE config 4's), bricks of 64^3
    ((2, 2, 2), (128, 128, 128), 0, 29713),  # ... and everything on one stream
])
def test_bricks_as_processes_on_one_gpu(nb, ncell, overlap, port, tmp_path):
    """2 and 8 processes on cuda:0 against (a) the single-domain oracle and (b) the single-domain run of the HIP path, at
    the step tests' gates: 1e-10 on the energies / moments, 1e-9 point-wise (relative to the field's maximum)."""
    rep = _spawn_bricks(nb, 3, 1, overlap, ncell, port, tmp_path)
    print(rep)
    n = nb[0] * nb[1] * nb[2]
    assert len(rep["pids"]) == n                      # really n processes
    assert rep["np_total"] == rep["np_ref"] and rep["inside"]
    assert rep["exchanges"] > 0 and all(b > 0 for b in rep["bytes_sent"])
    for name, err in rep["errors"].items():
        assert err < 1e-9, (name, err)
    for name, err in rep["errors_vs_one_hip_brick"].items():
        assert err < 1e-9, (name, err)
    assert rep["ekin_rel"] < 1e-10 and rep["abs_p_rel"] < 1e-10 and rep["ekin_rel_vs_one_hip_brick"] < 1e-10


@pytest.mark.skipif(not H.ON_GPU, reason="processes that share the one GPU of the box")
def test_single_precision_comms_between_processes(tmp_path):
    """warpx.do_single_precision_comms on the HIP path between 8 real processes: wxa_pack_box_f32 / wxa_unpack_box_f32 on
    the wire of every guard exchange, the fields inside single precision of the fp64 single-domain runs."""
    rep = _spawn_bricks((2, 2, 2), 3, 1, 1, (128, 128, 128), 29714, tmp_path, extra_env={"WXA_TEST_F32_WIRE": "1"})
    print(rep)
    assert rep["np_total"] == rep["np_ref"] and rep["inside"]
    for name, err in rep["errors"].items():
        assert err < 2e-6, (name, err)
    assert max(rep["errors"].values()) > 1e-12        # the wire really was float
    assert rep["ekin_rel"] < 2e-6 and rep["abs_p_rel"] < 2e-6
