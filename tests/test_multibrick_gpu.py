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

pytestmark = pytest.mark.gpu
L = 40e-6
FIELDS = ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz")


class ThreadBrickTransport:
    """Mailbox + barrier exchange between the bricks (threads) of one process."""

    def __init__(self, rank, nranks, shared):
        self.rank, self.nranks, self.shared = rank, nranks, shared
        self.n_exchanges = 0
        self._exchange_cb = _capi.EXCHANGE_FN(self._exchange)
        self._counts_cb = _capi.EXCHANGE_COUNTS_FN(self._exchange_counts)
        self.comm = _capi.Comm()
        self.comm.ctx = None
        self.comm.rank, self.comm.nranks = rank, nranks
        self.comm.exchange, self.comm.exchange_counts = self._exchange_cb, self._counts_cb

    def _wait(self):
        # One brick at a time runs native code (the lock is handed over only at the exchange points):
        # the bricks of a real run are separate processes, concurrency inside one process is not
        # what this test is about.
        self.shared["turn"].release()
        try:
            self.shared["barrier"].wait(timeout=120)
        finally:
            self.shared["turn"].acquire()

    def _exchange(self, ctx, nmsg, send_peer, send_buf, send_bytes, recv_peer, recv_buf, recv_bytes, stream):
        try:
            import torch
            H.device_sync()
            box = self.shared["box"]
            # the k-th message to a peer pairs with that peer's k-th receive from here (RCCL's matching rule, which the
            # host layer relies on: posting order per peer pair)
            nth = {}
            for i in range(nmsg):
                n = int(send_bytes[i])
                k = nth.get(("s", int(send_peer[i])), 0)
                nth[("s", int(send_peer[i]))] = k + 1
                box[(self.rank, int(send_peer[i]), k)] = _as_tensor(send_buf[i], n, H.ON_GPU).clone() if n else None
            self._wait()
            for i in range(nmsg):
                n = int(recv_bytes[i])
                k = nth.get(("r", int(recv_peer[i])), 0)
                nth[("r", int(recv_peer[i]))] = k + 1
                t = box[(int(recv_peer[i]), self.rank, k)]
                assert (t.numel() if t is not None else 0) == n
                if n:
                    _as_tensor(recv_buf[i], n, H.ON_GPU).copy_(t)
            H.device_sync()
            self._wait()
            self.n_exchanges += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"[ThreadBrickTransport] exchange failed on rank {self.rank}: {e!r}", flush=True)
            self.shared["barrier"].abort()
            return -1

    def _exchange_counts(self, ctx, nmsg, send_peer, send_val, recv_peer, recv_val):
        try:
            box = self.shared["cbox"]
            nth = {}
            for i in range(nmsg):
                k = nth.get(("s", int(send_peer[i])), 0)
                nth[("s", int(send_peer[i]))] = k + 1
                box[(self.rank, int(send_peer[i]), k)] = int(send_val[i])
            self._wait()
            for i in range(nmsg):
                k = nth.get(("r", int(recv_peer[i])), 0)
                nth[("r", int(recv_peer[i]))] = k + 1
                recv_val[i] = box[(int(recv_peer[i]), self.rank, k)]
            self._wait()
            return 0
        except Exception as e:
            print(f"[ThreadBrickTransport] exchange_counts failed on rank {self.rank}: {e!r}", flush=True)
            self.shared["barrier"].abort()
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


def run_bricks(product, nb, order, filt, overlap, n_cell, steps):
    nranks = nb[0] * nb[1] * nb[2]
    prob_lo, prob_hi = (-L / 2,) * 3, (L / 2,) * 3
    # hot plasma: particles cross brick faces (and tile boundaries) within a few steps
    parts = np.array(plasma.uniform_plasma(n_cell, prob_lo, prob_hi, (1, 2, 1), 1e25, 0.3, seed=11))
    bn = [n_cell[d] // nb[d] for d in range(3)]
    dx = [L / n_cell[d] for d in range(3)]
    shared = {"box": {}, "cbox": {}, "barrier": threading.Barrier(nranks), "turn": threading.Lock()}
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
            sim.evolve(steps)
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
            shared["barrier"].abort()
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
