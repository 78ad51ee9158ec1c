"""Brick-to-brick transport for the host layer's `wxa_comm` callbacks over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU
tests).  Only plumbing lives here: the exchange plan (which slabs, which neighbours, in
which order) is decided by the C++ host layer (csrc/host/BrickComm.hpp); this module posts
the point-to-point operations it is handed.

The reference has no counterpart: it uses MPI through AMReX.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


def brick_layout(nranks: int):
    """Weak-scaling decomposition of SURVEY.md 8(d) C4: bricks of equal size, split z first,
    then y, then x (1 -> (1,1,1), 2 -> (1,1,2), 4 -> (1,2,2), 8 -> (2,2,2))."""
    nb = [1, 1, 1]
    d = 2
    n = nranks
    while n > 1:
        if n % 2:
            raise ValueError("number of ranks must be a power of two")
        nb[d] *= 2
        n //= 2
        d = (d - 1) % 3
    return tuple(nb)


def brick_coord(rank: int, nbricks):
    """rank = cx + nbx*(cy + nby*cz), the mapping BrickComm::rank_of uses."""
    cx = rank % nbricks[0]
    cy = (rank // nbricks[0]) % nbricks[1]
    cz = rank // (nbricks[0] * nbricks[1])
    return (cx, cy, cz)


class _DevPtr:
    """Wraps a raw device pointer for torch.as_tensor via the CUDA array interface."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}


_tensor_cache = {}


def _as_tensor(ptr, nbytes, on_device):
    """uint8 tensor aliasing [ptr, ptr+nbytes).  The host layer's staging buffers keep their
    addresses from step to step, so the wrappers are cached (wrapping costs ~20 us each)."""
    import torch
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda" if on_device else "cpu")
    key = (int(ptr), int(nbytes), bool(on_device))
    t = _tensor_cache.get(key)
    if t is None:
        if on_device:
            t = torch.as_tensor(_DevPtr(ptr, nbytes), device="cuda")
        else:
            buf = (C.c_uint8 * nbytes).from_address(int(ptr))
            t = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))
        if len(_tensor_cache) > 256:
            _tensor_cache.clear()
        _tensor_cache[key] = t
    return t


class TorchBrickTransport:
    """Owns the ctypes callbacks (must outlive the simulation).

    staged=True (device buffers over a host backend, "gloo"): every message is copied to pinned host memory on the
    exchange's stream, sent with torch.distributed, and copied back to the device on the same stream -- the transport of
    several ranks that share ONE GPU (RCCL refuses two ranks on a device), i.e. the only way the multi-process path of
    BASELINE configs 4 and 5 (bricks, leaver lists, the overlapped schedule) runs on a 1-GPU box."""

    def __init__(self, on_device: bool, group=None, staged: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.on_device = on_device
        self.staged = bool(staged and on_device)
        self._pinned = {}
        self.n_exchanges = 0
        self._streams = {}
        self.bytes_sent = 0
        self._exchange_cb = _capi.EXCHANGE_FN(self._exchange)
        self._counts_cb = _capi.EXCHANGE_COUNTS_FN(self._exchange_counts)
        self.comm = _capi.Comm()
        self.comm.ctx = None
        self.comm.rank = self.rank
        self.comm.nranks = self.nranks
        self.comm.exchange = self._exchange_cb
        self.comm.exchange_counts = self._counts_cb

    def _external_stream(self, ptr):
        import torch
        s = self._streams.get(int(ptr))
        if s is None:
            s = self._streams[int(ptr)] = torch.cuda.ExternalStream(int(ptr))
        return s

    # nmsg sends + nmsg receives; peers equal to this rank are handled by a local copy
    def _exchange(self, ctx, nmsg, send_peer, send_buf, send_bytes, recv_peer, recv_buf, recv_bytes, stream):
        try:
            import torch
            if self.staged:
                return self._exchange_staged(nmsg, send_peer, send_buf, send_bytes, recv_peer, recv_buf, recv_bytes, stream)
            if self.on_device and stream:
                # the library's exchange stream (overlap_halo): RCCL orders its transfers behind the *current*
                # torch stream, so make that stream current for the duration of the call
                with torch.cuda.stream(self._external_stream(stream)):
                    return self._exchange(ctx, nmsg, send_peer, send_buf, send_bytes, recv_peer, recv_buf,
                                          recv_bytes, None)
            dist = self.dist
            ops = []
            keep = []
            sends = [(int(send_peer[i]), send_buf[i], int(send_bytes[i]), i) for i in range(nmsg)]
            recvs = [(int(recv_peer[i]), recv_buf[i], int(recv_bytes[i]), i) for i in range(nmsg)]
            # self-sends: message i of the send list pairs with the receive that expects the
            # opposite direction (send to minus [0] is received as "from plus" [0])
            for (sp, sb, sn, i) in sends:
                if sp == self.rank:
                    rp, rb, rn, _ = recvs[i]
                    assert rp == self.rank and rn == sn
                    if sn:
                        _as_tensor(rb, rn, self.on_device).copy_(_as_tensor(sb, sn, self.on_device))
            # tag = ordinal of the message among those of the same peer: the k-th send to a peer pairs with that peer's
            # k-th receive from here -- RCCL's own matching rule (posting order per peer pair), which is what the callers
            # rely on; gloo matches by tag.  (Zero-length messages are skipped on both sides but still counted.)
            nth = {}
            for (sp, sb, sn, i) in sends:
                k = nth.get(("s", sp), 0)
                nth[("s", sp)] = k + 1
                if sp != self.rank and sn > 0:
                    t = _as_tensor(sb, sn, self.on_device)
                    keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, sp, self.group, tag=k))
                    self.bytes_sent += sn
            for (rp, rb, rn, i) in recvs:
                k = nth.get(("r", rp), 0)
                nth[("r", rp)] = k + 1
                if rp != self.rank and rn > 0:
                    t = _as_tensor(rb, rn, self.on_device)
                    keep.append(t)
                    ops.append(dist.P2POp(dist.irecv, t, rp, self.group, tag=k))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            self.n_exchanges += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            print(f"[warpx_amd.distributed] exchange failed: {e}", flush=True)
            return -1

    def _pinned_like(self, kind, ptr, nbytes):
        """Pinned host twin of the device buffer [ptr, ptr + nbytes): kept per buffer (the host layer's staging buffers keep
        their addresses), so that a copy still in flight on the stream never loses its source."""
        import torch
        key = (kind, int(ptr), int(nbytes))
        t = self._pinned.get(key)
        if t is None:
            if len(self._pinned) > 512:
                torch.cuda.synchronize()
                self._pinned.clear()
            t = self._pinned[key] = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
        return t

    def _exchange_staged(self, nmsg, send_peer, send_buf, send_bytes, recv_peer, recv_buf, recv_bytes, stream):
        """Device buffers over a host backend.  Stream-ordered like the RCCL transport: the packed slabs are read behind
        whatever the stream holds (D2H on it), the host waits for that stream only, and the received slabs are written by
        H2D copies on the same stream -- the unpack kernels the host layer enqueues next are ordered behind them."""
        import torch
        dist = self.dist
        st = self._external_stream(stream) if stream else torch.cuda.current_stream()
        sends = [(int(send_peer[i]), send_buf[i], int(send_bytes[i])) for i in range(nmsg)]
        recvs = [(int(recv_peer[i]), recv_buf[i], int(recv_bytes[i])) for i in range(nmsg)]
        ops, landing = [], []
        with torch.cuda.stream(st):
            nth = {}
            for i, (sp, sb, sn) in enumerate(sends):
                k = nth.get(("s", sp), 0)
                nth[("s", sp)] = k + 1
                if sn == 0:
                    continue
                if sp == self.rank:
                    rp, rb, rn = recvs[i]
                    assert rp == self.rank and rn == sn
                    _as_tensor(rb, rn, True).copy_(_as_tensor(sb, sn, True), non_blocking=True)
                    continue
                h = self._pinned_like("s", sb, sn)
                h.copy_(_as_tensor(sb, sn, True), non_blocking=True)
                ops.append(dist.P2POp(dist.isend, h, sp, self.group, tag=k))
                self.bytes_sent += sn
            for (rp, rb, rn) in recvs:
                k = nth.get(("r", rp), 0)
                nth[("r", rp)] = k + 1
                if rp != self.rank and rn > 0:
                    h = self._pinned_like("r", rb, rn)
                    landing.append((h, rb, rn))
                    ops.append(dist.P2POp(dist.irecv, h, rp, self.group, tag=k))
            # also the last exchange's H2D copies out of the pinned receive buffers have finished after this wait
            st.synchronize()
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for (h, rb, rn) in landing:
                _as_tensor(rb, rn, True).copy_(h, non_blocking=True)
        self.n_exchanges += 1
        return 0

    def _exchange_counts(self, ctx, nmsg, send_peer, send_val, recv_peer, recv_val):
        try:
            import torch
            dist = self.dist
            dev = "cuda" if (self.on_device and not self.staged) else "cpu"
            ops, outs = [], []
            for i in range(nmsg):
                if int(send_peer[i]) == self.rank:
                    recv_val[i] = send_val[i]
            # one staging tensor each way: a single host->device copy before and a single device->host copy
            # (the only synchronisation) after the batch
            sbuf = torch.tensor([int(send_val[i]) for i in range(nmsg)], dtype=torch.int64).to(dev)
            rbuf = torch.zeros(nmsg, dtype=torch.int64, device=dev)
            nth = {}   # tags: ordinal per peer, as in _exchange
            for i in range(nmsg):
                sp = int(send_peer[i])
                k = nth.get(("s", sp), 0)
                nth[("s", sp)] = k + 1
                if sp != self.rank:
                    ops.append(dist.P2POp(dist.isend, sbuf[i:i + 1], sp, self.group, tag=100 + k))
            for i in range(nmsg):
                rp = int(recv_peer[i])
                k = nth.get(("r", rp), 0)
                nth[("r", rp)] = k + 1
                if rp != self.rank:
                    outs.append(i)
                    ops.append(dist.P2POp(dist.irecv, rbuf[i:i + 1], rp, self.group, tag=100 + k))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            if outs:
                got = rbuf.cpu()
                for i in outs:
                    recv_val[i] = int(got[i])
            return 0
        except Exception as e:
            import traceback
            traceback.print_exc()
            print(f"[warpx_amd.distributed] exchange_counts failed: {e}", flush=True)
            return -1


class RcclBrickTransport:
    """The library's own transport (csrc/rccl_comm.hip): ncclSend / ncclRecv groups enqueued on the library's streams
    by C++ callbacks -- no Python in the exchange path, no host wait per message.  torch.distributed is used once, to
    hand rank 0's RCCL unique id to the other ranks."""

    def __init__(self, lib, rank=None, nranks=None, group=None, loopback=False, timing=False):
        import torch.distributed as dist
        self.lib = lib
        have_group = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if have_group else 0)
        self.nranks = nranks if nranks is not None else (dist.get_world_size(group) if have_group else 1)
        # Every step of the set-up is agreed on by all ranks before the next one starts: a rank that failed alone (no
        # RCCL symbols, no unique id, communicator not created) would otherwise leave the others waiting in a
        # broadcast or inside ncclCommInitRank while it has already fallen back to another transport.
        def all_ok(ok, what):
            if self.nranks > 1:
                flags_ = [None] * self.nranks
                dist.all_gather_object(flags_, bool(ok), group=group)
                ok = all(flags_)
            if not ok:
                raise _capi.WxaError(f"RCCL transport: {what} failed on at least one rank")

        uid = (C.c_char * 128)()
        err = None
        if self.rank == 0:
            try:
                lib.rccl_unique_id(uid)
            except Exception as e:   # noqa: BLE001 -- reported to every rank below
                err = repr(e)
        if self.nranks > 1:
            box = [None if err else bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:
                raise _capi.WxaError("RCCL transport: rank 0 could not create the unique id" + (f" ({err})" if err else ""))
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        elif err:
            raise _capi.WxaError(f"RCCL transport: could not create the unique id ({err})")
        all_ok(hasattr(lib, "_rccl_comm_create"), "loading the library's RCCL entry points")
        self.comm = _capi.Comm()
        flags = (_capi.RCCL_LOOPBACK if loopback else 0) | (_capi.RCCL_TIMING if timing else 0)
        created = True
        try:
            lib.rccl_comm_create(uid, self.rank, self.nranks, flags, C.byref(self.comm))
        except Exception as e:   # noqa: BLE001
            created = False
            err = repr(e)
        try:
            all_ok(created, "creating the communicator" + (f" ({err})" if err else ""))
        except _capi.WxaError:
            if created:
                self.close()
            raise

    def stats(self, reset=False):
        st = _capi.RcclStats()
        self.lib.rccl_comm_stats(C.byref(self.comm), C.byref(st), 1 if reset else 0)
        return {k: getattr(st, k) for k, _ in _capi.RcclStats._fields_}

    def set_timing(self, on: bool):
        """A pair of HIP events around every exchange (stats()["timed_ms"]): diagnostic passes only -- reading the
        events waits on the host."""
        self.lib.rccl_comm_set_timing(C.byref(self.comm), 1 if on else 0)

    @property
    def n_exchanges(self):
        return self.stats()["n_exchanges"]

    def close(self):
        if self.comm.ctx:
            self.lib.rccl_comm_destroy(C.byref(self.comm))
