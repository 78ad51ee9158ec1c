"""Array containers handed to the C-ABI: one staggered field component with guards
(`FieldArray` ~ one amrex::FArrayBox of a MultiFab) and one pure-SoA particle tile
(`ParticleArrays` ~ WarpXParIter's view, PIdx order x,y,z,w,ux,uy,uz + idcpu;
reference: Source/Particles/NamedComponentParticleContainer.H:23-40).

Storage is a numpy array (host; used by the test-suite with the CPU oracle) or a
torch CUDA tensor (device memory plumbing only).  The layout is the reference's
Array4: Fortran order, i fastest, guards included.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._capi import FieldView, GridGeom, ParticleView

# Yee staggering, 1 = nodal (Source/WarpX.cpp:2117-2125)
STAG = {
    "Ex": (0, 1, 1), "Ey": (1, 0, 1), "Ez": (1, 1, 0),
    "Bx": (1, 0, 0), "By": (0, 1, 0), "Bz": (0, 0, 1),
    "jx": (0, 1, 1), "jy": (1, 0, 1), "jz": (1, 1, 0),
    "rho": (1, 1, 1),
}


def _round_up(a, b):
    return (a + b - 1) // b * b


class FieldArray:
    """One component: valid box [lo_valid, lo_valid + ncell + stag) plus `ng` guards per side."""

    def __init__(self, ncell, stag, ng, device="cpu", lo_valid=(0, 0, 0), pad=False):
        self.ncell = tuple(int(v) for v in ncell)
        self.stag = tuple(int(v) for v in stag)
        self.ng = tuple(int(v) for v in ng)
        self.lo = tuple(int(l) - g for l, g in zip(lo_valid, self.ng))
        self.n = tuple(c + s + 2 * g for c, s, g in zip(self.ncell, self.stag, self.ng))
        self.device = str(device)
        if pad:
            # rows padded to 128 B and the first valid point of every row 128-B aligned
            self.jstride = _round_up(self.n[0], 16)
            self.front = (16 - self.ng[0] % 16) % 16
        else:
            self.jstride = self.n[0]
            self.front = 0
        self.kstride = self.jstride * self.n[1]
        size = self.front + self.kstride * self.n[2]
        if self.device == "cpu":
            self.storage = np.zeros(size, dtype=np.float64)
            self._ptr = self.storage.ctypes.data + 8 * self.front
        else:
            import torch
            self.storage = torch.zeros(size, dtype=torch.float64, device=self.device)
            self._ptr = self.storage.data_ptr() + 8 * self.front
        v = FieldView()
        v.p = self._ptr
        for d in range(3):
            v.lo[d] = self.lo[d]
            v.n[d] = self.n[d]
            v.ng[d] = self.ng[d]
            v.stag[d] = self.stag[d]
        v.jstride = self.jstride
        v.kstride = self.kstride
        self.view = v

    # ---- host <-> container ------------------------------------------------
    def _box(self):
        body = self.storage[self.front:]
        return body.reshape(self.n[2], self.n[1], self.jstride)[:, :, : self.n[0]]

    def to_numpy(self) -> np.ndarray:
        """Dense copy indexed [i - lo0, j - lo1, k - lo2] (guards included)."""
        b = self._box()
        if self.device != "cpu":
            b = b.cpu().numpy()
        return np.ascontiguousarray(np.transpose(b, (2, 1, 0)))

    def from_numpy(self, a: np.ndarray):
        a = np.asarray(a, dtype=np.float64)
        assert a.shape == self.n, (a.shape, self.n)
        b = np.transpose(a, (2, 1, 0))
        if self.device == "cpu":
            self._box()[...] = b
        else:
            import torch
            self._box().copy_(torch.from_numpy(np.ascontiguousarray(b)).to(self.device))
        return self

    def valid(self) -> np.ndarray:
        a = self.to_numpy()
        g = self.ng
        return a[g[0]: self.n[0] - g[0], g[1]: self.n[1] - g[1], g[2]: self.n[2] - g[2]]

    def like(self, device=None, pad=None):
        lo_valid = tuple(l + g for l, g in zip(self.lo, self.ng))
        return FieldArray(self.ncell, self.stag, self.ng, device or self.device, lo_valid,
                          pad=(self.front > 0 or self.jstride != self.n[0]) if pad is None else pad)

    def copy_to(self, device, pad=False):
        other = self.like(device, pad)
        other.from_numpy(self.to_numpy())
        return other


def field_triplet(views):
    arr = (FieldView * 3)()
    for i, f in enumerate(views):
        arr[i] = f.view if isinstance(f, FieldArray) else f
    return arr


class ParticleArrays:
    NAMES = ("x", "y", "z", "w", "ux", "uy", "uz")

    def __init__(self, np_, device="cpu", with_id=False):
        self.np = int(np_)
        self.device = str(device)
        if self.device == "cpu":
            self.data = np.zeros((7, self.np), dtype=np.float64)
            self.idcpu = np.zeros(self.np, dtype=np.uint64) if with_id else None
        else:
            import torch
            self.data = torch.zeros((7, self.np), dtype=torch.float64, device=self.device)
            self.idcpu = torch.zeros(self.np, dtype=torch.int64, device=self.device) if with_id else None

    @classmethod
    def from_numpy(cls, arrays, device="cpu", idcpu=None):
        arrays = [np.asarray(a, dtype=np.float64) for a in arrays]
        self = cls(arrays[0].shape[0], device, with_id=idcpu is not None)
        host = np.stack(arrays)
        if self.device == "cpu":
            self.data[...] = host
            if idcpu is not None:
                self.idcpu[...] = idcpu
        else:
            import torch
            self.data.copy_(torch.from_numpy(host).to(self.device))
            if idcpu is not None:
                self.idcpu.copy_(torch.from_numpy(np.asarray(idcpu).astype(np.int64)).to(self.device))
        return self

    def _row_ptr(self, i):
        if self.device == "cpu":
            return self.data.ctypes.data + 8 * i * self.np
        return self.data.data_ptr() + 8 * i * self.np

    @property
    def view(self) -> ParticleView:
        v = ParticleView()
        v.x, v.y, v.z, v.w = (self._row_ptr(i) for i in range(4))
        v.ux, v.uy, v.uz = (self._row_ptr(i) for i in range(4, 7))
        if self.idcpu is None:
            v.idcpu = None
        elif self.device == "cpu":
            v.idcpu = self.idcpu.ctypes.data
        else:
            v.idcpu = self.idcpu.data_ptr()
        v.np = self.np
        return v

    def to_numpy(self) -> np.ndarray:
        if self.device == "cpu":
            return self.data.copy()
        return np.array(self.data.cpu().numpy())   # an owned copy wherever the tensor lives

    def ids_to_numpy(self) -> np.ndarray:
        """idcpu as int64 (WXA_IDCPU_RETIRED, all ones, reads -1)."""
        if self.idcpu is None:
            raise ValueError("this tile carries no idcpu")
        if self.device == "cpu":
            return self.idcpu.astype(np.int64)
        return np.array(self.idcpu.cpu().numpy()).astype(np.int64)

    def copy_to(self, device):
        ids = None
        if self.idcpu is not None:
            ids = self.idcpu if self.device == "cpu" else self.idcpu.cpu().numpy()
        return ParticleArrays.from_numpy(list(self.to_numpy()), device, ids)


def grid_geom(prob_lo, dx, box_lo, ng) -> GridGeom:
    """Index-space origin of the box grown by `ng`: WarpX::LowerCorner(box.grow(ng))
    = prob_lo + box.lo*dx (Source/WarpX.cpp:2851-2875), lo = lbound(box)."""
    g = GridGeom()
    for d in range(3):
        lo = int(box_lo[d]) - int(ng[d])
        g.lo[d] = lo
        g.xyzmin[d] = float(prob_lo[d]) + float(lo) * float(dx[d])
        g.dinv[d] = 1.0 / float(dx[d])
    return g


def view_to_numpy(view, copy_to_host=None) -> np.ndarray:
    """Dense [i,j,k] copy (guards included) of a FieldView returned by *_sim_get_field.
    `copy_to_host(dst_ptr, src_ptr, nbytes)` is required for device pointers."""
    n = tuple(view.n)
    count = int(view.kstride) * n[2]
    buf = np.empty(count, dtype=np.float64)
    if copy_to_host is None:
        C.memmove(buf.ctypes.data, view.p, 8 * count)
    else:
        copy_to_host(buf.ctypes.data, view.p, 8 * count)
    b = buf.reshape(n[2], n[1], int(view.jstride))[:, :, : n[0]]
    return np.ascontiguousarray(np.transpose(b, (2, 1, 0)))


def particles_to_numpy(pview, copy_to_host=None) -> np.ndarray:
    n = int(pview.np)
    out = np.empty((7, n), dtype=np.float64)
    for i, name in enumerate(ParticleArrays.NAMES):
        src = getattr(pview, name)
        dst = out.ctypes.data + 8 * i * n
        if n == 0:
            continue
        if copy_to_host is None:
            C.memmove(dst, src, 8 * n)
        else:
            copy_to_host(dst, src, 8 * n)
    return out
