"""Worker for the multi-brick CPU tests: every rank owns one brick, runs the product's host
layer (CPU build, oracle kernels) with the torch.distributed transport over gloo, and rank 0
compares the reassembled result with a single-domain run of the independent oracle stepper.

    python -m torch.distributed.run --nproc-per-node N tests/multibrick_worker.py NBX NBY NBZ ORDER FILTER OUT
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.oracle_lib import load_hip_on_cpu, load_host_cpu, load_oracle  # noqa: E402
from warpx_amd import _capi, plasma  # noqa: E402
from warpx_amd.distributed import TorchBrickTransport, brick_coord  # noqa: E402
from warpx_amd.sim import WarpXSim, field_energy, particle_moments  # noqa: E402

L = 40e-6
FIELDS = ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz", "rho")


def main():
    nb = tuple(int(v) for v in sys.argv[1:4])
    order, filt, out = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    overlap = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    steps = int(os.environ.get("WXA_TEST_STEPS", "6"))
    solver = int(os.environ.get("WXA_TEST_SOLVER", "0"))   # _capi.SOLVER_YEE / SOLVER_CKC
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == nb[0] * nb[1] * nb[2]
    n_cell = tuple(int(v) for v in os.environ.get("WXA_TEST_NCELL", "16 16 16").split())
    prob_lo, prob_hi = (-L / 2,) * 3, (L / 2,) * 3
    # hot plasma so that particles cross brick boundaries within a few steps
    ppc = tuple(int(v) for v in os.environ.get("WXA_TEST_PPC", "1 2 1").split())
    parts = np.array(plasma.uniform_plasma(n_cell, prob_lo, prob_hi, ppc, 1e25, float(os.environ.get("WXA_TEST_UTH", "0.3")), seed=11))
    coord = brick_coord(rank, nb)
    bn = [n_cell[d] // nb[d] for d in range(3)]
    dx = [L / n_cell[d] for d in range(3)]
    lo = [prob_lo[d] + coord[d] * bn[d] * dx[d] for d in range(3)]
    hi = [prob_lo[d] + (coord[d] + 1) * bn[d] * dx[d] for d in range(3)]
    mine = np.ones(parts.shape[1], dtype=bool)
    for d in range(3):
        mine &= (parts[d] >= lo[d]) & (parts[d] < hi[d])
    # WXA_WORKER_LIB=hipcpu: the HIP kernels themselves (tests/hipcpu execution model) instead of the oracle kernels;
    # WXA_WORKER_LIB=product: libwarpx_amd.so on the GPU -- every rank on cuda:0, device buffers staged through pinned host
    # memory over gloo (RCCL refuses two ranks on one device): the product as N real processes on a 1-GPU box
    which = os.environ.get("WXA_WORKER_LIB", "")
    if which == "product":
        from warpx_amd import load_product
        torch.cuda.set_device(0)
        lib = load_product()
        transport = TorchBrickTransport(on_device=True, staged=True)
    else:
        transport = TorchBrickTransport(on_device=False)
        lib = load_hip_on_cpu() if which == "hipcpu" else load_host_cpu()
    sim = WarpXSim(lib, n_cell, prob_lo, prob_hi, nox=order, use_filter=filt, sort_interval=int(os.environ.get("WXA_TEST_SORT", "2")),
                   nbricks=nb, coord=coord, comm=transport.comm, overlap_halo=overlap, maxwell_solver=solver)
    assert sim.halo_overlap == bool(overlap)
    if os.environ.get("WXA_TEST_SAFE_GUARD_CELLS") == "1":
        sim.set_safe_guard_cells(True)
    wire = os.environ.get("WXA_TEST_F32_WIRE", "")
    if wire == "1" or (wire == "rank0" and rank == 0):   # "rank0": the bricks disagree -- refused before the first step
        sim.set_single_precision_comms(True)
    sid = sim.add_species(-plasma.Q_E, plasma.M_E, list(parts[:, mine]))
    dry = None
    if os.environ.get("WXA_TEST_DRY_COMM") == "1":   # bench.py --dry-comm in the middle of a run: it must not disturb it
        sim.evolve(steps // 2)
        before = {n: sim.field_valid(n).copy() for n in ("Ex", "Ey", "Ez", "Bx", "By", "Bz")}
        np_before = sim.particle_view(sid).np
        dry = sim.dry_comm(2)
        assert all(np.array_equal(before[n], sim.field_valid(n)) for n in before), "dry_comm changed E or B"
        assert sim.particle_view(sid).np == np_before
        assert all(np.isfinite(v) and v >= 0.0 for v in dry.values()), dry
        sim.evolve(steps - steps // 2)
    else:
        sim.evolve(steps)
    sim.compute_rho()           # charge deposition + filter + guard sum across bricks
    # ---- collect on rank 0 ----
    local = {n: sim.field_valid(n) for n in FIELDS}
    mom = particle_moments(sim, sid)
    p_local = sim.particles(sid)
    inside = all(np.all((p_local[d] >= lo[d]) & (p_local[d] < hi[d])) for d in range(3))
    payload = {"coord": coord, "fields": local, "ekin": mom["ekin"], "np": p_local.shape[1],
               "abs_p": mom["abs_momentum"], "inside": bool(inside), "exchanges": transport.n_exchanges,
               "bytes_sent": transport.bytes_sent, "pid": os.getpid()}
    sim.close()
    gathered = [None] * world
    dist.gather_object(payload, gathered if rank == 0 else None, dst=0)
    if rank == 0:
        orc = load_oracle()
        ref = WarpXSim(orc, n_cell, prob_lo, prob_hi, nox=order, use_filter=filt, maxwell_solver=solver)
        rid = ref.add_species(-plasma.Q_E, plasma.M_E, list(parts))
        ref.evolve(steps)
        ref.compute_rho()
        rmom = particle_moments(ref, rid)
        import hashlib
        digest = hashlib.sha256()
        for g in sorted(gathered, key=lambda g: g["coord"]):
            for n in FIELDS:
                digest.update(np.ascontiguousarray(g["fields"][n]).tobytes())
        report = {"ok": True, "errors": {}, "digest": digest.hexdigest(), "np_total": sum(g["np"] for g in gathered),
                  "np_ref": int(parts.shape[1]), "inside": all(g["inside"] for g in gathered),
                  "exchanges": gathered[0]["exchanges"], "bytes_sent": [g["bytes_sent"] for g in gathered],
                  "pids": sorted(set(g["pid"] for g in gathered))}

        def worst_over_bricks(full, n):
            worst = 0.0
            for g in gathered:
                c = g["coord"]
                a = g["fields"][n]
                sl = tuple(slice(c[d] * bn[d], c[d] * bn[d] + a.shape[d]) for d in range(3))
                worst = max(worst, float(np.max(np.abs(a - full[sl])) / max(np.max(np.abs(full)), 1e-300)))
            return worst

        for n in FIELDS:
            report["errors"][n] = worst_over_bricks(ref.field_valid(n), n)
        if which == "product":   # ... and against the single-domain run of the HIP path itself
            one = WarpXSim(lib, n_cell, prob_lo, prob_hi, nox=order, use_filter=filt, maxwell_solver=solver,
                           sort_interval=int(os.environ.get("WXA_TEST_SORT", "2")))
            oid = one.add_species(-plasma.Q_E, plasma.M_E, list(parts))
            one.evolve(steps)
            one.compute_rho()
            omom = particle_moments(one, oid)
            report["errors_vs_one_hip_brick"] = {n: worst_over_bricks(one.field_valid(n), n) for n in FIELDS}
            report["ekin_rel_vs_one_hip_brick"] = abs(sum(g["ekin"] for g in gathered) - omom["ekin"]) / omom["ekin"]
            one.close()
        ek = sum(g["ekin"] for g in gathered)
        report["ekin_rel"] = abs(ek - rmom["ekin"]) / rmom["ekin"]
        ap = np.sum([g["abs_p"] for g in gathered], axis=0)
        report["abs_p_rel"] = float(np.max(np.abs(ap - np.array(rmom["abs_momentum"])) / np.array(rmom["abs_momentum"])))
        json.dump(report, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
