#!/usr/bin/env python
"""Benchmark of the PIC hot path (BASELINE.json config 2 at N=1: 3-D uniform plasma 256^3,
8 ppc, Yee FDTD, order-3 shape, Esirkepov, Boris, 1-pass bilinear filter; weak scaling for
N>1: one 256^3 brick per GPU).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full PIC step of the hot path (gather+push, current deposition, filter +
guard-cell sum, EvolveB/E/B, guard-cell fills, periodic wrap/sort) over every particle and
cell resident in HBM.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of these boxes only supports dmabuf IPC: without this RCCL's buffer sharing between the ranks of a node
# fails with "hipIpcGetMemHandle: invalid argument" (already exported on the boxes; kept here for a bare environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_CU = 256          # MI355X_MICROARCH.md: 8 XCDs x 32 CUs
CLOCK_GHZ = 2.4     # MI355X_MICROARCH.md: peak engine clock
# what the deposition runs at: SQ_BUSY_CYCLES of its launch / its duration, and LDS-array cycles per ds_add_f64 wave
# instruction of the shipped kernel (profiles/round4/README.md; the figures of the PMC file are used when its stamp matches)
CLOCK_UNDER_LOAD_GHZ = 2.09
LDS_CYCLES_PER_ATOMIC = 8.7
# the N = 1 line an N > 1 run compares itself with when --n1-ms is not given (committed with the round's evidence)
N1_REFERENCE_LINE = os.path.join("profiles", "round6", "n1_reference_line.json")

# algorithmic bytes per unit, fp64 (SURVEY.md 8(d) / BASELINE.md section 3)
BYTES = {
    "EvolveB": 72.0,            # per cell per call
    "EvolveE": 96.0,            # per cell per call
    "GatherAndPush": (96.0, 48.0),      # per particle, per cell
    "CurrentDeposition": (56.0, 48.0),  # per particle, per cell
}


def device_uniform_plasma(n_cell, prob_lo, prob_hi, ppc, density, u_th, seed, box_lo, box_n, device):
    """warpx_amd.plasma.uniform_plasma evaluated on the device (same lattice formula:
    pos = prob_lo + (cell + (0.5+i)/ppc)*dx, weight = n*dV/ppc, u = u_th*c*N(0,1))."""
    import torch
    from warpx_amd import plasma
    f64 = torch.float64
    dx = [(prob_hi[d] - prob_lo[d]) / n_cell[d] for d in range(3)]
    nx, ny, nz = ppc
    nppc = nx * ny * nz
    ip = torch.arange(nppc, device=device)
    r = [(0.5 + (ip // (ny * nz)).to(f64)) / nx, (0.5 + ((ip % (ny * nz)) // nz).to(f64)) / ny,
         (0.5 + (ip % nz).to(f64)) / nz]
    ncells = box_n[0] * box_n[1] * box_n[2]
    n = ncells * nppc
    out = torch.empty((7, n), dtype=f64, device=device)
    cell = torch.arange(ncells, device=device)
    idx = [cell % box_n[0], (cell // box_n[0]) % box_n[1], cell // (box_n[0] * box_n[1])]
    for d in range(3):
        c = (idx[d] + box_lo[d]).to(f64)
        out[d] = (prob_lo[d] + (c[:, None] + r[d][None, :]) * dx[d]).reshape(-1)
    out[3] = density * dx[0] * dx[1] * dx[2] / nppc
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for d in range(3):
        out[4 + d] = torch.randn(n, dtype=f64, device=device, generator=g) * (u_th * plasma.C_LIGHT)
    return out


def kernel_sources_sha():
    """Fingerprint of the kernel sources the PMC traffic numbers were collected on (warpx_amd/csrc/*.hip, *.hpp)."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "warpx_amd", "csrc")
    for name in sorted(os.listdir(root)):
        if name in ("rccl_comm.hip", "runtime.hip"):   # the transport and the library runtime: no kernel in them
            continue
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:16]


PMC_TRAFFIC_FILE = os.path.join("profiles", "round6", "r6_pmc_counters.json")


def pmc_traffic(args):
    """HBM bytes per launch of each phase's kernel from the committed rocprofv3 PMC passes (scripts/pmc_traffic.py ->
    bench.PMC_TRAFFIC_FILE; bench.py cannot collect PMC counters itself: they need their own rocprofv3
    runs).  Used only when the file was collected on this workload AND on these kernel sources (its `sources_sha16`
    stamp equals kernel_sources_sha()): stale counters are dropped, not shown."""
    try:
        rec = json.load(open(os.path.join(ROOT, PMC_TRAFFIC_FILE)))
    except (OSError, ValueError):
        return {}, "no committed PMC pass"
    w = rec["workload"]
    same = (w["ncell"] == args.ncell and w["ppc"] == args.ppc and w["order"] == args.order and
            w["deposition"] == args.deposition and w["pusher"] == args.pusher and w["filter"] == (not args.no_filter)
            and w.get("sort_interval") == args.sort_interval)
    if not same:
        return {}, "the committed PMC pass is of another workload"
    if rec.get("sources_sha16") != kernel_sources_sha():
        return {}, (f"the committed PMC pass ({PMC_TRAFFIC_FILE}) was taken on other kernel sources "
                    f"({rec.get('sources_sha16')} vs {kernel_sources_sha()}): dropped")
    traffic = {k: (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e9 for k, v in rec["KiB_per_dispatch"].items()}
    traffic["_sq"] = {k: v.get("derived") for k, v in rec.get("sq", {}).items() if v.get("derived")}   # read by main()
    return traffic, f"rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE per dispatch, {PMC_TRAFFIC_FILE} (same kernel sources)"


def energies(sim, sid):
    """(field energy, kinetic energy) in joules: fields through the host reduction of warpx_amd.sim, particles on the
    device."""
    import torch
    from warpx_amd import plasma
    from warpx_amd.distributed import _as_tensor
    from warpx_amd.sim import field_energy
    ee, eb = field_energy(sim)
    v = sim.particle_view(sid)
    n = int(v.np)
    on_device = str(sim.lib.memory).startswith("cuda")   # host memory on the CPU execution model of the tests
    arr = {k: _as_tensor(int(getattr(v, k)), 8 * n, on_device).view(torch.float64) for k in ("w", "ux", "uy", "uz")}
    u2 = arr["ux"] ** 2 + arr["uy"] ** 2 + arr["uz"] ** 2
    gamma = torch.sqrt(1.0 + u2 / plasma.C_LIGHT ** 2)
    ekin = float(torch.sum(arr["w"] * plasma.M_E * u2 / (1.0 + gamma)))
    return ee + eb, ekin, n


def cpu_baseline():
    """The CPU oracle (our restatement of the reference's algorithms; the reference itself cannot be built here: AMReX
    is not on disk) timed on bounded samples, SURVEY.md 8(d) protocol -- median of 5 runs of 10 steps, thread count
    stated, every phase of the step on all threads (the guard exchanges too since round 3):
      * all the CPUs the box grants this process (OpenMP team = affinity mask capped by the cgroup quota,
        tests/oracle_lib.py::available_cpus -- 16 on the pool's GPU boxes, which show 256 logical CPUs; rounds 1 and 2
        ran a team of 128-256 on those 16 and reported "cores": 128; thread-private J scratch + accumulate, as the
        reference's CPU path) on a 128^3 sample of
        the bench workload (8 ppc, order 3, Esirkepov, Boris, filter on), with the oracle's per-phase timers;
      * one thread ("CPU serial") on BASELINE.json config 1 itself: 64^3, 1 ppc, order 1 (Examples/Tests/uniform_plasma)."""
    import statistics
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    from tests.oracle_lib import load_oracle
    from warpx_amd import _capi, plasma
    from warpx_amd.sim import WarpXSim
    orc = load_oracle()
    L = 40e-6

    def sample(n, ppc, order, steps, runs, threads):
        orc._set_num_threads(threads)
        n_cell = (n, n, n)
        parts = plasma.uniform_plasma(n_cell, (-L / 2,) * 3, (L / 2,) * 3, ppc, 1e25, 0.01, seed=12345)
        sim = WarpXSim(orc, n_cell, (-L / 2,) * 3, (L / 2,) * 3, nox=order, galerkin=1,
                       particle_pusher=_capi.PUSHER_BORIS, current_deposition=_capi.DEPOSIT_ESIRKEPOV, use_filter=1)
        sim.add_species(-plasma.Q_E, plasma.M_E, parts)
        npart = len(parts[0])
        del parts
        sim.evolve(1)
        rates, phases = [], None
        try:
            sim.enable_timers(True)
            sim.timers(reset=True)
        except Exception:
            pass
        for _ in range(runs):
            t0 = time.perf_counter()
            sim.evolve(steps)
            rates.append(npart * steps / (time.perf_counter() - t0))
        try:
            ph = sim.timers(reset=True)
            phases = {k: v[0] / (runs * steps) for k, v in ph.items() if v[1] > 0}
        except Exception:
            phases = None
        sim.close()
        return statistics.median(rates), min(rates), max(rates), npart, phases

    all_threads = int(orc._set_num_threads(0))
    t0 = time.perf_counter()
    med, lo, hi, npart, phases = sample(128, (2, 2, 2), 3, 10, 5, all_threads)
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    smed, slo, shi, snp, _ = sample(64, (1, 1, 1), 1, 10, 5, 1)
    t_ser = time.perf_counter() - t0
    orc._set_num_threads(all_threads)
    out = {"value": med, "unit": "particle-steps/s", "cores": all_threads, "kind": "port",
           "sample": f"128^3 cells, 8 ppc, order 3, Esirkepov, Boris, filter on ({npart} particles): median of 5 runs of "
                     f"10 steps (min {lo:.3e}, max {hi:.3e}); {t_all:.1f} s wall incl. set-up; cell-updates/s = "
                     f"{med / 8.0:.3e}",
           "serial": {"value": smed, "unit": "particle-steps/s", "cores": 1,
                      "sample": f"BASELINE.json config 1: 64^3 cells, 1 ppc, order 1, Esirkepov, Boris, filter on ({snp} "
                                f"particles): median of 5 runs of 10 steps (min {slo:.3e}, max {shi:.3e}); {t_ser:.1f} s "
                                f"wall incl. set-up; cell-updates/s = {smed:.3e}"}}
    if phases:
        out["ms_per_step_by_phase"] = phases
    return out


def stencil_microbench(sim, reps=20):
    """EvolveB / EvolveE launched `reps` times back to back on the simulation's own fields between
    one pair of events on the kernels' stream (dt = 0 leaves the fields unchanged): the per-launch
    average without the event overhead that a single 0.25-ms launch carries."""
    import ctypes as C
    import torch
    from warpx_amd import _capi
    lib = sim.lib
    E = (_capi.FieldView * 3)(*[sim.field_view(n) for n in ("Ex", "Ey", "Ez")])
    B = (_capi.FieldView * 3)(*[sim.field_view(n) for n in ("Bx", "By", "Bz")])
    J = (_capi.FieldView * 3)(*[sim.field_view(n) for n in ("jx", "jy", "jz")])
    dinv = (C.c_double * 3)(*[1.0 / d for d in sim.dx])
    out = {}
    for name, call in (("EvolveB", lambda: lib.evolve_b(E, B, 0.0, dinv, None)),
                       ("EvolveE", lambda: lib.evolve_e(E, B, J, 0.0, dinv, None))):
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / reps
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncell", type=int, default=256, help="cells per brick edge")
    ap.add_argument("--ppc", type=int, default=2, help="particles per cell per direction")
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--deposition", choices=["esirkepov", "direct"], default="esirkepov")
    ap.add_argument("--pusher", choices=["boris", "vay"], default="boris")
    ap.add_argument("--no-filter", action="store_true")
    ap.add_argument("--sort-interval", type=int, default=2,
                    help="cell sort every N steps (warpx.sort_intervals).  With the sort folded into the push as ONE special "
                         "push per cycle (round 6: the sort step's push scatters with the last record and counts the next) "
                         "2 and 3 are level: 12.6-13.0 / 12.8-13.0 ms per step on one box, 13.8 at 1, 13.1 at 4 "
                         "(profiles/round6/README.md)")
    ap.add_argument("--preroll", type=int, default=40,
                    help="untimed steps before the warmup: the regular-lattice start is atypically cheap "
                         "(no particle crosses a cell for ~20 steps), the timed region must see the "
                         "thermalised steady state")
    ap.add_argument("--overlap", type=int, default=-1,
                    help="1: the guard exchanges of E+B and of J on a second stream, behind the push of the interior "
                         "tiles and B's half update; 0: everything on one stream (the fallback); default: 1 for N > 1 "
                         "(what north_star asks for; both schedules give the same fields, tests/test_multibrick_*)")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="N > 1: rccl = the library's own transport (ncclSend / ncclRecv groups on the library's streams, "
                         "csrc/rccl_comm.hip); torch = torch.distributed P2P through Python callbacks")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: nccl = RCCL over xGMI, one rank per GPU (the contract's path); gloo = the slabs staged "
                         "through pinned host memory (D2H -> gloo -> H2D on the exchange's stream): the only way several "
                         "ranks can share ONE GPU (RCCL refuses two ranks on a device) -- with --ranks-per-gpu it runs the "
                         "whole multi-process path (bricks, count round, leaver lists, overlapped schedule) on a 1-GPU box")
    ap.add_argument("--ranks-per-gpu", type=int, default=1,
                    help="with --backend gloo: rank r runs on device LOCAL_RANK // ranks_per_gpu; the line's n_gpus is "
                         "the number of devices, n_ranks the number of bricks")
    ap.add_argument("--single-precision-comms", action="store_true",
                    help="N > 1: warpx.do_single_precision_comms -- float on the wire of the guard exchanges (half the xGMI "
                         "bytes; the reference's own lever, ablastr/utils/Communication.cpp:37-56); off in the headline line")
    ap.add_argument("--deposit-acc", choices=["f64", "f32"], default="f64",
                    help="accumulators of the LDS deposition tiles: f64 = ds_add_f64 (the parity build, the headline line); "
                         "f32 = ds_add_f32, the throughput variant of BASELINE.json's north_star (2e-6 gate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phase-pass", action="store_true")
    ap.add_argument("--sync-each-call", action="store_true",
                    help="synchronise the momenta at the end of every evolve call like WarpX::Evolve(n) does (the timed "
                         "region then contains one PushP(-dt/2) / PushP(+dt/2) pair); default: one run advanced in pieces")
    ap.add_argument("--no-sanity", action="store_true", help="skip the energy / particle-count figures around the timed steps")
    ap.add_argument("--n1-ms", type=float, default=0.0,
                    help="N > 1: ms per step of the same per-GPU workload on one GPU, for weak_scaling.efficiency "
                         "(default: the committed line " + N1_REFERENCE_LINE + ")")
    ap.add_argument("--dry-comm", action="store_true",
                    help="after the pre-roll, run ONLY the step's neighbour exchanges (FillBoundary E+B, SumBoundary J, "
                         "Redistribute) on the run's own arrays and print their milliseconds next to the message counts "
                         "and bytes, instead of the bench line: separates 'RCCL is slow' from 'the step is slow' on a "
                         "first multi-GPU run")
    args = ap.parse_args()

    import torch
    from warpx_amd import _capi, load_product, plasma
    from warpx_amd.sim import WarpXSim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.ranks_per_gpu > 1 and args.backend != "gloo":
        raise SystemExit("--ranks-per-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    local_dev = local_rank // max(args.ranks_per_gpu, 1)
    torch.cuda.set_device(local_dev)
    device = f"cuda:{local_dev}"
    red_dev = "cpu" if args.backend == "gloo" else device    # where the few-number reductions of the bench line live
    lib = load_product()  # raises when the HIP library is missing: no fallback

    transport = None
    nbricks, coord = (1, 1, 1), (0, 0, 0)
    if world > 1:
        import torch.distributed as dist
        from warpx_amd.distributed import RcclBrickTransport, TorchBrickTransport, brick_coord, brick_layout
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))
        nbricks = brick_layout(world)
        coord = brick_coord(rank, nbricks)
        transport = None
        if args.backend == "gloo":
            transport = TorchBrickTransport(on_device=True, staged=True)
        elif args.transport == "rccl":
            try:
                transport = RcclBrickTransport(lib, timing=False)   # timed in the phase pass below, not in the headline
            except Exception as e:   # keep the run alive on the Python transport, and say so
                print(f"[bench] rank {rank}: in-library RCCL transport unavailable ({e}); using torch.distributed", flush=True)
        if transport is None:
            transport = TorchBrickTransport(on_device=True)

    nb = args.ncell
    n_cell = tuple(nb * nbricks[d] for d in range(3))
    L0 = 40e-6
    prob_lo = tuple(-L0 * nbricks[d] / 2 for d in range(3))
    prob_hi = tuple(+L0 * nbricks[d] / 2 for d in range(3))
    depos = _capi.DEPOSIT_ESIRKEPOV if args.deposition == "esirkepov" else _capi.DEPOSIT_DIRECT
    pusher = _capi.PUSHER_BORIS if args.pusher == "boris" else _capi.PUSHER_VAY
    sim = WarpXSim(lib, n_cell, prob_lo, prob_hi, nox=args.order, galerkin=None, particle_pusher=pusher,
                   current_deposition=depos, use_filter=0 if args.no_filter else 1, cfl=1.0,
                   sort_interval=args.sort_interval, nbricks=nbricks, coord=coord,
                   comm=transport.comm if transport else None,
                   overlap_halo=(1 if world > 1 else 0) if args.overlap < 0 else args.overlap)
    if args.single_precision_comms and world > 1:
        sim.set_single_precision_comms(True)
    box_lo = tuple(coord[d] * nb for d in range(3))
    parts = device_uniform_plasma(n_cell, prob_lo, prob_hi, (args.ppc,) * 3, 1e25, 0.01, 12345 + rank,
                                  box_lo, (nb,) * 3, device)
    from warpx_amd.containers import ParticleArrays
    pa = ParticleArrays(parts.shape[1], device)
    pa.data = parts
    sid0 = sim.add_species(-plasma.Q_E, plasma.M_E, pa)
    if args.deposit_acc == "f32":
        sim.set_deposit_accumulator(sid0, _capi.ACC_FP32)
    np_local = parts.shape[1]
    del parts, pa
    torch.cuda.empty_cache()
    ncells_local = nb ** 3

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    # The K timed steps are steps in the middle of a run: WarpX::Evolve(n) pushes the momenta back by dt/2 in its first
    # step and forward again in its last one (WarpXEvolve.cpp:142-145, 222-226), once per run, not per step; called once
    # per timing interval that pair (two extra field gathers over all particles) would be charged to the K steps.
    # --sync-each-call times Evolve(K) with the pair inside.
    if not args.sync_each_call:
        sim.set_synchronize_at_end(False)
    if args.preroll > 0:
        sim.evolve(args.preroll)
    sim.evolve(args.warmup)
    if args.dry_comm:
        rccl = transport is not None and hasattr(transport, "stats")
        if rccl:
            transport.stats(reset=True)
        reps = 20
        barrier()
        ms = sim.dry_comm(reps)
        st = transport.stats() if rccl else None
        vals = [ms[k] for k in ("FillBoundaryEB", "SumBoundaryJ", "Redistribute", "all_three")]
        if world > 1:
            t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            vals = [float(v) for v in t.tolist()]
        if rank == 0:
            calls = 4 * reps   # each of the three exchanges ran `reps` times alone and `reps` times in a row
            line = {"dry_comm": True, "n_gpus": world // max(args.ranks_per_gpu, 1), "n_ranks": world, "bricks": list(nbricks), "cells_per_gpu": ncells_local, "particles_per_gpu": np_local,
                    "ms_per_call_max_over_ranks": dict(zip(("FillBoundaryEB", "SumBoundaryJ", "Redistribute", "all_three"), vals)),
                    "reps": reps, "transport": "rccl (in-library)" if rccl else ("none (one brick)" if transport is None else
                                                                     ("torch.distributed gloo, host-staged" if args.backend == "gloo" else "torch.distributed")),
                    "note": "host clock around a stream sync, nothing computed between the exchanges; a step issues each of the "
                            "three once (E+B before the push, J after the deposition, particles at the end)"}
            if st:
                line["rank0_messages_per_exchange"] = st["n_messages"] / max(st["n_exchanges"], 1)
                line["rank0_MB_sent_per_round_of_three"] = st["bytes_sent"] / (2 * reps) / 1e6
                line["rank0_count_rounds"] = st["n_count_exchanges"]
            print(json.dumps(line), flush=True)
        sim.close()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    e_before = energies(sim, 0) if not args.no_sanity else None
    barrier()
    t0 = time.perf_counter()
    sim.evolve(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    sanity = None
    if e_before is not None:
        e_after = energies(sim, 0)
        tot = [e_before[0], e_before[1], float(e_before[2]), e_after[0], e_after[1], float(e_after[2])]
        if world > 1:   # the job's totals, not rank 0's share
            t = torch.tensor(tot, dtype=torch.float64, device=red_dev)
            torch.distributed.all_reduce(t)
            tot = [float(v) for v in t.tolist()]
        tot0, tot1 = tot[0] + tot[1], tot[3] + tot[4]
        drift = (tot1 - tot0) / tot0
        expected = int(np_local * world)
        ok = int(tot[5]) == expected and int(tot[2]) == expected and abs(drift) < 1e-2
        sanity = {"ok": ok, "particles_after": int(tot[5]), "particles_before": int(tot[2]), "particles_expected": expected,
                  "total_energy_drift_over_timed_steps": drift,
                  "kinetic_energy_J": tot[4], "field_energy_J": tot[3],
                  "note": "sums over all ranks, before / after the timed steps"
                          + ("" if args.sync_each_call else "; kinetic energy from the leap-frog momenta (half a step "
                             "behind the fields) at both ends")}
        if not ok and rank == 0:
            print(f"[bench] SANITY FAILED: particles {int(tot[2])} -> {int(tot[5])} (expected {expected}), "
                  f"energy drift {drift:.3e}", file=sys.stderr, flush=True)
    # second, short pass with per-phase HIP-event timers (on the stream the kernels run on); the transport's event pairs
    # around every exchange are on in this pass only
    phases = {}
    nph = 0
    rccl = transport is not None and hasattr(transport, "stats")
    st_head = transport.stats(reset=True) if rccl else None
    if not args.no_phase_pass:
        if rccl:
            transport.set_timing(True)
        sim.enable_timers(True)
        sim.timers(reset=True)
        # whole sort cycles, so that every phase's average has its share of sort steps (and of the steps whose push
        # records or applies the sort): 4 steps, rounded up to a multiple of the interval
        cyc = max(args.sort_interval, 1)
        nph = cyc * ((4 + cyc - 1) // cyc)
        while sim.istep % cyc != 0:   # start the pass at a cycle boundary
            sim.evolve(1)
        torch.cuda.synchronize()
        sim.timers(reset=True)
        sim.evolve(nph)
        torch.cuda.synchronize()
        phases = sim.timers(reset=True)
        sim.enable_timers(False)
        if rccl:
            transport.set_timing(False)

    comm_stats = None
    if rccl:
        st = transport.stats()
        nsteps_head = max(sim.istep - nph, 1)
        comm_stats = {"transport": "rccl (in-library)", "exchanges_per_step": st_head["n_exchanges"] / nsteps_head,
                      "messages_per_step": st_head["n_messages"] / nsteps_head,
                      "MB_sent_per_step": st_head["bytes_sent"] / nsteps_head / 1e6,
                      "count_exchanges_per_step": st_head["n_count_exchanges"] / nsteps_head,
                      "ms_per_exchange": st["timed_ms"] / max(st["timed_exchanges"], 1),
                      "exchange_ms_per_step": st["timed_ms"] / max(nph, 1),
                      "note": "counts over the whole run up to the end of the timed steps; milliseconds from HIP events "
                              "around every exchange in the separate phase pass (rank 0)"}
    elif transport is not None:
        comm_stats = {"transport": "torch.distributed (Python callbacks)" + (", gloo, slabs staged through pinned host memory"
                                                                             if args.backend == "gloo" else ""),
                      "exchanges_per_step": transport.n_exchanges / max(sim.istep, 1),
                      "MB_sent_per_step": transport.bytes_sent / max(sim.istep, 1) / 1e6}
    if rank == 0:
        total_particles = np_local * world
        total_cells = ncells_local * world
        pps = total_particles * args.steps / elapsed
        cps = total_cells * args.steps / elapsed
        kernels = {}
        dominant = None
        traffic, traffic_note = pmc_traffic(args) if world == 1 else ({}, "N > 1")
        sq_counters = traffic.pop("_sq", {})   # SQ passes of the shipped particle kernels, same file, same stamp
        micro = {} if args.no_phase_pass else stencil_microbench(sim)
        for name, (ms, cnt) in phases.items():
            if cnt == 0 or ms <= 0:
                continue
            avg_ms = ms / cnt
            entry = {"avg_ms": avg_ms, "launches": int(cnt)}
            if name in ("EvolveB", "EvolveE"):
                algo_bytes = BYTES[name] * ncells_local
                if name in micro:
                    entry["avg_ms_back_to_back"] = micro[name]
                    entry["hbm_frac_back_to_back"] = algo_bytes / 1e9 / (micro[name] * 1e-3) / HBM_PEAK_GBS
            elif name in ("GatherAndPush", "CurrentDeposition"):
                bp, bc = BYTES[name]
                algo_bytes = bp * np_local + bc * ncells_local
            else:
                algo_bytes = None
            if algo_bytes is not None:
                entry["algorithmic_GB"] = algo_bytes / 1e9
                entry["achieved_GBs"] = algo_bytes / 1e9 / (avg_ms * 1e-3)
                entry["hbm_frac"] = entry["achieved_GBs"] / HBM_PEAK_GBS
            if name in traffic:
                entry["pmc_traffic_GB"] = traffic[name]
            kernels[name] = entry
        # share of the step per phase (PushP of the (de)synchronisation is folded in GatherAndPush)
        if kernels:
            tot = {k: v["avg_ms"] * v["launches"] for k, v in kernels.items()}
            dominant = max((k for k in tot if "achieved_GBs" in kernels[k]), key=lambda k: tot[k])
        roofline = None
        if dominant:
            k = kernels[dominant]
            roofline = {"kernel": dominant, "bound": "hbm", "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": k["hbm_frac"], "traffic": k.get("pmc_traffic_GB"),
                        "traffic_unit": "GB per launch; " + traffic_note,
                        "note": "the bench contract's roofline is the HBM one; the particle kernels sit far below it "
                                "because they are limited by the fp64 VALU and LDS pipes (SURVEY.md 8(d), "
                                "profiles/round3/README.md) -- see `limiter` and the floors beside it; the HBM-bound "
                                "stencils are listed under kernels"}
            if dominant == "CurrentDeposition" and args.deposition == "esirkepov" and args.order == 3:
                # order 3 Esirkepov: 2 particles share one set of 4x4x4(+1) rows; 144 ds_add_f64 per pair and lane
                # (deposit_body.hpp).  One ds_add_f64 wave-instruction occupies a CU's LDS pipe for 8 cycles when
                # conflict-free (scripts/microbench/lds_atomic_bench.hip, profiles/round3/lds_atomic_microbench.txt).
                # at the clock the kernel runs at and the cycles its LDS array spends per instruction (both measured:
                # 2.09 GHz, 8.7; round 3's line used the peak clock and the microbenchmark's 8.0 and read 1.97 ms)
                d = sq_counters.get(dominant) or {}
                cyc = d.get("lds_array_cycles_per_lds_instruction", LDS_CYCLES_PER_ATOMIC)
                wave_instr = np_local / 2 * 144 / 64
                floor_ms = wave_instr / N_CU * cyc / (CLOCK_UNDER_LOAD_GHZ * 1e9) * 1e3
                roofline["lds_atomic_floor_ms"] = floor_ms
                roofline["lds_atomic_floor_frac"] = floor_ms / k["avg_ms"]
                roofline["lds_atomic_floor_note"] = (f"144 ds_add_f64 per merged pair and lane at {cyc:.2f} LDS-array cycles per "
                                                     f"wave instruction and {CLOCK_UNDER_LOAD_GHZ} GHz under load")
            if dominant in ("GatherAndPush", "CurrentDeposition"):
                d = sq_counters.get(dominant)
                if d:   # the committed SQ passes of the kernel that ships, taken on these kernel sources
                    roofline["limiter"] = (f"fp64 VALU {100 * d['valu_busy_frac']:.0f} % busy + LDS array "
                                           f"{100 * d['lds_array_busy_frac']:.0f} % busy over the launch (rocprofv3 SQ passes, "
                                           f"{PMC_TRAFFIC_FILE})")
                    roofline["sq_counters"] = {key: d[key] for key in ("valu_busy_frac", "lds_array_busy_frac", "lds_conflict_frac",
                                                                       "lds_array_cycles_per_lds_instruction", "wave_waiting_frac",
                                                                       "wave_issue_stall_frac") if key in d}
                else:
                    roofline["limiter"] = "fp64 VALU + LDS pipe (no SQ pass on these kernel sources: " + traffic_note + ")"
            for name, d in sq_counters.items():
                if name in kernels:
                    kernels[name]["sq_counters"] = d
        out = {
            "metric": "particle_steps_per_s", "value": pps, "unit": "particle-steps/s",
            "cell_updates_per_s": cps,
            "n_gpus": world // max(args.ranks_per_gpu, 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.deposit_acc == "f64" else "f64 (fp32 deposition tiles)",
            "data": "synthetic",
            "config": {"workload": f"3D uniform_plasma {n_cell[0]}x{n_cell[1]}x{n_cell[2]}, "
                                   f"{args.ppc ** 3} ppc, Yee FDTD, order-{args.order} shape, {args.deposition}, "
                                   f"{args.pusher}, filter {'off' if args.no_filter else 'on'}",
                       "cells_per_gpu": ncells_local, "particles_per_gpu": np_local,
                       "bricks": list(nbricks), "n_ranks": world, "ranks_per_gpu": args.ranks_per_gpu,
                       "backend": args.backend if world > 1 else None, "sort_interval": args.sort_interval,
                       "single_precision_comms": bool(args.single_precision_comms and world > 1),
                       "preroll_steps": args.preroll, "overlap_halo": bool(sim.halo_overlap),
                       "momentum_synchronisation": "inside the timed region (Evolve(K) as one run)" if args.sync_each_call
                       else "outside the timed region (the K steps are consecutive steps of one longer run)"},
            "roofline": roofline,
            "kernels": kernels,
        }
        if comm_stats:
            out["exchange"] = comm_stats
        if world > 1:
            # What the field exchanges of a step put on the wire at most (allocated guard depths; the named depths of the
            # exchanges are smaller or equal): per split direction two face slabs of E, B (FillBoundary) and J (SumBoundary).
            pred = 0.0
            for name in ("Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz"):
                v = sim.field_view(name)
                for d in range(3):
                    if nbricks[d] > 1:
                        slab = 8.0 * v.ng[d]
                        for e in range(3):
                            if e != d:
                                slab *= v.n[e]
                        pred += 2 * slab
            out.setdefault("exchange", {})["field_MB_per_step_upper_bound"] = pred / 1e6
            out["exchange"]["note_links"] = ("on 2 x 2 x 2 both faces of a direction go to the same peer: a third of these "
                                             "bytes per xGMI link and step, direction after direction")
            # weak scaling against the same per-GPU workload on one GPU
            n1_ms, src = args.n1_ms, "--n1-ms"
            if n1_ms <= 0.0:
                try:
                    ref = json.load(open(os.path.join(ROOT, N1_REFERENCE_LINE)))
                    same = (ref["config"]["cells_per_gpu"] == ncells_local and ref["config"]["particles_per_gpu"] == np_local
                            and ref["config"]["workload"].split(",", 1)[1] == out["config"]["workload"].split(",", 1)[1])
                    if same:
                        n1_ms, src = float(ref["ms_per_step"]), N1_REFERENCE_LINE
                    else:
                        src = N1_REFERENCE_LINE + " is another workload: pass --n1-ms"
                except Exception as e:
                    src = f"no N = 1 line to compare with ({e}): pass --n1-ms"
            out["weak_scaling"] = {"n1_ms_per_step": n1_ms if n1_ms > 0 else None,
                                   "efficiency": (n1_ms / out["ms_per_step"]) if n1_ms > 0 else None,
                                   "n1_source": src,
                                   "exchange_ms_per_step": (comm_stats or {}).get("exchange_ms_per_step"),
                                   "note": "efficiency = ms per step of this per-GPU workload on one GPU / ms per step here "
                                           "(the driver computes its own from its N = 1 run; target >= 0.70)"}
            if args.ranks_per_gpu > 1:   # the ranks share one device: their kernels take turns on it -- not a scaling figure
                out["weak_scaling"]["efficiency"] = None
                out["weak_scaling"]["note"] = (f"{args.ranks_per_gpu} ranks share each GPU (host-staged gloo): this line proves the "
                                               "multi-process path end to end, it is not a scaling measurement")
        if sanity:
            out["sanity"] = sanity
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the oracle is test infrastructure; never fail the bench on it
                out["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    sim.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    if sanity is not None and not sanity["ok"]:
        raise SystemExit(3)   # a line whose particles or energy went wrong is not a measurement


if __name__ == "__main__":
    main()
