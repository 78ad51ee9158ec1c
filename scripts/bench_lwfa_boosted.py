"""BASELINE.json config 5 on ONE MI355X: a 3-D laser-wakefield stage in a boosted frame, as a throughput line.

    python scripts/bench_lwfa_boosted.py [--ncell 256 256 512] [--ppc 2] [--steps 60] [--fill-steps 0]

The deck is tests/decks/laser_wakefield_boosted_3d.inputs (gamma = 5 along z, CKC solver, Vay pusher, order-3 shapes,
bilinear filter, NCI corrector on E and B before the gather, window moving with c, PEC along z, Gaussian antenna, plasma
injected continuously at the drifting front) at a size that loads the GPU: every kernel the boosted path adds to the
uniform-plasma headline (CKC EvolveB, the Godfrey filter, the antenna push, window shift + injection + the sort behind it)
is inside the timed steps.  The window is first filled with plasma (the front travels through it at (1 + beta) c), then
`--steps` steps are timed between device synchronisations; a second, short pass with the per-phase HIP-event timers gives
the table.  One JSON line: particle-steps/s and cell-updates/s of the timed steps (the particle count is the mean of its
values before and after them: the window gains and loses particles every step)."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncell", type=int, nargs=3, default=[256, 256, 512])
    ap.add_argument("--ppc", type=int, default=2, help="particles per cell per direction")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--fill-steps", type=int, default=0, help="steps before the timed ones (0: until the plasma front has crossed the window)")
    ap.add_argument("--sort-interval", type=int, default=1,
                    help="a streaming plasma moves 0.76 cells per step: sort every step (305 against 321 ms per step at 3, r5m)")
    args = ap.parse_args()

    import torch
    from warpx_amd import load_product
    from warpx_amd.sim import WarpXSim

    torch.cuda.set_device(0)
    lib = load_product()
    deck = os.path.join(ROOT, "tests", "decks", "laser_wakefield_boosted_3d.inputs")
    nx, ny, nz = args.ncell
    ov = [f"amr.n_cell={nx} {ny} {nz}", f"electrons.num_particles_per_cell_each_dim={args.ppc} {args.ppc} {args.ppc}",
          f"warpx.sort_intervals={args.sort_interval}", "max_step=1000000"]
    sim = WarpXSim.from_inputs(lib, deck, overrides=ov)
    sim.set_synchronize_at_end(False)
    # the front of the plasma crosses the window at (1 + beta) c: L_boost / ((1 + beta) c dt) steps
    gamma = 5.0
    beta = math.sqrt(1.0 - 1.0 / gamma ** 2)
    fill = args.fill_steps
    if fill <= 0:
        # boosted lengths: Lz' = Lz_lab gamma (1 + beta); dt = min(dx) / c (CKC, cfl 1)
        lz_boost = 16e-6 * gamma * (1.0 + beta)
        dx = min(60e-6 / nx, 60e-6 / ny, lz_boost / nz)
        fill = int(1.15 * lz_boost / ((1.0 + beta) * dx)) + 1
    t0 = time.perf_counter()
    sim.evolve(fill)
    torch.cuda.synchronize()
    fill_s = time.perf_counter() - t0
    np_before = int(sim.particle_view(0).np)
    t0 = time.perf_counter()
    sim.evolve(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    np_after = int(sim.particle_view(0).np)
    np_mean = 0.5 * (np_before + np_after)
    ncells = nx * ny * nz
    # per-phase table: whole sort cycles
    cyc = max(args.sort_interval, 1)
    nph = cyc * ((6 + cyc - 1) // cyc)
    prof = None
    if os.environ.get("WXA_PRODUCT_LIB", "").endswith("_prof.so"):   # a WXA_DEPOSIT_PROFILE build: the tile kernel's phase clocks
        import ctypes as C
        raw = C.CDLL(os.environ["WXA_PRODUCT_LIB"])
        prof = (C.c_ulonglong * 16)()
        raw.wxa_debug_deposit_profile(prof, 1)
        bins = (C.c_ulonglong * 128)()
        raw.wxa_debug_deposit_profile_bins(bins, 1)
    sim.enable_timers(True)
    sim.timers(reset=True)
    sim.evolve(nph)
    torch.cuda.synchronize()
    phases = sim.timers(reset=True)
    sim.enable_timers(False)
    if prof is not None:
        raw.wxa_debug_deposit_profile(prof, 1)
        raw.wxa_debug_deposit_profile_bins(bins, 1)
        allc = sum(bins[4 * b] for b in range(32))
        print("workgroups by the particles of their share (log2 bins): share of all workgroup cycles, workgroups per launch, "
              "mean and longest in kilocycles, particles per launch", file=sys.stderr)
        for b in range(32):
            if bins[4 * b + 1]:
                print("   2^%-2d  %5.1f %%  %8.0f  mean %8.1f  longest %8.1f  particles %.3e" % (
                    b, 100.0 * bins[4 * b] / allc, bins[4 * b + 1] / nph, bins[4 * b] / bins[4 * b + 1] / 1e3, bins[4 * b + 2] / 1e3,
                    bins[4 * b + 3] / nph), file=sys.stderr)
        tot = sum(prof[:6])
        print("deposition phase clocks (workgroup cycles, share):", [round(prof[i] / tot, 3) for i in range(6)], file=sys.stderr)
        print("chunks %d, all lane pairs on one frame %d, all quads %d, second particles apart %d, lanes sharing %d of %d with a particle"
              % tuple(int(prof[i]) for i in range(10, 16)), "in %d launches" % nph, file=sys.stderr)
        if prof[8]:
            print("wave 0 per chunk: loads %.0f cycles, body %.0f cycles, %.0f chunks per launch; in its loop %.2f of phase 2"
                  % (prof[6] / prof[8], prof[7] / prof[8], prof[8] / nph, prof[9] / max(prof[2], 1)), file=sys.stderr)
    kernels = {}
    for name, (ms, cnt) in phases.items():
        if cnt:
            kernels[name] = {"avg_ms": ms / cnt, "launches_per_step": cnt / nph, "ms_per_step": ms / nph}
    if "EvolveB" in kernels:   # the CKC update of B: 72 B per cell and call like Yee's
        k = kernels["EvolveB"]
        k["algorithmic_GB"] = 72.0 * ncells / 1e9
        k["hbm_frac"] = k["algorithmic_GB"] / (k["avg_ms"] * 1e-3) / HBM_PEAK_GBS
    if "EvolveE" in kernels:
        k = kernels["EvolveE"]
        k["algorithmic_GB"] = 96.0 * ncells / 1e9
        k["hbm_frac"] = k["algorithmic_GB"] / (k["avg_ms"] * 1e-3) / HBM_PEAK_GBS
    out = {"metric": "particle_steps_per_s", "value": np_mean * args.steps / elapsed, "unit": "particle-steps/s",
           "cell_updates_per_s": ncells * args.steps / elapsed, "n_gpus": 1, "steps": args.steps,
           "ms_per_step": elapsed / args.steps * 1e3, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"3D laser-wakefield in a frame boosted by gamma = 5 (BASELINE config 5 on one GPU), {nx}x{ny}x{nz} "
                                  f"cells, {args.ppc ** 3} ppc electrons, CKC, Vay, order-3 shape, Esirkepov, filter on, NCI corrector, "
                                  "moving window, PEC along z, Gaussian antenna, continuous injection",
                      "deck": "tests/decks/laser_wakefield_boosted_3d.inputs", "sort_interval": args.sort_interval,
                      "fill_steps": fill, "fill_seconds": fill_s, "particles_before": np_before, "particles_after": np_after},
           "kernels": kernels,
           "note": "GatherAndPush includes the NCI corrector's six filter launches per species and step and the antenna push; "
                   "Redistribute includes the particle walls, the window shift's injection and the sort behind it"}
    print(json.dumps(out), flush=True)
    sim.close()


if __name__ == "__main__":
    main()
