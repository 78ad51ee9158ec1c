/*
 * warpx_amd.h -- C-ABI of the MI355X-native PIC inner loop.
 *
 * This is the drop-in boundary: a plain-C interface
 * (pointers + sizes, no torch / AMReX types) underneath the C++17 operator
 * surface in warpx_amd/csrc/host/ that mirrors the reference's
 * WarpXParticleContainer / MultiFab / FiniteDifferenceSolver methods.
 *
 * Every entry point names the reference interface it replaces
 * (paths relative to the WarpX checkout @ 2024-10-24).
 *
 * Conventions
 *  - all device pointers; all kernels are asynchronous on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream);
 *  - return value: WXA_OK (0) or a negative wxa_status;
 *    no entry point allocates unless it takes a wxa_workspace;
 *  - array layout = AMReX Array4: Fortran order, i fastest, guards included
 *    (reference usage e.g. Source/FieldSolver/FiniteDifferenceSolver/EvolveE.cpp:148-156);
 *  - fp64 throughout (amrex::Real = amrex::ParticleReal = double, the
 *    reference's default build precision, CMakeLists.txt:109-118).
 */
#ifndef WARPX_AMD_H_
#define WARPX_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (reference: WARPX_ABORT_WITH_MESSAGE -> amrex::Abort;
 *      here an int so a foreign-language host can raise its own error) ---- */
typedef enum wxa_status {
    WXA_OK = 0,
    WXA_ERR_INVALID_ARG = -1,   /* bad pointer / order / algo / box          */
    WXA_ERR_HIP = -2,           /* a HIP runtime call or kernel launch failed */
    WXA_ERR_UNSUPPORTED = -3,   /* valid in the reference, outside this path  */
    WXA_ERR_NOMEM = -4,
    WXA_ERR_COMM = -5
} wxa_status;

/* ---- enums crossing the boundary; same enumerator order as the reference
 *      (Source/Utils/WarpXAlgorithmSelection.H:72-84, Source/Evolve/WarpXDtType.H:10-15) */
enum { WXA_PUSHER_BORIS = 0, WXA_PUSHER_VAY = 1, WXA_PUSHER_HC = 2,   /* algo.particle_pusher: boris, vay, higuera */
       WXA_PUSHER_BORIS_RR = 3 };   /* <species>.do_classical_radiation_reaction = 1: Boris + radiation reaction */
enum { WXA_DEPOSIT_ESIRKEPOV = 0, WXA_DEPOSIT_DIRECT = 1 };
enum { WXA_DT_FULL = 0, WXA_DT_FIRST_HALF = 1, WXA_DT_SECOND_HALF = 2 };

/* One staggered field component of one brick (= one amrex::FArrayBox of a
 * MultiFab with its guard cells; Array4 semantics).
 * Point (i,j,k) lives at p[(i-lo[0]) + (j-lo[1])*jstride + (k-lo[2])*kstride].
 * Valid (non-guard) index box: [lo+ng, lo+n-ng) per direction; it holds
 * ncell+stag points (a nodal direction has one more point than cells,
 * Source/WarpX.cpp:2117-2125). */
typedef struct wxa_field_view {
    double* p;
    int32_t lo[3];     /* index of the first allocated point (guards included) */
    int32_t n[3];      /* allocated points per direction (guards included)     */
    int32_t ng[3];     /* guard points per side                                */
    int32_t stag[3];   /* 0 = cell-centred, 1 = nodal along that direction     */
    int64_t jstride;   /* >= n[0]; rows may be padded for 128-B alignment      */
    int64_t kstride;   /* >= jstride*n[1]                                      */
} wxa_field_view;

/* Pure-SoA particle tile, PIdx order x,y,z,w,ux,uy,uz (+ idcpu)
 * (Source/Particles/NamedComponentParticleContainer.H:23-40). u = gamma*v [m/s]. */
typedef struct wxa_particle_view {
    double* x; double* y; double* z; double* w;
    double* ux; double* uy; double* uz;
    uint64_t* idcpu;   /* may be NULL */
    int64_t np;
} wxa_particle_view;

/* Index-space origin used by gather and deposition:
 * grid coordinate = (x - xyzmin)*dinv >= 0, array index = lo + that.
 * The reference builds it from the tile box grown by the guard depth
 * (Source/Particles/PhysicalParticleContainer.cpp:2575-2601,
 *  Source/Particles/WarpXParticleContainer.cpp:439,477-479, Source/WarpX.cpp:2851-2875). */
typedef struct wxa_grid_geom {
    double xyzmin[3];
    double dinv[3];
    int32_t lo[3];
} wxa_grid_geom;

/* Per-device scratch (sort permutation, histograms, tile offsets, staging).
 * Opaque; owned by the caller through create/destroy. */
typedef struct wxa_workspace wxa_workspace;

wxa_status wxa_workspace_create(wxa_workspace** ws);
void       wxa_workspace_destroy(wxa_workspace* ws);
/* particles.E_external_particle / particles.B_external_particle with *_ext_particle_init_style = constant
 * (members m_E_external_particle / m_B_external_particle of the container,
 * Source/Particles/PhysicalParticleContainer.cpp:2589-2596,2705-2710): added to the gathered fields by
 * wxa_gather_push_ws / wxa_gather_push_part when they are handed this workspace.  Zero by default. */
wxa_status wxa_workspace_set_external_particle_fields(wxa_workspace* ws, const double E[3], const double B[3]);
/* particles.E_ext_particle_init_style / B_ext_particle_init_style = repeated_plasma_lens
 * (GetExternalEBField::operator(), Source/Particles/Gather/GetExternalFields.H:137-189; parameters read in
 * MultiParticleContainer::ReadParameters, MultiParticleContainer.cpp:210-260): lens i occupies
 * [i period + starts[i], ... + lengths[i]) of the lab-frame z axis and acts on a particle with
 * E = strengths_E[i] frac (x, y, 0), B = strengths_B[i] frac (y, -x, 0), frac = the share of the particle's
 * slice [z, z + vz dt) of this step that lies inside the lens.  With gamma_boost > 1 the slice is taken to the
 * lab frame first (z -> gamma z + uz_boost t) and the fields are transformed back (:176-187).  The four arrays
 * (host memory, n_lenses entries each) are copied; n_lenses = 0 switches the lens off.  `dt` is the step of the
 * level (warpx.getdt), not the dt of the push that applies the fields (PushP at initialisation uses -dt/2). */
typedef struct wxa_repeated_plasma_lens {
    int32_t n_lenses;
    double  period;
    const double* starts;
    const double* lengths;
    const double* strengths_E;
    const double* strengths_B;
    double  gamma_boost;     /* 1 = lab frame */
    double  dt;
} wxa_repeated_plasma_lens;
wxa_status wxa_workspace_set_repeated_plasma_lens(wxa_workspace* ws, const wxa_repeated_plasma_lens* lens);
/* warpx.gett_new(lev) as the external-field functor sees it (m_time, GetExternalFields.cpp:43): set before the
 * pushes of a step when a time-dependent external field (the boosted lens) is active */
wxa_status wxa_workspace_set_time(wxa_workspace* ws, double t);
/* Accumulator of the LDS-tile Esirkepov deposition for the container that owns this workspace.  WXA_ACC_FP64 (default):
 * ds_add_f64 tiles, the double/double build of the reference (1e-10 parity gate).  WXA_ACC_FP32: ds_add_f32 tiles --
 * every deposit is still evaluated in fp64 and rounded once when it enters the tile; the tile's sums then carry fp32
 * round-off, J stays an fp64 array.  Counterpart of the reference's WarpX_PRECISION=SINGLE switch (CMakeLists.txt:109-118)
 * restricted to where BASELINE.json's north_star asks for it; gate: the reference's single-precision tolerance 2e-6
 * (Examples/analysis_default_regression.py:18). */
#define WXA_ACC_FP64 0
#define WXA_ACC_FP32 1
wxa_status wxa_workspace_set_deposit_accumulator(wxa_workspace* ws, int32_t accumulator);
/* The container's plasma STREAMS through the grid: in a boosted-frame run (warpx.gamma_boost > 1, Source/Utils/WarpXUtil.cpp:
 * 114-141) every particle of a plasma at rest in the lab moves c dt / dz cells per step against the boost and most of them
 * cross a cell every step.  on != 0: the LDS-tile Esirkepov deposition (doEsirkepovDepositionShapeN,
 * CurrentDeposition.H:642-907) takes every particle through its wide-frame body inside the tile loop -- the body of a
 * particle that may cross a cell, 300 LDS atomics instead of 72 -- instead of deferring the crossing ones to a list
 * sized for the 1-2 % of a plasma at rest (what overflows that list is deposited with global atomics: 660 ms per launch
 * for the boosted wakefield deck at 256 x 256 x 512 x 8 per cell).  Same sums, another order of additions.  Default 0. */
wxa_status wxa_workspace_set_streaming_plasma(wxa_workspace* ws, int32_t on);

const char* wxa_version(void);
const char* wxa_last_error(void);

/* ------------------------------------------------------------------ */
/* Field solver (FDTD Yee)                                             */
/* ------------------------------------------------------------------ */

/* Replaces FiniteDifferenceSolver::EvolveB -> EvolveBCartesian<CartesianYeeAlgorithm>
 * (Source/FieldSolver/FiniteDifferenceSolver/EvolveB.cpp:51-117,122-215).
 * B += dt * curl-like upward differences of E over the valid boxes of Bx,By,Bz.
 * dinv[d] = 1/dx_d (CartesianYeeAlgorithm.H:29-43). */
wxa_status wxa_evolve_b(const wxa_field_view E[3], const wxa_field_view B[3],
                        double dt, const double dinv[3], void* stream);

/* Replaces FiniteDifferenceSolver::EvolveE -> EvolveECartesian<CartesianYeeAlgorithm>
 * (Source/FieldSolver/FiniteDifferenceSolver/EvolveE.cpp:53-115,120-250),
 * without the EB mask and the grad(F) term (both off on this path). */
wxa_status wxa_evolve_e(const wxa_field_view E[3], const wxa_field_view B[3],
                        const wxa_field_view J[3],
                        double dt, const double dinv[3], void* stream);

/* The first guard layer of B, updated with the same formula from the guard points of E and B already
 * present, next to the faces of the directions with grow[d] != 0: the points EvolveE reads beyond the
 * valid box are those of the components cell-centred along d at the indices lo - 1 and lo + ncell
 * (faces only).  Called after wxa_evolve_b when the guards of E and B were filled since their last
 * update, it stands in for the FillBoundaryB that the reference issues between EvolveB and EvolveE
 * (Source/Evolve/WarpXEvolve.cpp:421-426): a neighbour brick, or the periodic image, computes the same
 * numbers from the same operands, so nothing has to be exchanged.  Needs 2 guard points on E. */
wxa_status wxa_evolve_b_guard_layer(const wxa_field_view E[3], const wxa_field_view B[3], double dt,
                                    const double inv_dx[3], const int32_t grow[3], void* stream);

/* algo.maxwell_solver = ckc (first piece): the Cole-Karkkainen-Cowan solver.
 * wxa_ckc_stencil_coefficients replaces CartesianCKCAlgorithm::InitializeStencilCoefficients
 * (Source/FieldSolver/FiniteDifferenceSolver/FiniteDifferenceAlgorithms/CartesianCKCAlgorithm.H:28-102, 3-D branch):
 * coefs_d = {1/dx_d, alpha_d, beta_d(first transverse), beta_d(second transverse), gamma_d / dx_d} in the
 * reference's order (x: betaxy, betaxz; y: betayz, betayx; z: betazx, betazy).
 * wxa_evolve_b_ckc replaces FiniteDifferenceSolver::EvolveB -> EvolveBCartesian<CartesianCKCAlgorithm>
 * (EvolveB.cpp:102-105,122-215): the upward differences of E extended over the transverse neighbours
 * (UpwardDx/Dy/Dz, CartesianCKCAlgorithm.H:129-160,183-214,237-272); it reads one guard point of E in every
 * direction.  The update of E is the Yee one (the downward differences are the same, :164-181): wxa_evolve_e.
 * wxa_ckc_max_dt: CartesianCKCAlgorithm::ComputeMaxDt (:107-120) = min(dx) / c. */
void wxa_ckc_stencil_coefficients(const double cell_size[3], double coefs_x[5], double coefs_y[5], double coefs_z[5]);
double wxa_ckc_max_dt(const double cell_size[3]);
wxa_status wxa_evolve_b_ckc(const wxa_field_view E[3], const wxa_field_view B[3], double dt,
                            const double coefs_x[5], const double coefs_y[5], const double coefs_z[5], void* stream);

/* ------------------------------------------------------------------ */
/* Particles                                                           */
/* ------------------------------------------------------------------ */

/* Replaces PhysicalParticleContainer::PushPX
 * (Source/Particles/PhysicalParticleContainer.cpp:2549-2786): per particle
 * doGatherShapeN<order,galerkin> (Source/Particles/Gather/FieldGather.H:36-424)
 * -> doParticleMomentumPush (Source/Particles/Pusher/PushSelector.H:38-102;
 * Boris: UpdateMomentumBoris.H:15-53, Vay: UpdateMomentumVay.H:19-62)
 * -> UpdatePosition (Source/Particles/Pusher/UpdatePosition.H:24-45).
 * order in {1,2,3}; galerkin in {0,1}. */
wxa_status wxa_gather_push(const wxa_particle_view* p,
                           const wxa_field_view E[3], const wxa_field_view B[3],
                           const wxa_grid_geom* geom,
                           double q, double m, double dt,
                           int order, int galerkin, int pusher, void* stream);

/* Replaces PhysicalParticleContainer::PushP (:2368-2516): gather + momentum
 * push by a signed dt, positions untouched (first/last-step (de)synchronisation). */
wxa_status wxa_push_p(const wxa_particle_view* p,
                      const wxa_field_view E[3], const wxa_field_view B[3],
                      const wxa_grid_geom* geom,
                      double q, double m, double dt,
                      int order, int galerkin, int pusher, void* stream);

/* The same two operators with a workspace: when `ws` holds a valid cell sort of exactly these
 * particle arrays (wxa_sort_particles_by_cell), the LDS-tile variant runs (fields of each
 * 8x8x8-cell tile staged in LDS); otherwise identical to the two calls above.
 * move != 0 -> PushPX, move == 0 -> PushP. */
wxa_status wxa_gather_push_ws(const wxa_particle_view* p,
                              const wxa_field_view E[3], const wxa_field_view B[3],
                              const wxa_grid_geom* geom,
                              double q, double m, double dt,
                              int order, int galerkin, int pusher, int move,
                              wxa_workspace* ws, void* stream);

/* PushPX in two parts, so that the guard exchange of E and B can travel while most particles are pushed:
 * WXA_PART_INTERIOR = the particles of the tiles of the last cell sort that touch no face of the sorted
 * box (they read no guard point; nothing without a valid sort), WXA_PART_REST = all the others (the tiles
 * on the faces and the particles appended since the sort).  The two calls together do exactly what one
 * wxa_gather_push_ws(move = 1) does.  Needs a cell sort at least every few steps: a particle must stay
 * within a tile width (8 cells) minus the stencil of the tile it was sorted into. */
enum { WXA_PART_INTERIOR = 1, WXA_PART_REST = 2 };
wxa_status wxa_gather_push_part(const wxa_particle_view* p,
                                const wxa_field_view E[3], const wxa_field_view B[3],
                                const wxa_grid_geom* geom,
                                double q, double m, double dt,
                                int order, int galerkin, int pusher,
                                wxa_workspace* ws, int part, void* stream);

/* Replaces WarpXParticleContainer::DepositCurrent
 * (Source/Particles/WarpXParticleContainer.cpp:352-827) for
 * algo = Esirkepov: doEsirkepovDepositionShapeN<order>
 *        (Source/Particles/Deposition/CurrentDeposition.H:642-907),
 * algo = Direct:    doDepositionShapeN<order> (:48-335).
 * J is accumulated into (J must be zeroed by the caller once per step,
 * Source/Particles/MultiParticleContainer.cpp:470-472).
 * geom is built from the box grown by ng_depos_J.  ws may be NULL
 * (global-atomics variant); with a workspace holding a valid cell sort
 * (wxa_sort_particles_by_cell on the same positions) the LDS-tile variant runs. */
wxa_status wxa_deposit_current(const wxa_particle_view* p,
                               const wxa_field_view J[3],
                               const wxa_grid_geom* geom,
                               double q, double dt, double relative_time,
                               int order, int algo,
                               wxa_workspace* ws, void* stream);

/* Replaces doChargeDepositionShapeN<order>
 * (Source/Particles/Deposition/ChargeDeposition.H:37-180); diagnostics only. */
wxa_status wxa_deposit_charge(const wxa_particle_view* p,
                              const wxa_field_view* rho,
                              const wxa_grid_geom* geom,
                              double q, int order, void* stream);

/* Periodic wrap of particle positions into [plo, phi); the part of
 * amrex ParticleContainer::Redistribute the periodic path needs
 * (Source/Particles/MultiParticleContainer.cpp:651-654).
 * periodic[d] != 0 selects the directions wrapped on this brick. */
wxa_status wxa_enforce_periodic(const wxa_particle_view* p,
                                const double plo[3], const double phi[3],
                                const int periodic[3], void* stream);
/* The same wrap, restricted through the workspace's last cell sort to the tiles that touch a periodic face of the sorted
 * box (+ the particles appended since): valid while `steps_since_sort` pushes (< 1 cell each) cannot have carried a
 * particle of an interior tile out of the domain; otherwise, or without a usable sort, the plain pass. */
wxa_status wxa_enforce_periodic_sorted(const wxa_particle_view* p, const double plo[3], const double phi[3],
                                       const int periodic[3], wxa_workspace* ws, int32_t steps_since_sort,
                                       void* stream);

/* Replaces MultiParticleContainer::SortParticlesByBin (bin = 1 cell,
 * Source/Particles/MultiParticleContainer.cpp:615-621 -> amrex SortParticlesByBin):
 * counting sort by cell index inside the box [cell_lo, cell_lo+ncell).
 * `dst` receives the sorted copy (out of place; same np).  Also records the
 * per-cell offsets in ws for the tile-based deposition.  Retired particles
 * (idcpu == WXA_IDCPU_RETIRED, see wxa_pack_leavers) are moved behind the live ones. */
wxa_status wxa_sort_particles_by_cell(const wxa_particle_view* src,
                                      const wxa_particle_view* dst,
                                      const double plo[3], const double dinv[3],
                                      const int32_t cell_lo[3], const int32_t ncell[3],
                                      wxa_workspace* ws, void* stream);

/* The same sort folded into PushPX: SortParticlesByBin only permutes the tile (Source/Particles/
 * MultiParticleContainer.cpp:615-621), and PhysicalParticleContainer::PushPX (PhysicalParticleContainer.cpp:2549-2786) has
 * every particle in registers, so the sort needs no passes of its own over the eight arrays.  Between wxa_push_sort_begin
 * and wxa_push_sort_end every moving push on `ws` (wxa_gather_push_ws with move = 1, wxa_gather_push_part; PushP is
 * untouched) additionally
 *   WXA_PUSH_SORT_COUNT:   keys each particle's NEW position with the tile-major cell key of wxa_sort_particles_by_cell
 *       (wrap[d] != 0: a cell index one period outside along d is brought back -- the key of the position that
 *       wxa_enforce_periodic will produce; otherwise clamped) and takes its rank among equal keys; _end scans the
 *       histogram.  Retired particles get the bin behind the cells (check_retired != 0: the tile may hold some and the
 *       ids are looked at; 0 saves the 8 bytes per particle).  predict_dt != 0: the key is that of the position after
 *       predict_dt more seconds of free flight (x + u / gamma predict_dt) -- with the time step of the scattering push
 *       the sorted tile is in the cell order of the positions that push produces, up to what the fields do to a
 *       particle in one step, instead of the order of the positions it starts from.  The record stays in ws until a SCATTER uses it, a
 *       new COUNT replaces it or wxa_sort_particles_by_cell / wxa_partition_particles invalidates it;
 *   WXA_PUSH_SORT_SCATTER: writes the pushed particle (x y z ux uy uz, and w and idcpu carried over) to `dst` at the index
 *       the record of the last COUNT on the same arrays gives it, instead of in place: `dst` then holds the particles
 *       in the cell order of their positions BEFORE this push -- the order wxa_sort_particles_by_cell would have
 *       produced one push earlier; the LDS-tile kernels take it like any sort that is one step old.  Particles
 *       retired since the COUNT stay where they were counted (until the next cycle); particles appended since
 *       (p->np grew) keep their order behind the cell-sorted ones; the retired ones of the record end up behind those.
 *   Both at once (a sort every step): the new keys are recorded at the destination indices.
 * _end: *live = cell-sorted particles now at the front of the tile (dst after a SCATTER), *appended = particles that
 * follow them; the caller keeps live + appended particles of dst.  read_live != 0: the record contained retired particles
 * and *live is read from the device (synchronises); read_live = 0: *live = the record's particle count.  After a SCATTER
 * ws describes dst (tile offsets of the LDS-tile kernels), exactly as after wxa_sort_particles_by_cell(p, dst). */
enum { WXA_PUSH_SORT_COUNT = 1, WXA_PUSH_SORT_SCATTER = 2 };
wxa_status wxa_push_sort_begin(wxa_workspace* ws, int32_t mode, const wxa_particle_view* p,
                               const wxa_particle_view* dst, const double plo[3], const double dinv[3],
                               const int32_t cell_lo[3], const int32_t ncell[3], const int32_t wrap[3],
                               int32_t check_retired, double predict_dt, void* stream);
wxa_status wxa_push_sort_end(wxa_workspace* ws, int32_t read_live, int64_t* live, int64_t* appended,
                             void* stream);
/* 1 when ws holds a COUNT record that a SCATTER of exactly these arrays can use */
int32_t wxa_push_sort_pending(const wxa_workspace* ws, const wxa_particle_view* p);

/* 3-way partition of a tile along `dim` for the brick-to-brick part of
 * amrex ParticleContainer::Redistribute: dst = [stay | to-minus | to-plus] for positions
 * in [lo,hi) / < lo / >= hi.  counts[3] (host) is valid on return (synchronises). */
wxa_status wxa_partition_particles(const wxa_particle_view* src, const wxa_particle_view* dst,
                                   int dim, double lo, double hi, int64_t counts[3],
                                   wxa_workspace* ws, void* stream);

/* ---- brick-to-brick Redistribute without moving the tile ------------------------------
 * amrex ParticleContainer::Redistribute (called from MultiParticleContainer::Redistribute,
 * Source/Particles/MultiParticleContainer.cpp:651-654, once per step from
 * Source/Evolve/WarpXEvolve.cpp:514-537) re-buckets every particle.  Per step only ~1e-5 of a
 * brick's particles cross a face, so here the tile keeps its (cell-sorted) order between sorts:
 *   1. wxa_wrap_and_classify scans the positions once: periodic wrap, plus six index lists of
 *      the particles that left the brick, by the FIRST split direction d in which they are outside
 *      (list 2d = towards minus, 2d+1 = towards plus; decided before the wrap);
 *   2. wxa_pack_leavers copies the listed particles into a message (8 rows of n entries:
 *      x,y,z,w,ux,uy,uz,idcpu) and retires them in place: weight 0, momentum 0, position clamped
 *      into the brick, idcpu = WXA_IDCPU_RETIRED.  A retired particle deposits exact zeros and
 *      is dropped by the next wxa_sort_particles_by_cell;
 *   3. arrivals are appended behind the sorted part of the tile; the LDS-tile kernels cover the
 *      sorted part and the global-memory kernels the tail (wxa_gather_push_ws / wxa_deposit_current
 *      do this split themselves when ws holds a sort of the same array).
 * Needs p->idcpu != NULL (it carries the retired mark). */
#define WXA_IDCPU_RETIRED 0xFFFFFFFFFFFFFFFFull

/* Scans particles [first, first+count).  split[d] != 0: the brick has distinct neighbours along d
 * (leavers are listed); periodic[d] != 0: wrap positions into [prob_lo, prob_hi) along d.
 * lists: 6 device arrays of `capacity` int32 each (lists[c] + ... contiguous: list c starts at
 * lists + c*capacity); entries are indices relative to the view.  counts[6] (host) is valid on
 * return (synchronises); a count above `capacity` means the list was truncated: call again with
 * a larger capacity (positions are only wrapped once, the call is idempotent). */
wxa_status wxa_wrap_and_classify(const wxa_particle_view* p, int64_t first, int64_t count,
                                 const double prob_lo[3], const double prob_hi[3],
                                 const int periodic[3], const double brick_lo[3],
                                 const double brick_hi[3], const int split[3], int32_t* lists,
                                 int64_t capacity, int64_t counts[6], wxa_workspace* ws,
                                 void* stream);

/* The same scan with the leavers listed by their DESTINATION brick instead of by the first split direction: 27 lists,
 * list (ox + 1) + 3 (oy + 1) + 9 (oz + 1) for the brick at offset (ox, oy, oz) in {-1, 0, 1}^3 from this one (decided
 * on the unwrapped position, along split directions only; list 13 -- no offset -- stays empty).  A particle moves less
 * than a cell per step, so its destination is one of the 26 neighbours and it can be handed over in ONE message: one
 * count round per step for all neighbours instead of one per split direction, and no second classification of the
 * arrivals (edges and corners took up to three hops before).  What amrex ParticleContainer::Redistribute does with its
 * neighbour lists (MultiParticleContainer.cpp:651-654: Redistribute(..., local = 1)).  counts[27] (host) valid on return. */
wxa_status wxa_wrap_and_classify_dest(const wxa_particle_view* p, int64_t first, int64_t count,
                                      const double prob_lo[3], const double prob_hi[3],
                                      const int periodic[3], const double brick_lo[3],
                                      const double brick_hi[3], const int split[3], int32_t* lists,
                                      int64_t capacity, int64_t counts[27], wxa_workspace* ws,
                                      void* stream);

/* Writes the n listed particles into entries [offset, offset+n) of a message of 8 rows of
 * row_len doubles/uint64 each (row r at msg + 8*r*row_len bytes), so that several lists can
 * share one message.  retire != 0: see above. */
wxa_status wxa_pack_leavers(const wxa_particle_view* p, const int32_t* list, int64_t n, void* msg,
                            int64_t row_len, int64_t offset, int retire, const double brick_lo[3],
                            const double brick_hi[3], void* stream);

/* Number of particles of the last wxa_sort_particles_by_cell on ws that were NOT retired: they
 * occupy dst[0, n); the retired ones follow.  Synchronises. */
wxa_status wxa_sort_live_count(wxa_workspace* ws, int64_t* n, void* stream);

/* ------------------------------------------------------------------ */
/* Field boundary conditions (first "next" row of SURVEY.md 8(f))       */
/* ------------------------------------------------------------------ */

#define WXA_BOUNDARY_PERIODIC 0   /* boundary.field_lo/hi = periodic */
#define WXA_BOUNDARY_PEC      1   /* boundary.field_lo/hi = pec      */

/* Replace PEC::ApplyPECtoEfield / PEC::ApplyPECtoBfield
 * (Source/BoundaryConditions/WarpX_PEC.cpp:457-538 / :540-626; point rules SetEfieldOnPEC :117-196,
 * SetBfieldOnPEC :256-331), called at the end of WarpX::EvolveE / EvolveB
 * (Source/FieldSolver/WarpXPushFieldsEM.cpp:990 / :926).  On every point of the valid box of each
 * component grown by ng (= ng_FieldGather): E components tangential to a PEC face are zeroed on the
 * face and odd-mirrored into the guard cells behind it, normal components even-mirrored; for B the
 * normal component is zeroed / odd-mirrored and the tangential ones even-mirrored.
 * dom_lo/dom_hi: cell-centred index box of the whole domain (both inclusive, amrex Box::smallEnd /
 * bigEnd); pec_lo[d] / pec_hi[d] != 0 marks the PEC faces.  The views are this brick's arrays. */
wxa_status wxa_apply_pec_e(const wxa_field_view E[3], const int32_t dom_lo[3],
                           const int32_t dom_hi[3], const int32_t pec_lo[3],
                           const int32_t pec_hi[3], const int32_t ng[3], void* stream);
wxa_status wxa_apply_pec_b(const wxa_field_view B[3], const int32_t dom_lo[3],
                           const int32_t dom_hi[3], const int32_t pec_lo[3],
                           const int32_t pec_hi[3], const int32_t ng[3], void* stream);

#define WXA_PBOUNDARY_DEFAULT    0   /* periodic where the field boundary is periodic, absorbing where it is PEC */
#define WXA_PBOUNDARY_ABSORBING  1   /* boundary.particle_lo/hi = absorbing  */
#define WXA_PBOUNDARY_REFLECTING 2   /* boundary.particle_lo/hi = reflecting */
#define WXA_PBOUNDARY_PERIODIC   3   /* boundary.particle_lo/hi = periodic   */

/* Replaces WarpXParticleContainer::ApplyBoundaryConditions (Source/Particles/WarpXParticleContainer.cpp:
 * 1574-1660) with ApplyParticleBoundaries::apply_boundaries (Source/Particles/ParticleBoundaries_K.H:
 * 20-175), absorbing (reflection probability 0) and reflecting walls; periodic directions are left to
 * wxa_enforce_periodic / wxa_wrap_and_classify.  A particle beyond a reflecting wall (x < lo or x > hi)
 * is mirrored, x = 2 wall - x, and the momentum component normal to the wall changes sign; beyond an
 * absorbing wall it is lost: retired in place as wxa_pack_leavers does (idcpu = WXA_IDCPU_RETIRED,
 * weight and momentum 0, position clamped into the domain), to be dropped by the next sort.
 * bc_lo/bc_hi[d]: WXA_PBOUNDARY_ABSORBING / _REFLECTING act, anything else is ignored.
 * *n_lost (host) is valid on return (synchronises) unless NULL.  Needs p->idcpu. */
wxa_status wxa_apply_particle_boundaries(const wxa_particle_view* p, const double prob_lo[3],
                                         const double prob_hi[3], const int32_t bc_lo[3],
                                         const int32_t bc_hi[3], int64_t* n_lost,
                                         wxa_workspace* ws, void* stream);

/* Replaces PEC::ApplyReflectiveBoundarytoJfield (Source/BoundaryConditions/WarpX_PEC.cpp:713-900, point
 * rule SetRhoOrJfieldFromPEC :354-420), called from WarpX::SyncCurrentAndRho after SyncCurrent
 * (Source/Evolve/WarpXEvolve.cpp:625-640), for PEC field boundaries with absorbing particle
 * boundaries: the current deposited in the guard cells behind a PEC wall is folded onto its mirror
 * cell inside with the sign of an image charge (-1 for the components tangential to the wall, +1 for
 * the normal one), a component living on the wall is zeroed there, and the guard cells then receive
 * the image of the updated interior values (odd / even).  dom_lo/dom_hi: cell-centred domain box. */
wxa_status wxa_apply_pec_j(const wxa_field_view J[3], const int32_t dom_lo[3],
                           const int32_t dom_hi[3], const int32_t pec_lo[3],
                           const int32_t pec_hi[3], void* stream);

/* Replaces PEC::ApplyReflectiveBoundarytoRhofield (WarpX_PEC.cpp:628-711): rho behaves like a component
 * tangential to every wall (:664-666).  Called by WarpXParticleContainer::DepositCharge right after the
 * deposition (Source/Particles/WarpXParticleContainer.cpp:1285-1290), before the filter and the sum.
 * The guard columns of the directions without a wall are folded as well (the reference folds each
 * box's valid points only, which makes its rho next to a wall depend on the box decomposition). */
wxa_status wxa_apply_pec_rho(const wxa_field_view* rho, const int32_t dom_lo[3],
                             const int32_t dom_hi[3], const int32_t pec_lo[3],
                             const int32_t pec_hi[3], void* stream);

/* ------------------------------------------------------------------ */
/* Current filter and guard-cell exchange                              */
/* ------------------------------------------------------------------ */

/* Replaces BilinearFilter::ApplyStencil -> Filter::DoFilter with a 1-pass
 * binomial stencil per direction (Source/Filter/BilinearFilter.cpp:26-94,
 * Source/Filter/Filter.cpp:92-133): dst = filtered src over the whole
 * allocated box, zero padding beyond it.  src and dst must not alias. */
wxa_status wxa_filter_bilinear(const wxa_field_view* src, const wxa_field_view* dst,
                               void* stream);

/* Filter::ApplyStencil -> Filter::DoFilter for any symmetric stencil (Source/Filter/Filter.cpp:92-133): s0, s1, s2 are
 * the half stencils per direction as the reference stores them (entry 0 pre-halved), n0, n1, n2 <= 8 their lengths.
 * dst = filtered src over the whole allocated box, zero padding beyond it; src and dst must not alias.  Used for the
 * NCI corrector (NCIGodfreyFilter, lengths 1, 1, 5): PhysicalParticleContainer::applyNCIFilter filters E and B into
 * temporaries before the gather (Source/Particles/PhysicalParticleContainer.cpp:1900-1911, 2097-2172). */
wxa_status wxa_filter_stencil(const wxa_field_view* src, const wxa_field_view* dst, const double* s0, int32_t n0,
                              const double* s1, int32_t n1, const double* s2, int32_t n2, void* stream);

/* NCIGodfreyFilter::ComputeStencils (Source/Filter/NCIGodfreyFilter.cpp:45-154): the five z coefficients of the
 * Godfrey filter for c dt / dz = cdtodz, interpolated from the fitted tables of Source/Utils/NCIGodfreyTables.H
 * (coeff_set 0: Ex, Ey, Bz; 1: Bx, By, Ez; nodal_gather != 0: the momentum-conserving tables), entry 0 pre-halved as
 * Filter::DoFilter expects; the x and y stencils are {0.5}.  Host function. */
#define WXA_NCI_EX_EY_BZ 0
#define WXA_NCI_BX_BY_EZ 1
wxa_status wxa_nci_godfrey_stencil(double cdtodz, int32_t nodal_gather, int32_t coeff_set, double stencil_z[5]);

/* Single-brick periodic FillBoundary: guard points within ng of the valid box
 * take the value of their periodic image (amrex FabArray::FillBoundary with
 * geom.periodicity(); Source/ablastr/utils/Communication.cpp:71-115 called
 * from WarpX::FillBoundaryE/B, Source/Parallelization/WarpXComm.cpp:699-827).
 * periodic[d] = 0 leaves direction d untouched (it is handled by the
 * multi-brick exchange instead). */
wxa_status wxa_fill_boundary_periodic(const wxa_field_view* f, const int ng[3],
                                      const int periodic[3], void* stream);

/* The same fill for nf <= 6 fields at once (the components of E and B before the gather: WarpX::FillBoundaryE + FillBoundaryB,
 * Source/Evolve/WarpXEvolve.cpp:515-516): one launch per direction instead of one per field and direction.  f[0 .. nf). */
wxa_status wxa_fill_boundary_periodic_multi(const wxa_field_view* f, int32_t nf, const int ng[3],
                                            const int periodic[3], void* stream);

/* The extra step of FillBoundaryAndSync (Source/ablastr/utils/Communication.cpp:99-101,
 * 109-110; WarpX::sync_nodal_points, Source/WarpX.H:1523) on a self-periodic direction:
 * the high-edge nodal point takes the value of the low-edge one (its owner). */
wxa_status wxa_sync_nodal_periodic(const wxa_field_view* f, const int periodic[3], void* stream);

/* Single-brick periodic SumBoundary(src_ng): every point ends up holding the
 * sum over all periodic images taken from valid + src_ng guard regions; all
 * guards are refreshed (Source/Parallelization/WarpXSumGuardCells.cpp:17-37). */
wxa_status wxa_sum_boundary_periodic(const wxa_field_view* f, const int src_ng[3],
                                     const int periodic[3], void* stream);

/* ... for nf <= 6 fields at once (the three components of J: WarpX::SyncCurrent, Source/Parallelization/WarpXComm.cpp:1386-1424). */
wxa_status wxa_sum_boundary_periodic_multi(const wxa_field_view* f, int32_t nf, const int src_ng[3],
                                           const int periodic[3], void* stream);

/* Multi-brick exchange helpers: copy a sub-box of a field to/from a dense
 * staging buffer (i fastest).  box = [blo, bhi) in the field's index space.
 * unpack mode: 0 = overwrite (FillBoundary), 1 = add (SumBoundary). */
wxa_status wxa_pack_box(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3],
                        double* buf, void* stream);
wxa_status wxa_unpack_box(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3],
                          const double* buf, int mode, void* stream);
/* ... with comm_float_type = float on the wire: warpx.do_single_precision_comms
 * (Source/ablastr/utils/Communication.cpp:37-56 ParallelCopy, :90-106 FillBoundary, :159-170 SumBoundary;
 * Communication.H:30 comm_float_type).  The slab is rounded to float when packed and widened when unpacked
 * (mode 1 adds the widened value in double); the arrays themselves stay double.  The reference rounds the WHOLE
 * array through a float copy at every exchange (mixedCopy of valid points and guards, both ways); here only what
 * travels is rounded -- the points a brick owns keep their double values, a strictly smaller perturbation. */
wxa_status wxa_pack_box_f32(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3],
                            float* buf, void* stream);
wxa_status wxa_unpack_box_f32(const wxa_field_view* f, const int32_t blo[3], const int32_t bhi[3],
                              const float* buf, int mode, void* stream);

/* Zero a field including guards (MultiFab::setVal(0), MultiParticleContainer.cpp:470-472). */
wxa_status wxa_field_set_zero(const wxa_field_view* f, void* stream);
/* ... nf <= 6 fields in one launch (the three components of J, MultiParticleContainer.cpp:470-472) */
wxa_status wxa_field_set_zero_multi(const wxa_field_view* f, int32_t nf, void* stream);

/* Blocking device<->host copies for hosts without their own HIP binding. */
wxa_status wxa_copy_to_host(void* dst_host, const void* src_dev, int64_t bytes);
wxa_status wxa_copy_to_device(void* dst_dev, const void* src_host, int64_t bytes);
wxa_status wxa_device_synchronize(void);

/* ------------------------------------------------------------------ */
/* Step-level API: the host layer (warpx_amd/csrc/host/) re-states      */
/* WarpX::Evolve / OneStep_nosub on the explicit FDTD, single-level,    */
/* periodic branch (Source/Evolve/WarpXEvolve.cpp:94-347,354-455) on    */
/* top of the kernels above.  One wxa_sim = one brick on one GPU.       */
/* ------------------------------------------------------------------ */

enum { WXA_GRID_STAGGERED = 0, WXA_GRID_COLLOCATED = 1 };
enum { WXA_SOLVER_YEE = 0, WXA_SOLVER_CKC = 1 };   /* algo.maxwell_solver */

typedef struct wxa_sim_config {
    int32_t n_cell[3];           /* amr.n_cell, whole domain                    */
    double  prob_lo[3];          /* geometry.prob_lo                            */
    double  prob_hi[3];          /* geometry.prob_hi                            */
    double  cfl;                 /* warpx.cfl (CartesianYeeAlgorithm.H:48-56)   */
    int32_t nox;                 /* algo.particle_shape, 1..4 (all on the LDS tiles; 4 since round 6) */
    int32_t galerkin;            /* WarpX::galerkin_interpolation (WarpX.cpp:154) */
    int32_t particle_pusher;     /* WXA_PUSHER_*                                */
    int32_t current_deposition;  /* WXA_DEPOSIT_*                               */
    int32_t use_filter;          /* warpx.use_filter: 1-pass bilinear on J      */
    int32_t sort_interval;       /* warpx.sort_intervals; <=0 = never           */
    int32_t nbricks[3];          /* domain decomposition, one brick per GPU     */
    int32_t coord[3];            /* this brick's coordinates                    */
    int32_t field_boundary_lo[3];/* boundary.field_lo: WXA_BOUNDARY_* (0 = periodic, the default) */
    int32_t field_boundary_hi[3];/* boundary.field_hi; PEC only along unsplit directions (J next to a wall
                                    is folded back with the image-charge sign of an absorbing wall) */
    int32_t particle_boundary_lo[3]; /* boundary.particle_lo: WXA_PBOUNDARY_* (0 = default)              */
    int32_t particle_boundary_hi[3]; /* boundary.particle_hi                                             */
    int32_t overlap_halo;        /* 1 (bricks, all-periodic runs): the guard exchanges travel on a second stream --
                                    E and B behind the push of the interior tiles, J behind B's half update  */
    int32_t grid_type;           /* warpx.grid_type: WXA_GRID_STAGGERED (0, the default).  WXA_GRID_COLLOCATED exists
                                    in the CPU restatement only (it pins the direct deposition to the reference's
                                    test_3d_langmuir_multi_nodal checksums); the library refuses it              */
    int32_t maxwell_solver;      /* algo.maxwell_solver: WXA_SOLVER_YEE (0, the default) or WXA_SOLVER_CKC; with CKC
                                    dt = cfl min(dx)/c and every field exchange of the reference's schedule is issued
                                    (the B update reads guard points of E)                                        */
    double  gamma_boost;         /* warpx.gamma_boost with warpx.boost_direction = z (WarpXUtil.cpp:114-141); 0 or 1 = lab
                                    frame.  prob_lo / prob_hi above are boosted-frame values already (the inputs reader
                                    applies ConvertLabParamsToBoost, WarpXUtil.cpp:180-262); the value reaches the
                                    injection (MapParticletoBoostedFrame, the boosted branch of AddPlasma), the
                                    injection position of the moving window and the repeated plasma lens          */
    int32_t use_fdtd_nci_corr;   /* particles.use_fdtd_nci_corr (WarpX::use_fdtd_nci_corr, MultiParticleContainer.cpp:327): the
                                    NCI corrector of the boosted-frame runs -- E and B are filtered along z with the Godfrey
                                    stencil before every species' gather (applyNCIFilter), E/B carry 4 more guard cells in z */
} wxa_sim_config;

/* ---- second "next" row: moving window, continuous plasma injection, laser antenna -----------------
 * (what Examples/Physics_applications/laser_acceleration needs on top of the PEC walls; CPU
 * restatement pinned to the reference's test_3d_laser_acceleration golden checksums) */

/* warpx.do_moving_window / moving_window_dir / moving_window_v (Source/Utils/WarpXMovingWindow.cpp:138-476):
 * after every step the window position advances by v c dt; when it has crossed whole cells the
 * fields are shifted by that many cells (zeros enter), prob_lo/hi move, and species with continuous
 * injection receive new plasma in the cells that entered.  The window direction must be unsplit. */
typedef struct wxa_moving_window {
    int32_t dir;     /* 0,1,2 */
    double  v;       /* in units of c (> 0: towards +dir) */
} wxa_moving_window;

/* <species>.injection_style = NUniformPerCell with a constant density, at rest, inside
 * [lo, hi) (xmin..zmax), <species>.do_continuous_injection = 1
 * (PhysicalParticleContainer::AddPlasma, Source/Particles/PhysicalParticleContainer.cpp:924-1333;
 * ContinuousInjection :2518-2528) */
typedef struct wxa_plasma_injector {
    double  density;        /* <species>.density, m^-3              */
    int32_t ppc[3];         /* num_particles_per_cell_each_dim      */
    double  lo[3], hi[3];   /* xmin,ymin,zmin / xmax,ymax,zmax (+-1e300 = unbounded), lab frame */
    double  gamma_boost;    /* warpx.gamma_boost (boost along z); 0 or 1 = lab frame.  Boosted frame (the branch at
                               PhysicalParticleContainer.cpp:1210-1247): bounds and density are the lab-frame ones, looked
                               up at z0_lab = gamma (z (1 - beta beta_bulk) - c t (beta_bulk - beta)); the density becomes
                               gamma n (1 - beta beta_z) and u_z -> gamma (u_z - beta gamma_lab) per particle          */
    double  t;              /* warpx.gett_new(lev) when the plasma is added: the ballistic correction
                               (applyBallisticCorrection, :138-148) also moves the bounds of a lab-frame plasma with a
                               bulk velocity along z                                                               */
} wxa_plasma_injector;

/* PhysicalParticleContainer::AddPlasma on the device for that injector: the particles of the `ncells` cells
 * whose low corner is `corner` (InjectorPositionRegular lattice, weight = density dV / ppc, momentum
 * (u_mean + u_th N(0,1)) c per component -- NULL = at rest; the normal draws are a counter-based stream
 * (Philox4x32-10 keyed by `seed`, counter = the particle's integer coordinates on the global lattice of
 * injection points, Box-Muller), so a particle gets the same momentum whatever the brick layout; the reference draws from AMReX's
 * generator, which no other program reproduces) that lie inside the injector's bounds and strictly inside the
 * brick [brick_lo, brick_hi], written into the free slots `dst` (dst->np = room available; idcpu = 0).
 * *n_added (host) is valid on return (synchronises); the order of the new particles is not specified
 * (the cell sort that follows an injection fixes it).  WXA_ERR_NOMEM if dst is too small. */
typedef struct wxa_injected_momentum {
    double   u_mean[3];   /* ux_m, uy_m, uz_m (gamma beta; also the value of momentum_distribution_type = constant) */
    double   u_th[3];     /* ux_th, uy_th, uz_th of momentum_distribution_type = gaussian; 0 = no spread        */
    uint64_t seed;        /* stream of the gaussian draws (warpx.random_seed mixed with the species index)     */
    double   origin[3];   /* a corner of the cell lattice that never changes (geometry.prob_lo at t = 0): the
                             draws are numbered by the particle's lattice coordinates relative to it          */
} wxa_injected_momentum;
wxa_status wxa_add_plasma(const wxa_particle_view* dst, const wxa_plasma_injector* inj,
                          const double corner[3], const int32_t ncells[3], const double dx[3],
                          const double brick_lo[3], const double brick_hi[3],
                          const wxa_injected_momentum* momentum,
                          int64_t* n_added, wxa_workspace* ws, void* stream);

/* lasers.names / <laser>.profile = Gaussian (Source/Particles/LaserParticleContainer.cpp,
 * Source/Laser/LaserProfilesImpl/LaserProfileGaussian.cpp), lab frame, no space-time couplings */
typedef struct wxa_laser_antenna {
    double position[3], direction[3], polarization[3];
    double e_max, wavelength;
    double waist, duration, t_peak, focal_distance;   /* profile_* */
} wxa_laser_antenna;

/* Replaces WarpX::shiftMF (Source/Utils/WarpXMovingWindow.cpp:478-648) for a zero external field and a
 * forward window: f(i) <- f(i + num_shift along dir) over the whole array (guards included) except its
 * top num_shift layers, after (1) refreshing one guard cell of the periodic directions and (2) zeroing
 * everything beyond the domain on the high side, both on a scratch copy `tmp` (a second array of the same
 * shape, contents irrelevant).  num_shift <= the guard depth along dir. */
#define WXA_WINDOW_KEEP_GUARDS 2   /* periodic[dir] of wxa_shift_field_window: the guards beyond the high face of the
                                      window direction hold the next brick's cells (filled by the caller): they enter
                                      this brick instead of the zero external field (bricks along the window)        */
wxa_status wxa_shift_field_window(const wxa_field_view* f, double* tmp, int32_t dir, int32_t num_shift,
                                  const int periodic[3], void* stream);

/* The laser antenna push: LaserParticleContainer::calculate_laser_plane_coordinates,
 * GaussianLaserProfile::fill_amplitude (no space-time couplings) and update_laser_particle
 * (Source/Particles/LaserParticleContainer.cpp:795-951, Source/Laser/LaserProfilesImpl/
 * LaserProfileGaussian.cpp:104-161): the antenna particles get the velocity
 * -/+ mobility * E(X, Y, t) c along the polarization (sign opposite to the sign of their weight), their
 * momentum gamma v, and move by v dt.  p_X, p_Y: unit polarization vectors (p_Y = n x p_X). */
typedef struct wxa_laser_push_params {
    double position[3], p_X[3], p_Y[3];
    double mobility;                                  /* ComputeWeightMobility: 0.05 / e_max (/ gamma_boost) */
    double e_max, wavelength, waist, duration, t_peak, focal_distance;
    /* boosted frame (update_laser_particle, LaserParticleContainer.cpp:892-916): the antenna drifts with
     * -beta_boost c along nvec on top of the emitting velocity, and gamma = gamma_boost / sqrt(1 - (v/c)^2);
     * `t` of wxa_laser_push is then the lab-frame time t / gamma_boost + beta_boost Z0_lab / c (:574-579).
     * gamma_boost = 0 or 1: lab frame, nvec unused. */
    double nvec[3];
    double gamma_boost;
} wxa_laser_push_params;
wxa_status wxa_laser_push(const wxa_particle_view* p, const wxa_laser_push_params* par, double t,
                          double dt, void* stream);

/* Neighbour exchange supplied by the host program (torch.distributed over
 * RCCL in bench.py; absent = single brick, all directions self-periodic).
 * Replaces the MPI layer under amrex FabArray::FillBoundary/SumBoundary and
 * ParticleContainer::Redistribute.  `exchange` posts nmsg
 * sends and nmsg receives of device buffers and returns when they are
 * enqueued on `stream` (stream-ordered completion; `stream` is the library's
 * main stream, or its exchange stream when overlap_halo is set). */
typedef struct wxa_comm {
    void* ctx;
    int32_t rank, nranks;
    int (*exchange)(void* ctx, int nmsg,
                    const int32_t* send_peer, void* const* send_buf, const int64_t* send_bytes,
                    const int32_t* recv_peer, void* const* recv_buf, const int64_t* recv_bytes,
                    void* stream);
    /* small host-side all-to-neighbours count exchange (blocking) */
    int (*exchange_counts)(void* ctx, int nmsg,
                           const int32_t* send_peer, const int64_t* send_val,
                           const int32_t* recv_peer, int64_t* recv_val);
} wxa_comm;

/* ---- the library's own transport: RCCL over xGMI ------------------------------------------------------
 * Fills a wxa_comm whose callbacks enqueue ncclSend / ncclRecv groups on the stream they are handed (no host
 * wait per exchange; the 8-byte particle counts travel on a stream of the transport's own).  What the reference
 * gets from MPI under amrex FillBoundary / SumBoundary / Redistribute (Source/ablastr/utils/Communication.cpp:
 * 71-175, Source/Parallelization/WarpXComm.cpp:699-827,1386-1424).  One process per GPU: rank 0 calls
 * wxa_rccl_unique_id and hands the 128 bytes to the others (any channel: torch.distributed, MPI, a file), then
 * every rank calls wxa_rccl_comm_create after selecting its device.  RCCL is loaded with dlopen at the first
 * call: WXA_ERR_UNSUPPORTED where librccl.so is missing. */
#define WXA_RCCL_ID_BYTES 128
#define WXA_RCCL_LOOPBACK 1   /* messages to this rank itself go through ncclSend / ncclRecv as well (tests) */
#define WXA_RCCL_TIMING   2   /* a pair of HIP events around every exchange (wxa_rccl_comm_stats.timed_ms)  */
typedef struct wxa_rccl_stats {
    int64_t n_exchanges, n_messages, bytes_sent, n_count_exchanges, timed_exchanges;
    double timed_ms;
} wxa_rccl_stats;
wxa_status wxa_rccl_unique_id(char id[WXA_RCCL_ID_BYTES]);
/* Event pairs around every exchange on / off at run time (diagnostic passes: the events are pooled, reading them
 * waits on the host, so the headline timing of bench.py runs with timing off). */
wxa_status wxa_rccl_comm_set_timing(wxa_comm* comm, int32_t on);
wxa_status wxa_rccl_comm_create(const char id[WXA_RCCL_ID_BYTES], int32_t rank, int32_t nranks, int32_t flags,
                                wxa_comm* out);
void       wxa_rccl_comm_destroy(wxa_comm* comm);
wxa_status wxa_rccl_comm_stats(wxa_comm* comm, wxa_rccl_stats* out, int32_t reset);

typedef struct wxa_sim wxa_sim;

wxa_status wxa_sim_create(const wxa_sim_config* cfg, const wxa_comm* comm, wxa_sim** out);
void       wxa_sim_destroy(wxa_sim* s);
/* Adds a species (WarpXParticleContainer: charge, mass + SoA tile).
 * Arrays are device pointers, copied; particles must lie inside this brick. */
wxa_status wxa_sim_add_species(wxa_sim* s, double charge, double mass,
                               const wxa_particle_view* init, int32_t* species_id);
/* WarpX::Evolve(numsteps): first step de-synchronises u by PushP(-dt/2),
 * the last one re-synchronises (WarpXEvolve.cpp:142-145,222-226). */
wxa_status wxa_sim_evolve(wxa_sim* s, int32_t numsteps);
/* on = 0: wxa_sim_evolve leaves the momenta at the half step when it returns and the next call continues from there,
 * so that several calls are exactly the steps of one long WarpX::Evolve (what a run of many steps looks like between
 * its first and last step); wxa_sim_synchronize does the push of WarpX::Synchronize (WarpXEvolve.cpp:65-93) when the
 * momenta are wanted at the time of the positions (diagnostics, checksums).  Default: on = 1, the reference's behaviour. */
wxa_status wxa_sim_set_synchronize_at_end(wxa_sim* s, int32_t on);
wxa_status wxa_sim_synchronize(wxa_sim* s);
/* warpx.safe_guard_cells (Source/WarpX.cpp:625, Source/Parallelization/GuardCellManager.cpp:297-308,
 * WarpXComm.cpp:759,824, WarpXEvolve.cpp:449-451): every FillBoundary exchanges all allocated guard cells and every
 * exchange of the reference's schedule is issued; the valid points are unchanged.  Before the first step; refused
 * together with overlap_halo. */
wxa_status wxa_sim_set_safe_guard_cells(wxa_sim* s, int32_t on);
/* warpx.do_single_precision_comms (Source/WarpX.cpp:614, ablastr/utils/Communication.cpp:37-56,90-106,159-170): float
 * on the wire of every guard exchange between bricks (wxa_pack_box_f32 / wxa_unpack_box_f32), half the xGMI bytes.
 * Every brick of a run must ask for the same (checked before the first step). */
wxa_status wxa_sim_set_single_precision_comms(wxa_sim* s, int32_t on);
/* particles.E_external_particle / particles.B_external_particle (constant external fields on the particles of
 * species `id`; the reference keeps them per container and reads them from the `particles.` block) */
wxa_status wxa_sim_set_external_particle_fields(wxa_sim* s, int32_t id, const double E[3], const double B[3]);
/* WXA_ACC_FP64 (default) / WXA_ACC_FP32 tiles for the Esirkepov deposition of species `id`
 * (wxa_workspace_set_deposit_accumulator) */
wxa_status wxa_sim_set_deposit_accumulator(wxa_sim* s, int32_t id, int32_t accumulator);
/* <species>.do_classical_radiation_reaction (PhysicalParticleContainer.cpp:330-340; PushSelector.H:60-87: the species is
 * pushed by UpdateMomentumBorisWithRadiationReaction whatever algo.particle_pusher says) */
wxa_status wxa_sim_set_radiation_reaction(wxa_sim* s, int32_t id, int32_t on);
/* ---- input-deck front end -------------------------------------------------
 * Builds the simulation a WarpX inputs file describes (amrex::ParmParse syntax, FILE includes,
 * my_constants, math expressions; WarpX::ReadParameters' defaults) for the parameters on this path:
 * 3-D Cartesian, Yee, periodic / PEC field and periodic / absorbing / reflecting particle boundaries,
 * Esirkepov / direct deposition, Boris / Vay, bilinear filter, moving window, species injected as
 * NUniformPerCell (constant density; at rest, constant or parsed momentum; continuous injection),
 * SingleParticle, MultipleParticles, Gaussian laser antennas, E/B initialised by constants or parsed
 * functions.  Any other parameter that is not plain output / AMReX box sizing is an error, by name.
 * overrides: "name=value" strings applied after the file, like the reference's command line.
 * nbricks / coord: this library's decomposition; nbricks = NULL lets the library split the domain into
 * comm->nranks bricks (one brick without a comm) along its periodic, window-free directions, with
 * rank = cx + nbx (cy + nby cz).  The deck's amr.max_grid_size and blocking_factor describe AMReX
 * boxes and are ignored. */
wxa_status wxa_sim_create_from_inputs(const char* inputs_path, int32_t n_overrides,
                                      const char* const* overrides, const wxa_comm* comm,
                                      const int32_t* nbricks, const int32_t* coord, wxa_sim** out);
int32_t     wxa_sim_max_step(const wxa_sim* s);                 /* the deck's max_step, -1 if unset */
int32_t     wxa_sim_num_species(const wxa_sim* s);
const char* wxa_sim_species_name(const wxa_sim* s, int32_t id); /* particles.species_names[id]      */
/* The reference's regression checksum (Regression/Checksum/checksum.py) of the current state as JSON
 * text: "lev=0" sums of |cell-centred field| for E, B, j, rho and per species sums of |x|, |m u|, w
 * (this brick's share).  Returns the text length (buf may be NULL to ask for it) or < 0. */
int64_t     wxa_sim_checksum_json(wxa_sim* s, char* buf, int64_t capacity);
/* FlushFormatPlotfile::WriteToFile (Source/Diagnostics/FlushFormats/FlushFormatPlotfile.cpp:61-113): an AMReX plotfile
 * directory `dir` -- Header, Level_0/Cell_H + Cell_D_<brick> with Ex..Bz, jx..jz, rho averaged to the cell centres,
 * <species>/Header + Level_0/Particle_H + DATA_<brick> with x y z weight momentum_x/y/z (SI) of the live particles,
 * WarpXHeader, warpx_job_info -- in the text / binary layouts of AMReX as the reference restates them in
 * Source/Diagnostics/BTD_Plotfile_Header_Impl.cpp; what Regression/Checksum/checksum.py (yt) and
 * Tools/PostProcessing/read_raw_data.py read.  Collective over the bricks of a run, like a parallel
 * amrex::WriteMultiLevelPlotfile: every brick calls it with the same `dir` (one file system) and writes its own grid,
 * brick 0 writes the headers that list all of them; nobody returns before the plotfile is complete. */
wxa_status  wxa_sim_write_plotfile(wxa_sim* s, const char* dir);
/* <diag>.diag_type = Full with format = plotfile (Source/Diagnostics/FullDiagnostics.cpp:109-125, 295-303;
 * Diagnostics.cpp:47-60, 611-625; MultiDiagnostics.cpp:83-115): from now on the step loop writes the plotfile
 * <file_prefix><istep, file_min_digits> (wxa_sim_write_plotfile; the reference's "diags/diag1000040") after every step
 * whose number is in `intervals` (the reference's slice syntax), before the first step if 0 is, and -- dump_last_timestep --
 * once more when a deck-built run reaches its max_step (a run built through wxa_sim_create ends when its caller says so:
 * wxa_sim_flush_diags_last_timestep).  file_prefix NULL or "" = "diags/<name>", file_min_digits <= 0 = 6, fields =
 * space-separated names out of Ex Ey Ez Bx By Bz jx jy jz rho (NULL = the reference's default, Ex .. jz; names outside
 * the list are left out with a warning), write_species = 0: fields only.  In a deck: diagnostics.diags_names. */
wxa_status  wxa_sim_add_full_diag(wxa_sim* s, const char* name, const char* intervals, const char* file_prefix,
                                  int32_t file_min_digits, const char* fields, int32_t write_species,
                                  int32_t dump_last_timestep);
wxa_status  wxa_sim_flush_diags_last_timestep(wxa_sim* s);
/* Lab-frame snapshot i of wxa_sim_add_btd as a plotfile (fields and the back-transformed particles of every species;
 * geometry and time of the lab frame): what the reference's BTD flushes hold once merged
 * (BTDiagnostics::MergeBuffersForPlotfile, BTDiagnostics.cpp:1146-1314), as one grid -- this brick's share. */
wxa_status  wxa_sim_btd_write_plotfile(wxa_sim* s, int32_t i, const char* dir);
/* BTDiagnostics::Flush + MergeBuffersForPlotfile (BTDiagnostics.cpp:1027-1314): snapshot i is written to
 * <file_prefix><i, file_min_digits digits>/ while it is assembled -- every full buffer (buffer_size slices) becomes one more
 * grid of the snapshot's plotfile (Level_0/Cell_D_<n>, <species>/Level_0/DATA_<n>, headers rewritten for the grids so far)
 * and only the buffer being filled is kept in memory (wxa_sim_btd_data / _particles then have nothing to return).  After
 * wxa_sim_add_btd, before the first step; `<diag>.file_prefix` in a deck.  On several bricks every brick is given the SAME
 * prefix: brick 0 collects the bricks' shares of a buffer and writes the one plotfile of the snapshot, with the grids the run
 * on one brick writes (the flushes are collective; the bricks meet them in the same step).
 * wxa_sim_btd_flush: the forced flush after the last step (partly filled buffers go to disk as they are); collective. */
wxa_status  wxa_sim_btd_set_flush(wxa_sim* s, const char* file_prefix, int32_t file_min_digits);
wxa_status  wxa_sim_btd_flush(wxa_sim* s);
/* Index box (inclusive) of this brick's share of snapshot i in the snapshot's (x, y, k_lab) index space: m_snapshot_box
 * (BTDiagnostics.cpp:489-506) cut to the brick's cells in x and y.  A brick keeps all of z and fills the slices whose
 * plane lies in its cells; the shares of the bricks of a run add up to the snapshot. */
wxa_status  wxa_sim_btd_box(wxa_sim* s, int32_t i, int32_t lo[3], int32_t hi[3]);
/* The decks' expression evaluator (what the reference gets from amrex::Parser): value of `expr` with
 * the named variables bound to `values`; q_e, m_e, m_p, m_u, epsilon0, mu0, clight, kb, pi predefined. */
wxa_status  wxa_parser_eval(const char* expr, int32_t nvars, const char* const* names,
                            const double* values, double* out);

/* 1 if wxa_sim_config::overlap_halo took effect (needs a split direction, all-periodic boundaries and
 * the library's second stream). */
int32_t wxa_sim_halo_overlap(const wxa_sim* s);

/* RhoFunctor::operator() (Source/Diagnostics/ComputeDiagFunctors/RhoFunctor.cpp:42-61): total charge
 * density of all species (and laser antennas) at the current positions, mirrored over PEC walls,
 * filtered and summed over guards / bricks; readable afterwards as field "rho" (nodal). */
wxa_status wxa_sim_compute_rho(wxa_sim* s);
double     wxa_sim_dt(const wxa_sim* s);
int64_t    wxa_sim_istep(const wxa_sim* s);
/* name in {"Ex","Ey","Ez","Bx","By","Bz","jx","jy","jz"} (MultiFabRegister::get,
 * Source/ablastr/fields/MultiFabRegister.H:389-470). */
wxa_status wxa_sim_get_field(wxa_sim* s, const char* name, wxa_field_view* out);
wxa_status wxa_sim_get_particles(wxa_sim* s, int32_t species_id, wxa_particle_view* out);
/* <diag>.diag_type = BackTransformed with do_back_transformed_fields = 1 (Source/Diagnostics/BTDiagnostics.cpp,
 * ComputeDiagFunctors/BackTransformFunctor.cpp), fields only, one brick, boost and moving window along z:
 * num_snapshots lab-frame snapshots at t_lab = i dt_snapshots_lab (+ the offset of :346-347), each assembled from one
 * z slice per step -- the cell-centred Ex Ey Ez Bx By Bz jx jy jz rho interpolated at the snapshot's current plane in the
 * boosted frame (amrex::get_slice_data), Lorentz-transformed (LorentzTransformZ, :246-317), stored at lab index k_lab
 * (k_index_zlab, :892-905).  buffer_size only rounds the snapshot's length as :464-469 do; the snapshot is kept whole in
 * host memory; write_species != 0 adds the particles (wxa_btd_select_particles).  wxa_sim_btd_info: cells n (x, y, z), lab-frame extent along z, t_lab, slices received, closed flag;
 * wxa_sim_btd_data: component comp (order above) into out[k][j][i] (host pointer).  Diagnostic stage: host arithmetic. */
/* The particle half: BackTransformParticleFunctor (ComputeDiagFunctors/BackTransformParticleFunctor.cpp:76-152) -- the
 * particles of p that crossed the snapshot's plane during the step (SelectParticles, .H:49-62: plane at z_boost now,
 * z_boost_old one step ago; old6 = x y z ux uy uz before the push, CopyParticleAttribs, PhysicalParticleContainer.cpp:
 * 2626-2629), interpolated in time to t_lab and Lorentz-transformed to the lab frame (LorentzTransformParticles, .H:106-168),
 * appended to out[7][capacity] (rows x y z w ux uy uz, device memory) in no particular order.  *n_selected is the number
 * that crossed; if it exceeds capacity the surplus was not stored.  Blocks on the stream (it reads the count). */
wxa_status wxa_btd_select_particles(const wxa_particle_view* p, const double* const old6[6], double z_boost,
                                    double z_boost_old, double t_boost, double dt, double t_lab, double gamma_boost,
                                    double* out, int64_t capacity, int64_t* n_selected, void* stream);
wxa_status wxa_sim_add_btd(wxa_sim* s, int32_t num_snapshots, double dt_snapshots_lab, int32_t buffer_size,
                           int32_t write_species);
/* particles of species id that snapshot i has met so far (write_species != 0): their number, and the rows
 * x y z w ux uy uz, lab frame, into out[7][n] (host pointer) */
wxa_status wxa_sim_btd_num_particles(wxa_sim* s, int32_t i, int32_t id, int64_t* n);
wxa_status wxa_sim_btd_particles(wxa_sim* s, int32_t i, int32_t id, double* out);
wxa_status wxa_sim_btd_info(wxa_sim* s, int32_t i, int32_t n[3], double z_lab[2], double* t_lab, int32_t* slices,
                            int32_t* full);
wxa_status wxa_sim_btd_data(wxa_sim* s, int32_t i, int32_t comp, double* out);

/* Per-phase accumulated device time in ms since the last reset, named after the
 * reference's profiler regions: 0 GatherAndPush,
 * 1 CurrentDeposition, 2 SyncCurrent(filter+SumBoundary), 3 EvolveB, 4 EvolveE,
 * 5 FillBoundary, 6 Redistribute+Sort.  counts[i] = number of launches. */
wxa_status wxa_sim_get_timers(wxa_sim* s, double ms[8], int64_t counts[8], int reset);
/* Only the step's neighbour exchanges, on the run's own arrays and with its real message sizes, nothing computed in
 * between: ms[0] FillBoundary of E and B at the gather's guard depth (Source/Evolve/WarpXEvolve.cpp:515-516 ->
 * Source/Parallelization/WarpXComm.cpp:699-827), ms[1] SumBoundary of J (WarpXComm.cpp:1386-1424), ms[2] Redistribute of
 * every species (count round + data exchange; Source/Particles/MultiParticleContainer.cpp:627-632), ms[3] the three in
 * a row -- milliseconds per call over `reps` calls, host clock around a stream sync.  Collective: every rank of the
 * run calls it at the same point.  E and B are unchanged; J's guard sums land in cells the next deposition zeroes. */
wxa_status wxa_sim_dry_comm(wxa_sim* s, int32_t reps, double ms[4]);
wxa_status wxa_sim_enable_timers(wxa_sim* s, int enable);

/* ---- Reduced diagnostics: the parity metric on the device (Source/Diagnostics/ReducedDiags/) ----------------------
 * The two reductions behind FieldEnergy, ParticleEnergy, ParticleMomentum and ParticleNumber.
 * wxa_reduce_field: over the points [lo, hi) (global indices) of one component, sum of squares and largest magnitude
 *   -- what MultiFab::norm2(0, periodicity) and norminf give FieldEnergy::ComputeDiags (FieldEnergy.cpp:81-157); the caller
 *   chooses the box so that a point shared by two bricks or by the two ends of a periodic direction is counted once
 *   (amrex's owner mask).  sum_sq / max_abs: host pointers, either may be null.
 * wxa_reduce_particles: over the live particles of p (retired slots skipped), out[0] = sum w Ekin with
 *   Ekin = m u^2 / (1 + gamma) (Algorithms::KineticEnergy, Source/Particles/Algorithms/KineticEnergy.H:33-47) or
 *   m_e c |u| for photon != 0 (:59-67), out[1] = sum w, out[2..4] = sum w m u_{x,y,z} (ParticleMomentum.cpp:143-156;
 *   m = m_e for photons), out[5] = number of live particles (ParticleNumber.cpp:97-139).  Host pointer.
 * Both are two-pass block reductions in a fixed order: the same input gives the same bits.  They block on the stream. */
wxa_status wxa_reduce_field(const wxa_field_view* f, const int32_t lo[3], const int32_t hi[3], double* sum_sq,
                            double* max_abs, void* stream);
wxa_status wxa_reduce_particles(const wxa_particle_view* p, double mass, int32_t photon, double out[6], void* stream);
/* warpx.reduced_diags_names / <name>.type / <name>.intervals / <name>.path (MultiReducedDiags.cpp:36-83,
 * ReducedDiags.cpp:26-72): type in {"FieldEnergy", "ParticleEnergy", "ParticleMomentum", "ParticleNumber"}; intervals in
 * the reference's slice syntax ("5", "0:100:10", "3:7, 20:"; utils::parser::IntervalsParser); path = output directory
 * with a trailing '/' (the reference's default is "./diags/reducedfiles/"), null or "" = nothing is written.  From then
 * on every step whose number is in the intervals computes the row (all bricks; sums over the bricks of a run) and
 * brick 0 appends it to <path><name>.txt in the reference's format (header line "#[0]step() [1]time(s) ...", 14
 * digits, ReducedDiags.cpp:97-125); step 0 is written at the first wxa_sim_evolve.  Add after the species. */
wxa_status wxa_sim_add_reduced_diag(wxa_sim* s, const char* name, const char* type, const char* intervals,
                                    const char* path);
/* the data columns of diagnostic `name` (everything after step and time): their number in *n, the values of the last
 * row computed into out[capacity]; compute_now != 0 evaluates the diagnostic at the current state first (no row is
 * written) */
wxa_status wxa_sim_reduced_diag_data(wxa_sim* s, const char* name, int32_t compute_now, double* out, int32_t capacity,
                                     int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* WARPX_AMD_H_ */
