// Reduced diagnostics, device part: the two reductions behind FieldEnergy (Source/Diagnostics/ReducedDiags/FieldEnergy.cpp:
// 81-157: MultiFab::norm2 of the six components), ParticleEnergy (ParticleEnergy.cpp:95-200: amrex::ParticleReduce of
// w Ekin and w), ParticleMomentum (ParticleMomentum.cpp:122-253: w m u and w) and ParticleNumber (ParticleNumber.cpp:97-139).
// Diagnostic kernels, not on the step path; they define the parity metric of BASELINE.json ("field energies and
// particle moments") on the device, so that a 256^3 run is compared without copying its fields to the host.
//
// Both are grid-stride passes into one partial per workgroup (wave shuffles, then the waves' partials through the LDS)
// followed by a one-workgroup pass over the partials: no atomics, a fixed order -- the same input gives the same bits.
#include <hip/hip_runtime.h>

#include "../common.hpp"
#include "../gather_body.hpp"

namespace wxa {

constexpr int RED_NT = 256;          // lanes per workgroup (4 waves)
constexpr int RED_MAX_BLOCKS = 1024; // partials of the first pass = lanes x 4 of the second

template <int NV>
struct RedVals {
    double s[NV];    // sums
    double m;        // a maximum (of non-negative values)
};

// the workgroup's total in thread 0
template <int NV>
__device__ __forceinline__ RedVals<NV> block_reduce(RedVals<NV> v) {
    __shared__ double part[RED_NT / 64][NV + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int c = 0; c < NV; ++c) v.s[c] += __shfl_down(v.s[c], d);
        v.m = fmax(v.m, __shfl_down(v.m, d));
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NV; ++c) part[wave][c] = v.s[c];
        part[wave][NV] = v.m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < RED_NT / 64; ++w) {
#pragma unroll
            for (int c = 0; c < NV; ++c) v.s[c] += part[w][c];
            v.m = fmax(v.m, part[w][NV]);
        }
    }
    __syncthreads();   // `part` may be written again by the caller's next reduction
    return v;
}

// second pass: partial[b][NV + 1], b < nb, summed by one workgroup in a fixed order
template <int NV>
__global__ void __launch_bounds__(RED_NT) reduce_partials_kernel(const double* __restrict__ partial, int nb,
                                                                  double* __restrict__ out) {
    RedVals<NV> v;
#pragma unroll
    for (int c = 0; c < NV; ++c) v.s[c] = 0.0;
    v.m = 0.0;
    for (int b = threadIdx.x; b < nb; b += RED_NT) {
#pragma unroll
        for (int c = 0; c < NV; ++c) v.s[c] += partial[(long)b * (NV + 1) + c];
        v.m = fmax(v.m, partial[(long)b * (NV + 1) + NV]);
    }
    v = block_reduce<NV>(v);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NV; ++c) out[c] = v.s[c];
        out[NV] = v.m;
    }
}

// sum of squares and largest magnitude over the box [lo, hi) of one component (i fastest: coalesced rows)
__global__ void __launch_bounds__(RED_NT) reduce_field_kernel(DevF f, int lo0, int lo1, int lo2, int n0, int n1, int n2,
                                                               double* __restrict__ partial) {
    RedVals<1> v;
    v.s[0] = 0.0;
    v.m = 0.0;
    const long npts = (long)n0 * n1 * n2;
    for (long a = (long)blockIdx.x * RED_NT + threadIdx.x; a < npts; a += (long)gridDim.x * RED_NT) {
        const int i = (int)(a % n0), j = (int)((a / n0) % n1), k = (int)(a / ((long)n0 * n1));
        const double x = f.p[f.off(lo0 + i, lo1 + j, lo2 + k)];
        v.s[0] += x * x;
        v.m = fmax(v.m, fabs(x));
    }
    v = block_reduce<1>(v);
    if (threadIdx.x == 0) {
        partial[(long)blockIdx.x * 2 + 0] = v.s[0];
        partial[(long)blockIdx.x * 2 + 1] = v.m;
    }
}

// w Ekin, w, w m ux, w m uy, w m uz, live count (as a double: exact below 2^53)
__global__ void __launch_bounds__(RED_NT) reduce_particles_kernel(PV p, double mass, int photon,
                                                                   double* __restrict__ partial) {
    RedVals<6> v;
#pragma unroll
    for (int c = 0; c < 6; ++c) v.s[c] = 0.0;
    v.m = 0.0;
    constexpr double c = PhysConst::c;
    constexpr double inv_c2 = 1.0 / (c * c);
    constexpr double me_c = PhysConst::m_e * c;
    for (long i = (long)blockIdx.x * RED_NT + threadIdx.x; i < p.np; i += (long)gridDim.x * RED_NT) {
        if (p.id && p.id[i] == WXA_IDCPU_RETIRED) continue;   // handed to a neighbour or absorbed
        const double w = p.w[i], ux = p.ux[i], uy = p.uy[i], uz = p.uz[i];
        const double u2 = ux * ux + uy * uy + uz * uz;
        double ekin;
        if (photon) {
            ekin = me_c * sqrt(u2);                       // KineticEnergyPhotons (KineticEnergy.H:59-67)
        } else {
            const double gamma = sqrt(1.0 + u2 * inv_c2);   // KineticEnergy (KineticEnergy.H:33-47)
            ekin = 1.0 / (1.0 + gamma) * mass * u2;
        }
        v.s[0] += w * ekin;
        v.s[1] += w;
        v.s[2] += w * mass * ux;
        v.s[3] += w * mass * uy;
        v.s[4] += w * mass * uz;
        v.s[5] += 1.0;
    }
    v = block_reduce<6>(v);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) partial[(long)blockIdx.x * 7 + cc] = v.s[cc];
        partial[(long)blockIdx.x * 7 + 6] = 0.0;
    }
}

// Scratch of the two reductions: RED_MAX_BLOCKS partial rows + the result, allocated once per host thread and kept
// (with <rd>.intervals = 1 these entry points run inside the time loop every step, several times per row: a hipMalloc /
// hipFree pair per call is an allocator round trip and an implicit device-wide sync each -- ADVICE round 3)
// Keyed by the device that is current when the call is made (a host thread that drives simulations on two devices through
// hipSetDevice gets a scratch on each), and never freed at thread or process exit: by then the HIP runtime may be gone,
// and the few KB per device and thread are the process's to the end anyway.
static double* red_scratch(size_t doubles) {
    struct Holder {
        double* p = nullptr;
        size_t n = 0;
    };
    constexpr int MAX_DEV = 16;
    static thread_local Holder held[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    Holder& h = held[dev];
    if (doubles > h.n) {
        if (h.p) (void)hipFree(h.p);
        h.p = nullptr; h.n = 0;
        if (hipMalloc(&h.p, sizeof(double) * doubles) != hipSuccess) return nullptr;
        h.n = doubles;
    }
    return h.p;
}

static int red_blocks(long n) {
    const long nb = (n + RED_NT - 1) / RED_NT;
    return (int)(nb < 1 ? 1 : nb > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : nb);
}

}  // namespace wxa

extern "C" wxa_status wxa_reduce_field(const wxa_field_view* f, const int32_t lo[3], const int32_t hi[3], double* sum_sq,
                                       double* max_abs, void* stream) {
    using namespace wxa;
    WXA_REQUIRE(f && f->p && lo && hi, "null argument");
    for (int d = 0; d < 3; ++d)
        WXA_REQUIRE(lo[d] >= f->lo[d] && hi[d] <= f->lo[d] + f->n[d], "box outside the array");
    if (sum_sq) *sum_sq = 0.0;
    if (max_abs) *max_abs = 0.0;
    const long npts = (long)std::max(0, hi[0] - lo[0]) * std::max(0, hi[1] - lo[1]) * std::max(0, hi[2] - lo[2]);
    if (npts == 0) return WXA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = red_blocks(npts);
    double* scratch = red_scratch(2 * ((size_t)nb + 1));   // nb partials + the result
    WXA_REQUIRE(scratch, "scratch allocation failed");
    hipLaunchKernelGGL(reduce_field_kernel, dim3((unsigned)nb), dim3(RED_NT), 0, st, make_devf(*f), lo[0], lo[1], lo[2],
                       hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], scratch);
    hipLaunchKernelGGL(reduce_partials_kernel<1>, dim3(1), dim3(RED_NT), 0, st, scratch, nb, scratch + 2 * (size_t)nb);
    double h[2] = {0.0, 0.0};
    const hipError_t e1 = hipMemcpyAsync(h, scratch + 2 * (size_t)nb, sizeof(h), hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    WXA_HIP_CHECK(e1);
    WXA_HIP_CHECK(e2);
    WXA_LAUNCH_CHECK();
    if (sum_sq) *sum_sq = h[0];
    if (max_abs) *max_abs = h[1];
    return WXA_OK;
}

extern "C" wxa_status wxa_reduce_particles(const wxa_particle_view* p, double mass, int32_t photon, double out[6],
                                           void* stream) {
    using namespace wxa;
    WXA_REQUIRE(pv_ok(p) && out, "bad argument");
    for (int c = 0; c < 6; ++c) out[c] = 0.0;
    if (p->np == 0) return WXA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = red_blocks((long)p->np);
    double* scratch = red_scratch(7 * ((size_t)nb + 1));
    WXA_REQUIRE(scratch, "scratch allocation failed");
    hipLaunchKernelGGL(reduce_particles_kernel, dim3((unsigned)nb), dim3(RED_NT), 0, st, make_pv(*p), mass, (int)photon,
                       scratch);
    hipLaunchKernelGGL(reduce_partials_kernel<6>, dim3(1), dim3(RED_NT), 0, st, scratch, nb, scratch + 7 * (size_t)nb);
    double h[7] = {};
    const hipError_t e1 = hipMemcpyAsync(h, scratch + 7 * (size_t)nb, sizeof(h), hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    WXA_HIP_CHECK(e1);
    WXA_HIP_CHECK(e2);
    WXA_LAUNCH_CHECK();
    for (int c = 0; c < 6; ++c) out[c] = h[c];
    return WXA_OK;
}
