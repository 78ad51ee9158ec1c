// Back-transformed diagnostics, particles: the device part of BackTransformParticleFunctor
// (Source/Diagnostics/ComputeDiagFunctors/BackTransformParticleFunctor.cpp:76-152, .H:49-62 SelectParticles, .H:106-168
// LorentzTransformParticles).  A diagnostic kernel, not on the step path: one lane per particle, the (few) particles that
// crossed the snapshot's plane during the step are transformed and appended to the output through one atomic each.
#include <hip/hip_runtime.h>

#include "../common.hpp"
#include "../gather_body.hpp"

namespace wxa {

struct BtdParams {
    double z_boost, z_boost_old;   // the snapshot's plane now and one step ago (boosted frame)
    double t_boost, dt, t_lab;
    double gamma, beta;
};

__global__ void __launch_bounds__(256)
btd_select_particles_kernel(PV p, const double* __restrict__ xo, const double* __restrict__ yo, const double* __restrict__ zo,
                            const double* __restrict__ uxo, const double* __restrict__ uyo, const double* __restrict__ uzo,
                            BtdParams b, double* __restrict__ out, long cap, unsigned long long* __restrict__ count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.np) return;
    if (p.id && p.id[i] == WXA_IDCPU_RETIRED) return;   // handed to a neighbour or absorbed: not a particle any more
    const double zpnew = p.z[i], zpold = zo[i];
    // SelectParticles: the particle and the plane have crossed during the step
    if (!((zpnew >= b.z_boost && zpold <= b.z_boost_old) || (zpnew <= b.z_boost && zpold >= b.z_boost_old))) return;
    constexpr double c = PhysConst::c;
    const double inv_c2 = 1.0 / (c * c);
    const double uzfrm = -b.gamma * b.beta * c;
    const double uxn = p.ux[i], uyn = p.uy[i], uzn = p.uz[i];
    const double gamma_new_p = sqrt(1.0 + inv_c2 * (uxn * uxn + uyn * uyn + uzn * uzn));
    const double gamma_old_p = sqrt(1.0 + inv_c2 * (uxo[i] * uxo[i] + uyo[i] * uyo[i] + uzo[i] * uzo[i]));
    const double t_new_p = b.gamma * b.t_boost - uzfrm * zpnew * inv_c2;
    const double z_new_p = b.gamma * (zpnew + b.beta * c * b.t_boost);
    const double uz_new_p = b.gamma * uzn - gamma_new_p * uzfrm;
    const double t_old_p = b.gamma * (b.t_boost - b.dt) - uzfrm * zpold * inv_c2;
    const double z_old_p = b.gamma * (zpold + b.beta * c * (b.t_boost - b.dt));
    const double uz_old_p = b.gamma * uzo[i] - gamma_old_p * uzfrm;
    const double weight_old = (t_new_p - b.t_lab) / (t_new_p - t_old_p);   // interpolate in time to t_lab
    const double weight_new = (b.t_lab - t_old_p) / (t_new_p - t_old_p);
    const unsigned long long at = atomicAdd(count, 1ull);
    if ((long)at >= cap) return;   // counted, not stored: the caller sees count > cap and repeats with a larger buffer
    out[0 * cap + at] = xo[i] * weight_old + p.x[i] * weight_new;
    out[1 * cap + at] = yo[i] * weight_old + p.y[i] * weight_new;
    out[2 * cap + at] = z_old_p * weight_old + z_new_p * weight_new;
    out[3 * cap + at] = p.w[i];
    out[4 * cap + at] = uxo[i] * weight_old + uxn * weight_new;
    out[5 * cap + at] = uyo[i] * weight_old + uyn * weight_new;
    out[6 * cap + at] = uz_old_p * weight_old + uz_new_p * weight_new;
}

}  // namespace wxa

extern "C" wxa_status wxa_btd_select_particles(const wxa_particle_view* p, const double* const old6[6], double z_boost,
                                               double z_boost_old, double t_boost, double dt, double t_lab,
                                               double gamma_boost, double* out, int64_t capacity, int64_t* n_selected,
                                               void* stream) {
    using namespace wxa;
    WXA_REQUIRE(p && old6 && out && n_selected && capacity > 0 && gamma_boost > 1.0, "bad argument");
    *n_selected = 0;
    if (p->np == 0) return WXA_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* dcount = nullptr;
    WXA_HIP_CHECK(hipMalloc(&dcount, sizeof(unsigned long long)));
    WXA_HIP_CHECK(hipMemsetAsync(dcount, 0, sizeof(unsigned long long), st));
    BtdParams b{z_boost, z_boost_old, t_boost, dt, t_lab, gamma_boost, std::sqrt(1.0 - 1.0 / (gamma_boost * gamma_boost))};
    const PV pv = make_pv(*p);
    hipLaunchKernelGGL(btd_select_particles_kernel, dim3((unsigned)((p->np + 255) / 256)), dim3(256), 0, st, pv, old6[0],
                       old6[1], old6[2], old6[3], old6[4], old6[5], b, out, (long)capacity, dcount);
    unsigned long long h = 0;
    WXA_HIP_CHECK(hipMemcpyAsync(&h, dcount, sizeof(h), hipMemcpyDeviceToHost, st));
    WXA_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(dcount);
    *n_selected = (int64_t)h;
    WXA_LAUNCH_CHECK();
    return WXA_OK;
}
