// Shared device/host helpers for the gfx950 kernels.
#ifndef WXA_COMMON_HPP_
#define WXA_COMMON_HPP_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/warpx_amd.h"

// Value barrier: the compiler has to treat the fp64 value as redefined at this point (used to keep it from merging a
// deliberate second evaluation with the first one).  tests/hipcpu defines its own for the host compiler.
#ifndef WXA_OPAQUE_F64
#define WXA_OPAQUE_F64(v) asm volatile("" : "+v"(v))
#define WXA_OPAQUE_I32(v) asm volatile("" : "+v"(v))
#endif

// Register budget of a kernel as waves per SIMD (512 VGPRs per lane and SIMD: 4 -> 128, 3 -> 168 VGPRs)
#ifndef WXA_WAVES_PER_SIMD
#define WXA_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

namespace wxa {

// Source/ablastr/constant.H:41-50 (CODATA 2018), digit for digit.
namespace PhysConst {
constexpr double c = 299'792'458.;
constexpr double ep0 = 8.8541878128e-12;
constexpr double mu0 = 1.25663706212e-06;
constexpr double q_e = 1.602176634e-19;
constexpr double m_e = 9.1093837015e-31;
constexpr double r_e = 2.817940326204929e-15;   // classical electron radius (constant.H:63)
}

void set_last_error(const char* fmt, ...);

#define WXA_HIP_CHECK(expr)                                                          \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            ::wxa::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                  __FILE__, __LINE__);                               \
            return WXA_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

#define WXA_LAUNCH_CHECK()                                                           \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            ::wxa::set_last_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                                  __FILE__, __LINE__);                               \
            return WXA_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

#define WXA_REQUIRE(cond, msg)                                                       \
    do {                                                                             \
        if (!(cond)) {                                                               \
            ::wxa::set_last_error("invalid argument: %s (%s)", msg, #cond);          \
            return WXA_ERR_INVALID_ARG;                                              \
        }                                                                            \
    } while (0)

// Device-side Array4: element (i,j,k) at p[(i-lo0) + (j-lo1)*js + (k-lo2)*ks].
struct DevF {
    double* __restrict__ p;
    int lo0, lo1, lo2;
    int n0, n1, n2;
    long js, ks;
    __host__ __device__ inline long off(int i, int j, int k) const {
        return (long)(i - lo0) + (long)(j - lo1) * js + (long)(k - lo2) * ks;
    }
    __device__ inline double& operator()(int i, int j, int k) const { return p[off(i, j, k)]; }
};

inline DevF make_devf(const wxa_field_view& v) {
    DevF f;
    f.p = v.p;
    f.lo0 = v.lo[0]; f.lo1 = v.lo[1]; f.lo2 = v.lo[2];
    f.n0 = v.n[0]; f.n1 = v.n[1]; f.n2 = v.n[2];
    f.js = v.jstride; f.ks = v.kstride;
    return f;
}

struct Box3 {
    int lo[3];
    int hi[3];  // exclusive
};

inline Box3 valid_box(const wxa_field_view& v) {
    Box3 b;
    for (int d = 0; d < 3; ++d) {
        b.lo[d] = v.lo[d] + v.ng[d];
        b.hi[d] = v.lo[d] + v.n[d] - v.ng[d];
    }
    return b;
}

inline bool is_stag(const wxa_field_view& v, int a, int b, int c) {
    return v.stag[0] == a && v.stag[1] == b && v.stag[2] == c;
}
inline bool yee_E(const wxa_field_view f[3]) {
    return is_stag(f[0], 0, 1, 1) && is_stag(f[1], 1, 0, 1) && is_stag(f[2], 1, 1, 0);
}
inline bool yee_B(const wxa_field_view f[3]) {
    return is_stag(f[0], 1, 0, 0) && is_stag(f[1], 0, 1, 0) && is_stag(f[2], 0, 0, 1);
}

inline bool view_ok(const wxa_field_view& v) {
    return v.p != nullptr && v.n[0] > 0 && v.n[1] > 0 && v.n[2] > 0 && v.jstride >= v.n[0] &&
           v.kstride >= v.jstride * v.n[1] && v.ng[0] >= 0 && v.ng[1] >= 0 && v.ng[2] >= 0;
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so
// give each XCD a contiguous range of tiles (spatially adjacent tiles share an L2).
__device__ inline long xcd_tile_id(long bid, long ntiles) {
    const long per = (ntiles + 7) / 8;
    return (bid & 7) * per + (bid >> 3);
}
inline long xcd_grid_size(long ntiles) { return ((ntiles + 7) / 8) * 8; }
// The t-th tile of a traversal in blocks of 4 x 4 x 2 tiles (x fastest inside a block, blocks x fastest) instead of rows
// of nt0 tiles: the ~32 workgroups an XCD runs at a time are then one block, whose tiles share their halos' J lines
// while those are in the L2 (as a row, a tile met 2 of its 26 neighbours).  Grids that the blocks do not tile keep the rows.
__device__ inline long blocked_tile(long t, int nt0, int nt1, int nt2) {
    if ((nt0 & 3) || (nt1 & 3) || (nt2 & 1)) return t;
    const long b = t >> 5;
    const int w = (int)(t & 31), bx = nt0 >> 2, by = nt1 >> 2;
    const long i = 4 * (b % bx) + (w & 3), j = 4 * ((b / bx) % by) + ((w >> 2) & 3), k = 2 * (b / ((long)bx * by)) + (w >> 4);
    return i + nt0 * (j + (long)nt1 * k);
}
constexpr long WXA_NUM_CU = 256;   // MI355X: 8 XCDs x 32 CUs (persistent kernels launch one workgroup per CU)

// hardware fp64 atomic add (global_atomic_add_f64 / ds_add_f64), no CAS loop
__device__ inline void atomic_add_f64(double* addr, double v) { unsafeAtomicAdd(addr, v); }

}  // namespace wxa
#endif
