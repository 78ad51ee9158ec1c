// TEST INFRASTRUCTURE -- never shipped, never linked into the product.
//
// A minimal HIP-on-CPU execution model, enough to compile the product's .hip sources
// (warpx_amd/csrc/*.hip, unmodified) with g++ and run their kernels on the host:
// tests/hipcpu/Makefile puts this directory in front of the include path, so that
// <hip/hip_runtime.h> and <hipcub/hipcub.hpp> resolve here.  It exists because this
// build container has no GPU: the kernels' logic (indexing, launch geometry, barriers,
// wave collectives, LDS tiles, atomics) can be checked against the oracle before the
// code reaches an MI355X.  It says nothing about performance, and nothing about what
// the gfx950 compiler does with the same source.
//
// Execution model: a launch runs its workgroups one after the other; the work-items of a
// workgroup are fibers (own x86-64 stack switch) of the calling thread, scheduled round-robin and
// switched only at __syncthreads() and at wave collectives (__shfl*, __ballot), 64 lanes
// per wavefront in linear work-item order.  Work-items that have returned do not take
// part in barriers or collectives (as on the hardware).  A collective that only some of
// the live lanes of a wavefront reach executes with those lanes once every other live
// work-item of the workgroup is blocked at a barrier (the partial exec mask of a divergent
// branch); lanes of one wavefront meeting in different kinds of collectives are an error.
// __shared__ is a function-local static: one workgroup
// runs at a time (launches are serialised by a process-wide lock).
#ifndef WXA_TESTS_HIPCPU_RUNTIME_H_
#define WXA_TESTS_HIPCPU_RUNTIME_H_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_SYMBOL(x) x
#define HIPCPU_EMULATION 1

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipcpu_uint3 { unsigned x, y, z; };

typedef struct hipcpu_stream* hipStream_t;
typedef struct hipcpu_event* hipEvent_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorOutOfMemory = 2;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
constexpr unsigned hipStreamNonBlocking = 1;

namespace hipcpu {
struct LaneView {            // what a work-item sees; refreshed by the scheduler on every switch
    hipcpu_uint3 tid, bid;
    dim3 bdim, gdim;
};
extern LaneView g_lane;
void run_grid(dim3 grid, dim3 block, void (*entry)(void*), void* closure, const char* name);
void sync_threads();
// all live lanes of the calling lane's wavefront publish `v`; returns the 64 published values
// and the mask of lanes that took part
const uint64_t* wave_exchange(int kind, uint64_t v, uint64_t* mask, int* lane);

template <class T> inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "wave collectives carry at most 8 bytes");
    uint64_t b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T> inline T from_bits(uint64_t b) {
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}

template <class... P, class... A>
inline void launch(const char* name, void (*kernel)(P...), dim3 grid, dim3 block, A&&... a) {
    std::tuple<std::decay_t<P>...> params(static_cast<std::decay_t<P>>(std::forward<A>(a))...);
    struct Closure { void (*k)(P...); std::tuple<std::decay_t<P>...>* t; } c{kernel, &params};
    run_grid(grid, block, [](void* p) { auto* cl = static_cast<Closure*>(p); std::apply(cl->k, *cl->t); }, &c, name);
}
}  // namespace hipcpu

#define threadIdx (::hipcpu::g_lane.tid)
#define blockIdx (::hipcpu::g_lane.bid)
#define blockDim (::hipcpu::g_lane.bdim)
#define gridDim (::hipcpu::g_lane.gdim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hipcpu::launch(#kernel, kernel, dim3(grid), dim3(block), ##__VA_ARGS__)

// ---- runtime API (synchronous; "device" memory is host memory) -------------------------
hipError_t hipMalloc(void** p, size_t bytes);
template <class T> inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }   // one "device"
const char* hipGetErrorString(hipError_t);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
template <class T> inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { std::memcpy(dst, &sym, n); return hipSuccess; }
template <class T> inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { std::memcpy(&sym, src, n); return hipSuccess; }

// ---- device functions --------------------------------------------------------------------
inline void __syncthreads() { ::hipcpu::sync_threads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

inline unsigned long long __ballot(int pred) {
    uint64_t mask; int lane;
    const uint64_t* v = ::hipcpu::wave_exchange(0, pred ? 1u : 0u, &mask, &lane);
    unsigned long long r = 0;
    for (int l = 0; l < 64; ++l) if (((mask >> l) & 1ull) && v[l]) r |= 1ull << l;
    return r;
}
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred); }
inline int __any(int pred) { return __ballot(pred) != 0ull; }
inline int __all(int pred) { return __ballot(!pred) == 0ull; }

template <class T> inline T __shfl(T var, int src, int width = 64) {
    uint64_t mask; int lane;
    const uint64_t* v = ::hipcpu::wave_exchange(1, ::hipcpu::to_bits(var), &mask, &lane);
    const int base = lane & ~(width - 1);
    const int s = base + (src & (width - 1));
    return ((mask >> s) & 1ull) ? ::hipcpu::from_bits<T>(v[s]) : var;
}
template <class T> inline T __shfl_up(T var, unsigned delta, int width = 64) {
    uint64_t mask; int lane;
    const uint64_t* v = ::hipcpu::wave_exchange(2, ::hipcpu::to_bits(var), &mask, &lane);
    const int in = lane & (width - 1);
    if (in < (int)delta) return var;
    const int s = lane - (int)delta;
    return ((mask >> s) & 1ull) ? ::hipcpu::from_bits<T>(v[s]) : var;
}
template <class T> inline T __shfl_down(T var, unsigned delta, int width = 64) {
    uint64_t mask; int lane;
    const uint64_t* v = ::hipcpu::wave_exchange(3, ::hipcpu::to_bits(var), &mask, &lane);
    const int in = lane & (width - 1);
    if (in + (int)delta >= width) return var;
    const int s = lane + (int)delta;
    return ((mask >> s) & 1ull) ? ::hipcpu::from_bits<T>(v[s]) : var;
}
template <class T> inline T __shfl_xor(T var, int lmask, int width = 64) {
    uint64_t mask; int lane;
    const uint64_t* v = ::hipcpu::wave_exchange(4, ::hipcpu::to_bits(var), &mask, &lane);
    const int s = lane ^ lmask;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) return var;
    return ((mask >> s) & 1ull) ? ::hipcpu::from_bits<T>(v[s]) : var;
}

inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
inline long long clock64() { return 0; }
inline long long wall_clock64() { return 0; }

// work-items of one process never run concurrently: plain read-modify-write
template <class T> inline T hipcpu_rmw_add(T* a, T v) { T o = *a; *a = o + v; return o; }
inline int atomicAdd(int* a, int v) { return hipcpu_rmw_add(a, v); }
inline unsigned atomicAdd(unsigned* a, unsigned v) { return hipcpu_rmw_add(a, v); }
inline unsigned long long atomicAdd(unsigned long long* a, unsigned long long v) { return hipcpu_rmw_add(a, v); }
inline float atomicAdd(float* a, float v) { return hipcpu_rmw_add(a, v); }
inline double atomicAdd(double* a, double v) { return hipcpu_rmw_add(a, v); }
inline double unsafeAtomicAdd(double* a, double v) { return hipcpu_rmw_add(a, v); }
inline float unsafeAtomicAdd(float* a, float v) { return hipcpu_rmw_add(a, v); }
inline int atomicMax(int* a, int v) { int o = *a; if (v > o) *a = v; return o; }
inline int atomicMin(int* a, int v) { int o = *a; if (v < o) *a = v; return o; }
inline unsigned long long atomicMax(unsigned long long* a, unsigned long long v) { auto o = *a; if (v > o) *a = v; return o; }
inline int atomicExch(int* a, int v) { int o = *a; *a = v; return o; }
inline int atomicCAS(int* a, int cmp, int v) { int o = *a; if (o == cmp) *a = v; return o; }
inline unsigned atomicOr(unsigned* a, unsigned v) { unsigned o = *a; *a = o | v; return o; }

// asynchronous global -> LDS copy, used by the product as an L2 prefetch whose LDS target is scratch
template <class G, class L> inline void __builtin_amdgcn_global_load_lds(G, L, int, int, int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // used on wave-uniform values only
#define WXA_OPAQUE_F64(v) asm volatile("" : "+x"(v))
#define WXA_OPAQUE_I32(v) asm volatile("" : "+r"(v))
#define WXA_WAVES_PER_SIMD(n)
#define WXA_LATE_KERNARG(T, first_param) (&(first_param))
#define WXA_OPAQUE_UNIFORM_F64(v)
// the inline-asm LDS reads of gather_body.hpp: plain loads here, nothing to wait for
typedef const char* wxa_lds_addr;
#define WXA_LDS_ADDR(p) ((const char*)(p))
#define WXA_LDS_READ_B64(dst, addr, off) ((dst) = *(const double*)((addr) + (off)))
#define WXA_LDS_WAIT1(n, a) ((void)0)
#define WXA_LDS_WAIT2(n, a, b) ((void)0)
#define WXA_LDS_WAIT3(n, a, b, c) ((void)0)
#define WXA_LDS_WAIT4(n, a, b, c, d) ((void)0)

// v_permlane32_swap based helpers of deposit_body.hpp: lanes l and l ^ 32 exchange through the wave shuffle
#define WXA_HAVE_SWAP_ADD_HALVES 1
namespace wxa {
inline double swap_add_halves(const double v_lo_planes, const double v_hi_planes) {
    const double plo = __shfl_xor(v_lo_planes, 32), phi = __shfl_xor(v_hi_planes, 32);
    return (threadIdx.x & 32) ? phi + v_hi_planes : v_lo_planes + plo;
}
inline int partner32(const int v) { return __shfl_xor(v, 32); }
#define WXA_HAVE_PARTNER32 1
inline double partner32_f64(const double v) { return __shfl_xor(v, 32); }
#define WXA_HAVE_LANE_XOR1
inline double lane_xor1(const double v) { return __shfl_xor(v, 1); }
#define WXA_HAVE_WAVE_SUM_F64
inline double wave_sum_f64(double v) {   // every lane gets the sum (the product reads lane 63)
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
}

// HIP's unqualified min / max over mixed integer types
#define HIPCPU_MINMAX(A, B, R)                                   \
    inline R min(A a, B b) { return (R)a < (R)b ? (R)a : (R)b; } \
    inline R max(A a, B b) { return (R)a > (R)b ? (R)a : (R)b; }
HIPCPU_MINMAX(int, int, int)
HIPCPU_MINMAX(long, long, long)
HIPCPU_MINMAX(int, long, long)
HIPCPU_MINMAX(long, int, long)
HIPCPU_MINMAX(unsigned, unsigned, unsigned)
HIPCPU_MINMAX(unsigned long, unsigned long, unsigned long)
HIPCPU_MINMAX(long long, long long, long long)
HIPCPU_MINMAX(unsigned long long, unsigned long long, unsigned long long)
HIPCPU_MINMAX(double, double, double)
HIPCPU_MINMAX(float, float, float)
#undef HIPCPU_MINMAX

#endif
